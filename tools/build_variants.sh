#!/bin/bash
# Build kernel variants locally (hipcc cross-compiles): tools/build_variants.sh "name1:-DFOO=1" "name2:-DBAR=2 -DBAZ"
# Each becomes variants/libaic_hip_<name>.so; tools/exp.sh swaps them in on the GPU box.
# AIC_PATCH=<file>: the variants are built from a copy of csrc/ with that patch applied (e.g. profiles/scripts_r04/experiments_r01_r04.patch, which
# puts the measured-negative experiments of rounds 1-4 -- AIC_SPEC_STEPS, AIC_SHADE_STEP, AIC_PRIO_SHIFT, AIC_HURRY_STEPS, AIC_RAY_MIGRATION, AIC_LDS_PAD,
# AIC_SCHED_SIMPLE, AIC_TRIP_MIN -- back into aic_trace.hip as it stood at the commit that removed them: `git log -- profiles/scripts_r04/experiments_r01_r04.patch`).
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
C=all_is_cubes_amd/csrc
if [ -n "$AIC_PATCH" ]; then
  rm -rf variants/src && mkdir -p variants/src/all_is_cubes_amd && cp -r $C variants/src/all_is_cubes_amd/csrc && cp -r include variants/src/include
  ( cd variants/src && patch -p1 < "../../$AIC_PATCH" )
  C=variants/src/all_is_cubes_amd/csrc
fi
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c $C/aic_trace.hip -o variants/trace_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -x hip -c $C/aic_abi.cpp -o variants/abi_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c $C/aic_light.hip -o variants/light_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip -c $C/aic_multi.cpp -o variants/multi_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libaic_hip_$name.so variants/trace_$name.o variants/light_$name.o variants/abi_$name.o variants/multi_$name.o && rm variants/*_$name.o && echo "built $name" ) &
done
wait
