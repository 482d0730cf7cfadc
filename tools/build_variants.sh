#!/bin/bash
# Build kernel variants locally (hipcc cross-compiles): tools/build_variants.sh "name1:-DFOO=1" "name2:-DBAR=2 -DBAZ"
# Each becomes variants/libaic_hip_<name>.so; tools/exp.sh swaps them in on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
C=all_is_cubes_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c $C/aic_trace.hip -o variants/trace_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -x hip -c $C/aic_abi.cpp -o variants/abi_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c $C/aic_light.hip -o variants/light_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip -c $C/aic_multi.cpp -o variants/multi_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libaic_hip_$name.so variants/trace_$name.o variants/light_$name.o variants/abi_$name.o variants/multi_$name.o && rm variants/*_$name.o && echo "built $name" ) &
done
wait
