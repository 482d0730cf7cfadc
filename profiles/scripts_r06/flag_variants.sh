#!/bin/bash
# Experiment Y: the trace kernels under other backend options (never swept before). Builds variants/libaic_hip_<name>.so = the tree's objects with aic_trace.o
# recompiled under the extra flags; the compiler's resource remarks of every kernel go to variants/flags/<name>.log. Run from the repository root (CPU only).
#   usage: profiles/scripts_r06/flag_variants.sh name "extra flags" [name "extra flags" ...]
cd "$(dirname "$0")/../.."
C=all_is_cubes_amd/csrc
mkdir -p variants/flags
build() {
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c $C/aic_trace.hip -o variants/flags/trace_$name.o > variants/flags/$name.log 2>&1 \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libaic_hip_$name.so variants/flags/trace_$name.o $C/aic_light.o $C/aic_abi.o $C/aic_multi.o \
    && echo "built $name ($*)" || echo "FAILED $name: $(tail -2 variants/flags/$name.log)"
  rm -f variants/flags/trace_$name.o
}
n=0
while [ $# -ge 2 ]; do
  build "$1" $2 &
  shift 2
  n=$((n + 1))
  if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
