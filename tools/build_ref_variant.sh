#!/bin/bash
# Build variants/libaic_hip_<name>.so from the sources of a git revision (default HEAD): the baseline a working-tree change is measured against on the same box
# (profiles/scripts_r06/exp.sh `lib:<name>`). usage: tools/build_ref_variant.sh [name] [rev]
set -e
cd "$(dirname "$0")/.."
NAME=${1:-head}; REV=${2:-HEAD}
D=variants/src_$NAME
rm -rf $D && mkdir -p $D && git archive $REV all_is_cubes_amd/csrc include | tar -x -C $D
C=$D/all_is_cubes_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
/opt/rocm/bin/hipcc $F -c $C/aic_trace.hip -o $D/trace.o &
/opt/rocm/bin/hipcc $F -x hip -c $C/aic_abi.cpp -o $D/abi.o &
/opt/rocm/bin/hipcc $F -c $C/aic_light.hip -o $D/light.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $C/aic_multi.cpp -o $D/multi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libaic_hip_$NAME.so $D/trace.o $D/light.o $D/abi.o $D/multi.o
rm -f $D/*.o
echo "built variants/libaic_hip_$NAME.so from $REV"
