#!/bin/bash
# in-kernel phase counters / cycle split: needs variants/libaic_hip_prof.so (tools/build_variants.sh "prof:-DAIC_PROFILE")
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_prof.so all_is_cubes_amd/libaic_hip.so
for wl in ${AIC_PROF_WORKLOADS:-atrium s256}; do echo "== $wl"; python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep PROF | tail -31; done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
