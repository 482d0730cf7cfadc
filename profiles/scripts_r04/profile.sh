#!/bin/bash
# Phase counters (variants/libaic_hip_prof.so = -DAIC_PROFILE) and per-wave clocks (rayprof) of the production kernel variant, cold and warm frames.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_prof.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== $wl"; python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 2>&1 | grep PROF | tail -31; done > $O/prof.txt
cp variants/libaic_hip_rayprof.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do
  AIC_WAVE_PROF=$O/wave_cold_$wl.txt python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 > /dev/null 2>&1
  python tools/wave_tail.py $O/wave_cold_$wl.txt "$wl cold"; python tools/wave_rays.py $O/wave_cold_$wl.txt "$wl cold"
  AIC_WAVE_PROF=$O/wave_warm_$wl.txt python bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 > /dev/null 2>&1
  python tools/wave_tail.py $O/wave_warm_$wl.txt "$wl warm"; python tools/wave_rays.py $O/wave_warm_$wl.txt "$wl warm"
done > $O/wave_tail.txt
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
cat $O/prof.txt; cat $O/wave_tail.txt
