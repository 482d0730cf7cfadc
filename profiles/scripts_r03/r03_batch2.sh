#!/bin/bash
# Round 3, GPU batch 2: instruction counters of the new and of round 2's trace kernel (same box), in-kernel phase cycles and
# the per-wave tail of the new one, the extended issue-rate micro-benchmark.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b2; rm -rf $O; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for wl in atrium s256; do echo "== pmc new $wl"; bash tools/pmc_quick.sh r03b2/pmc_new $wl 2>&1 | tail -20; done
cp variants/libaic_hip_old.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== pmc old $wl"; bash tools/pmc_quick.sh r03b2/pmc_old $wl 2>&1 | tail -20; done
cp variants/libaic_hip_prof.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== prof $wl"; AIC_WAVE_PROF=$O/wave_prof_$wl.txt python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --min-seconds 0 2>&1 | grep PROF | tail -20; done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
echo "== issue rate"
timeout 600 tools/ubench/issue_rate > $O/issue_rate.txt 2>&1; head -30 $O/issue_rate.txt | cut -c1-260
