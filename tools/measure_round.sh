#!/bin/bash
# Round measurement recipe (round 3 on), run on the GPU box in TWO calls, because bench.py's `issue` / `roofline.traffic`
# objects quote the counter file of the same round:
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh r03 counters'   then   python tools/summarize_profile.py r03   (here)
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh r03 bench'      then   python tools/summarize_profile.py r03   (here)
#   gpurun --timeout 900  -- 'bash tools/measure_round.sh r04 profile'    (round 4 on: phase counters, wave tails, a rank's share; needs the prof variants)
# Raw outputs under gpurun_out/<tag>/; tools/summarize_profile.py condenses them into profiles/<tag>_*.
TAG=${1:-r03}; PHASE=${2:-counters}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/$TAG; mkdir -p "$O"
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 1"
if [ "$PHASE" = counters ]; then
  python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; tail -3 "$O/pytest_gpu.log"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
  # per-kernel durations: streamed, and one frame at a time (a launch alone on the device)
  for W in atrium s256; do
    X=""; [ $W = s256 ] && X="--workload s256 --steps 5 --warmup 1"
    rm -rf "$O"/stats_$W "$O"/stats_${W}_nopipe
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$W" -- $BENCH $X --no-extras > "$O/stats_$W.log" 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_${W}_nopipe" -- $BENCH $X --no-extras --no-pipeline > "$O/stats_${W}_nopipe.log" 2>&1
  done
  rm -rf "$O"/stats_lightbench
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_lightbench" -- $BENCH --workload light-bench --steps 50 --no-extras > "$O/stats_lightbench.log" 2>&1
  bash tools/measure_pmc.sh "$TAG"
  # the light kernel's counters (same rules: separate passes, nothing else in the run that matters)
  LB="python bench.py --workload light-bench --steps 5 --warmup 1 --no-cpu-baseline --no-extras --min-seconds 0 --no-pipeline"
  rm -rf "$O"/pmc_light_*
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$O/pmc_light_sq1" -- $LB > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$O/pmc_light_sq2" -- $LB > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_light_fetch" -- $LB > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_light_write" -- $LB > /dev/null 2>&1
  python tools/reduce_pmc_csv.py "$O"/pmc_light_sq1 "$O"/pmc_light_sq2 "$O"/pmc_light_fetch "$O"/pmc_light_write  # (the per-launch CSVs of ~1800 launches exceed what is kept)
  ( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o issue_rate issue_rate.hip > /dev/null 2>&1 && timeout 300 ./issue_rate ) > "$O/issue_rate.txt" 2>&1
elif [ "$PHASE" = bench ]; then
  python bench.py > "$O/bench_atrium.json" 2> "$O/bench_atrium.err"; tail -c 400 "$O/bench_atrium.json"; echo
  python bench.py --workload s256 --steps 10 --warmup 2 --cpu-seconds 6 > "$O/bench_s256.json" 2> "$O/bench_s256.err"; tail -c 300 "$O/bench_s256.json"; echo
  python bench.py --workload light-bench --steps 200 --warmup 10 > "$O/bench_lightbench.json" 2> "$O/bench_lightbench.err"; tail -c 300 "$O/bench_lightbench.json"; echo
  python bench.py --workload relight --steps 200 --warmup 10 --no-cpu-baseline > "$O/bench_relight.json" 2> "$O/bench_relight.err"; tail -c 500 "$O/bench_relight.json"; echo
  python bench.py --workload orbit --steps 60 --warmup 5 --no-cpu-baseline > "$O/bench_orbit.json" 2> "$O/bench_orbit.err"; tail -c 300 "$O/bench_orbit.json"; echo
  # the exchange step over the nccl backend on this one GPU (a one-rank group), device-side and host-side hand-off, with the HIP API
  # calls counted: no hipStreamSynchronize per frame may remain between a trace's submit and its gather in the first (round 4)
  for mode in "" "--host-handoff"; do
    rm -rf "$O/hip_gather1$mode"
    rocprofv3 --hip-trace --stats --output-format csv -d "$O/hip_gather1$mode" -- python bench.py --no-cpu-baseline --no-extras --gather-at-one $mode --steps 30 --warmup 3 --min-seconds 1 > "$O/bench_gather1$mode.json" 2> "$O/bench_gather1$mode.err"
    python tools/hip_handoff_summary.py "$O/hip_gather1$mode" "$O/bench_gather1$mode.json" "$mode"
  done | tee "$O/hip_handoff.txt"
elif [ "$PHASE" = profile ]; then
  # in-kernel phase counters and per-wave clocks: needs variants/libaic_hip_prof.so, variants/libaic_hip_rayprof.so (tools/build_variants.sh "prof:-DAIC_PROFILE -DAIC_POOL=54" "rayprof:-DAIC_PROFILE -DAIC_RAY_PROF -DAIC_POOL=54": the counters take LDS the pool gives up)
  cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
  cp variants/libaic_hip_prof.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium s256; do echo "== $wl"; python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 2>&1 | grep PROF | tail -39; done > $O/prof.txt
  cp variants/libaic_hip_rayprof.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium s256; do
    AIC_WAVE_PROF=$O/wave_cold_$wl.txt python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 > /dev/null 2>&1
    python tools/wave_tail.py $O/wave_cold_$wl.txt "$wl cold"; python tools/wave_rays.py $O/wave_cold_$wl.txt "$wl cold"
    AIC_WAVE_PROF=$O/wave_warm_$wl.txt python bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-pipeline --no-extras --no-secondary --min-seconds 0 > /dev/null 2>&1
    python tools/wave_tail.py $O/wave_warm_$wl.txt "$wl warm"; python tools/wave_rays.py $O/wave_warm_$wl.txt "$wl warm"
  done > $O/wave_tail.txt
  cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
  cat $O/prof.txt; cat $O/wave_tail.txt
  for np in 2 4 8; do python tools/rank_share.py $np 1 atrium; python tools/rank_share.py $np 8 atrium; done 2>&1 | grep -v amdgpu | tee $O/rank_share.txt
fi
find "$O" -type f -size +4M -delete
du -sh "$O"
