#!/bin/bash
# round 5, second final pass (the library after experiments J and L): bench lines first, then the GPU suite, a 6000-seed sweep (both production variants per seed),
# kernel durations (streamed and one at a time), the trace kernel's counters, the in-kernel phase counters / wave tails / rank shares. The light updater's kernels did
# not change after final.sh: its counter passes, its bench line and the issue-rate micro-benchmark are not repeated.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
TAG=r05; O=gpurun_out/$TAG; mkdir -p "$O" gpurun_out/r05final2
python bench.py > "$O/bench_atrium.json" 2> "$O/bench_atrium.err"; tail -c 300 "$O/bench_atrium.json"; echo
python bench.py --workload s256 --steps 10 --warmup 2 --cpu-seconds 6 > "$O/bench_s256.json" 2> "$O/bench_s256.err"; tail -c 200 "$O/bench_s256.json"; echo
python bench.py --workload relight --steps 200 --warmup 10 --no-cpu-baseline > "$O/bench_relight.json" 2> "$O/bench_relight.err"; tail -c 200 "$O/bench_relight.json"; echo
python bench.py --workload orbit --steps 60 --warmup 5 --no-cpu-baseline > "$O/bench_orbit.json" 2> "$O/bench_orbit.err"; tail -c 200 "$O/bench_orbit.json"; echo
python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; tail -2 "$O/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
AIC_FUZZ_N=6000 timeout 600 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r05final2/fuzz6000.log 2>&1; tail -2 gpurun_out/r05final2/fuzz6000.log
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 1"
for W in atrium s256; do
  X=""; [ $W = s256 ] && X="--workload s256 --steps 5 --warmup 1"
  rm -rf "$O"/stats_$W "$O"/stats_${W}_nopipe
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$W" -- $BENCH $X --no-extras > "$O/stats_$W.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_${W}_nopipe" -- $BENCH $X --no-extras --no-pipeline > "$O/stats_${W}_nopipe.log" 2>&1
done
bash tools/measure_pmc.sh "$TAG" > /dev/null 2>&1
bash tools/measure_round.sh r05 profile > gpurun_out/r05final2/profile.out 2>&1; tail -3 gpurun_out/r05final2/profile.out
find "$O" -type f -size +4M -delete
du -sh "$O"
