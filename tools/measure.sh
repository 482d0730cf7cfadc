#!/bin/bash
# Round measurement recipe, run on the GPU box:  gpurun --timeout 1500 -- 'bash tools/measure.sh r01'
# Writes raw outputs under gpurun_out/<tag>/; tools/summarize_profile.py condenses them into profiles/.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; tail -3 "$O/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
python bench.py > "$O/bench_atrium.json" 2> "$O/bench_atrium.err"; tail -1 "$O/bench_atrium.json"
python bench.py --workload s256 --steps 10 --warmup 2 --cpu-seconds 6 > "$O/bench_s256.json" 2> "$O/bench_s256.err"; tail -1 "$O/bench_s256.json"
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_atrium" -- $BENCH > "$O/stats_atrium.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_s256" -- $BENCH --workload s256 --steps 5 --warmup 1 > "$O/stats_s256.log" 2>&1
# one frame at a time: the per-launch duration with no second frame sharing the device
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_atrium_nopipe" -- $BENCH --no-pipeline > "$O/stats_atrium_nopipe.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_s256_nopipe" -- $BENCH --workload s256 --steps 5 --warmup 1 --no-pipeline > "$O/stats_s256_nopipe.log" 2>&1
# the reference's own bench scene, lit on the device (N2): bench line + kernel stats of the light gather kernel
python bench.py --workload light-bench --steps 200 --warmup 10 > "$O/bench_lightbench.json" 2> "$O/bench_lightbench.err"; tail -c 700 "$O/bench_lightbench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_lightbench" -- $BENCH --workload light-bench --steps 50 > "$O/stats_lightbench.log" 2>&1
# the sim + render loop with the light updated on the device every frame
python bench.py --workload relight --steps 200 --warmup 10 --no-cpu-baseline > "$O/bench_relight.json" 2> "$O/bench_relight.err"; tail -c 600 "$O/bench_relight.json"
# instruction issue-rate micro-benchmark (tools/ubench/issue_rate.hip): what bounds the trace kernel
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip > /dev/null 2>&1 && timeout 300 ./issue_rate ) > "$O/issue_rate.txt" 2>&1
bash tools/measure_pmc.sh "$TAG"
# keep only the small CSVs (agent_info / counter_collection / kernel_stats), drop anything large
find "$O" -type f -size +4M -delete
du -sh "$O"
