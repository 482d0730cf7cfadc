#!/bin/bash
# The light updater's part of the round measurement (tools/measure.sh has the whole recipe), run on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/measure_light.sh r02'
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/${TAG}_light
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; tail -3 "$O/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
python bench.py --workload light-bench --steps 200 --warmup 10 > "$O/bench_lightbench.json" 2> "$O/bench_lightbench.err"; tail -c 1500 "$O/bench_lightbench.json"
python bench.py --workload relight --steps 200 --warmup 10 --no-cpu-baseline > "$O/bench_relight.json" 2> "$O/bench_relight.err"; tail -c 900 "$O/bench_relight.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_lightbench" -- python bench.py --steps 50 --warmup 3 --no-cpu-baseline --workload light-bench > "$O/stats_lightbench.log" 2>&1
find "$O" -type f -size +4M -delete
du -sh "$O"
