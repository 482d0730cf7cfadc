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
# counter passes trace one frame at a time (--no-pipeline): PMC values are device-wide over a kernel's
# execution window, so overlapping frames would be counted into each other
for W in atrium s256; do
  X="--no-pipeline"; [ $W = s256 ] && X="--no-pipeline --workload s256"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$W" -- $BENCH --steps 3 --warmup 1 $X > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$W" -- $BENCH --steps 3 --warmup 1 $X > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/pmc_l2_$W" -- $BENCH --steps 3 --warmup 1 $X > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$O/pmc_sq1_$W" -- $BENCH --steps 3 --warmup 1 $X > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d "$O/pmc_sq2_$W" -- $BENCH --steps 3 --warmup 1 $X > /dev/null 2>&1
done
# keep only the small CSVs (agent_info / counter_collection / kernel_stats), drop anything large
find "$O" -type f -size +4M -delete
du -sh "$O"
