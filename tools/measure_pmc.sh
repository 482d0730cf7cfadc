#!/bin/bash
# Counter passes only (the second half of tools/measure.sh): gpurun --timeout 1200 -- 'bash tools/measure_pmc.sh r02'
# One frame at a time (--no-pipeline), no extras, one timed region: PMC values are device-wide over a kernel's
# execution window, so only identical, non-overlapping launches of the production kernel may be in the run.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/$TAG; mkdir -p "$O"
BENCH="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 --no-pipeline"
for W in atrium s256; do
  X=""; [ $W = s256 ] && X="--workload s256 --steps 3 --warmup 1"
  rm -rf "$O"/pmc_*_$W
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$W" -- $BENCH $X > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$W" -- $BENCH $X > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/pmc_l2_$W" -- $BENCH $X > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$O/pmc_sq1_$W" -- $BENCH $X > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$O/pmc_sq2_$W" -- $BENCH $X > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_EXP_GDS SQ_INSTS_GDS SQ_INSTS_WAVE32_LDS --output-format csv -d "$O/pmc_sq3_$W" -- $BENCH $X > /dev/null 2>&1
done
find "$O" -type f -size +4M -delete
ls "$O" | head -40; du -sh "$O"
