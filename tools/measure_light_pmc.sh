#!/bin/bash
# The light kernel's counter passes alone (part of tools/measure_round.sh counters): gpurun -- 'bash tools/measure_light_pmc.sh r03'
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/$TAG; mkdir -p "$O"
LB="python bench.py --workload light-bench --steps 5 --warmup 1 --no-cpu-baseline --no-extras --min-seconds 0 --no-pipeline"
rm -rf "$O"/pmc_light_*
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$O/pmc_light_sq1" -- $LB > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$O/pmc_light_sq2" -- $LB > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_light_fetch" -- $LB > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_light_write" -- $LB > /dev/null 2>&1
python tools/reduce_pmc_csv.py "$O"/pmc_light_sq1 "$O"/pmc_light_sq2 "$O"/pmc_light_fetch "$O"/pmc_light_write
find "$O" -type f -size +4M -delete
du -sh "$O"
