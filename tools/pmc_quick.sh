#!/bin/bash
# SQ counter passes for one workload, one frame at a time (counters are device-wide over a kernel's window)
# usage: bash tools/pmc_quick.sh <tag> [workload]
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}; W=${2:-atrium}
O=gpurun_out/$TAG; mkdir -p $O
X="--no-pipeline --workload $W --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq1_$W -- python bench.py $X > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 --output-format csv -d $O/sq2_$W -- python bench.py $X > /dev/null 2>&1
find $O -type f -size +4M -delete
python - <<PY
import csv, glob, collections
for d in ("sq1_$W", "sq2_$W"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_image_kernel" in r["Kernel_Name"] and (", false, false>" in r["Kernel_Name"] or ", false, false, " in r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        # one row per dispatch (and per dimension instance): sum per dispatch
        n = len(v)
        print(d, k, "n=%d" % n, "mean per row %.4g" % (sum(v) / n), "total %.6g" % sum(v))
PY
