cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06x; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|DCACHE)[A-Z_]*)\b" | sort -u > $O/counters.txt; cat $O/counters.txt | tr '\n' ' '; echo
X="--no-pipeline --workload atrium --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-extras"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -- python bench.py $X > $d.log 2>&1 || tail -3 $d.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_image_kernel" in r["Kernel_Name"] and ", false, false, " in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d" % len(v), "mean %.5g" % (sum(v) / len(v)))
PY
find $O -type f -size +4M -delete
