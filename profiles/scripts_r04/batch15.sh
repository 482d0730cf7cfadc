#!/bin/bash
# Round 4, batch 15: L2 hit rate and HBM-side fetch of the trace kernel with one tile dispenser against XCD-local queues (s256 and atrium).
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b15; mkdir -p $O
for W in s256 atrium; do
  X="--steps 6 --warmup 2"; [ $W = s256 ] && X="--workload s256 --steps 3 --warmup 1"
  for cfg in "q1:AIC_TILE_QUEUES=1" "q8s3:AIC_SUPER_SHIFT=3" "q8s5:AIC_SUPER_SHIFT=5"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    BENCH="python bench.py $X --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 --no-pipeline"
    rm -rf $O/l2_${W}_$name $O/fetch_${W}_$name
    env $envs rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2_${W}_$name -- $BENCH > /dev/null 2>&1
    env $envs rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_${W}_$name -- $BENCH > /dev/null 2>&1
    python - <<PY
import csv, glob
def mean(d, counter):
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'trace_image_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter: vals.append(float(r['Counter_Value']))
    return sum(vals) / max(1, len(vals)), len(vals)
h, n = mean('$O/l2_${W}_$name', 'TCC_HIT_sum'); m, _ = mean('$O/l2_${W}_$name', 'TCC_MISS_sum'); f, _ = mean('$O/fetch_${W}_$name', 'FETCH_SIZE')
print('$W $name launches', n, 'L2 hits %.1f M misses %.1f M hit rate %.3f' % (h / 1e6, m / 1e6, h / max(1.0, h + m)), 'FETCH_SIZE raw %.1f MB (x2: %.1f MB)' % (f * 1024 / 1e6, f * 2048 / 1e6))
PY
  done
done
find $O -type f -size +1M -delete
