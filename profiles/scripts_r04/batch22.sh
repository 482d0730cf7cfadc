#!/bin/bash
# Round 4, batch 22: WRITE_SIZE of the trace kernel went from 127 MB to 1.2 GB per s256 frame with the XCD-local queues: who writes?
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b22; mkdir -p $O
W=s256; X="--workload s256 --steps 3 --warmup 1"
for cfg in "default:AIC_NOP=1" "nofb:AIC_TILE_FEEDBACK=0" "q1:AIC_TILE_QUEUES=1" "q1nofb:AIC_TILE_QUEUES=1 AIC_TILE_FEEDBACK=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  BENCH="python bench.py $X --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 --no-pipeline"
  rm -rf $O/w_$name
  env $envs rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR --output-format csv -d $O/w_$name -- $BENCH > /dev/null 2>&1
  python - <<PY
import csv, glob
def mean(d, counter):
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'trace_image_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter: vals.append(float(r['Counter_Value']))
    return vals
w = mean('$O/w_$name', 'WRITE_SIZE'); i = mean('$O/w_$name', 'SQ_INSTS_VMEM_WR')
print('$name', 'WRITE_SIZE per launch MB', [round(v * 1024 / 1e6, 1) for v in w], 'VMEM_WR M', [round(v / 1e6, 2) for v in i])
PY
done
find $O -type f -size +1M -delete
