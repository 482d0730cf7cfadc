#!/bin/bash
# the sim + render loops (configs[4]): orbit (light volume re-uploaded per frame) and relight (light computed on the device)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06loops}; mkdir -p $O
for wl in orbit relight; do
  for k in 1 2; do
    timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_$wl$k.json 2> $O/bench_$wl$k.err
    python -c "import json; d=json.loads(open('$O/bench_$wl$k.json').readlines()[-1]); print('$wl', d['ms_per_step'], d['value'], d.get('relight'))" || tail -3 $O/bench_$wl$k.err
  done
done
