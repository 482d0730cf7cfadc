#!/bin/bash
# Round 4, batch 14: XCD-local tile queues (one dispenser per XCD, macro tiles dealt by super-block) against the single dispenser.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b14; mkdir -p $O
tools/ubench/xcc_id | cut -c1-300
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'kernel_warm', s.get('kernel_ms_warm'))" 2>/dev/null || echo "$2 FAILED"; }
run() {  # name, env
  for wl in atrium s256; do env $2 timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$1 /"; done
  env $2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${1}_atrium.json 2> $O/${1}_atrium.err; one $O/${1}_atrium.json "$1 atrium"
  env $2 timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${1}_s256.json 2> $O/${1}_s256.err; one $O/${1}_s256.json "$1 s256"
}
run q1 "AIC_TILE_QUEUES=1"
run q8 "AIC_TILE_QUEUES=8"
for s in 0 1 2 4 5 6; do run q8s$s "AIC_SUPER_SHIFT=$s"; done
run q1b "AIC_TILE_QUEUES=1"
