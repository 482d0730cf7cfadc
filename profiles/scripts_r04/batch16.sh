#!/bin/bash
# Round 4, batch 16: with XCD-local queues in place -- is the cost-feedback order still worth it, and which macro tile?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b16; mkdir -p $O
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'streamed_moving', s.get('streamed_moving_camera_ms'))" 2>/dev/null || echo "$2 FAILED"; }
run() {  # name, env
  env $2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${1}_atrium.json 2> $O/${1}_atrium.err; one $O/${1}_atrium.json "$1 atrium"
  env $2 timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${1}_s256.json 2> $O/${1}_s256.err; one $O/${1}_s256.json "$1 s256"
}
run default ""
run nofb "AIC_TILE_FEEDBACK=0"
run q1 "AIC_TILE_QUEUES=1"
run q1nofb "AIC_TILE_QUEUES=1 AIC_TILE_FEEDBACK=0"
for m in 1 4 8; do run macro$m "AIC_MACRO=$m"; run macro${m}nofb "AIC_MACRO=$m AIC_TILE_FEEDBACK=0"; done
timeout 300 python bench.py --workload orbit --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orbit', d['ms_per_step'])"
AIC_TILE_QUEUES=1 timeout 300 python bench.py --workload orbit --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orbit q1', d['ms_per_step'])"
