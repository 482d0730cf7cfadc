#!/bin/bash
# Round 3, GPU batch 10: frames without tile feedback take their macro tiles in a golden-ratio stride (AIC_SCATTER_TILES=0: index order).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b11; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --min-seconds 1"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'warm/cold/moving', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'), sf.get('single_frame_moving_camera_ms'), 'streamed moving', sf.get('streamed_moving_camera_ms'), 'kernel warm/cold', sf.get('kernel_ms_warm'), sf.get('kernel_ms_cold'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
bench3() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium"
  timeout 200 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256"
}
timeout 120 python tools/check_frame_hash.py atrium 1 2>&1 | tail -1
bench3 scatter
AIC_SCATTER_TILES=0 bench3 index
bench3 scatter2
AIC_SCATTER_TILES=0 bench3 index2
