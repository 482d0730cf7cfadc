#!/bin/bash
# Round 4, batch 17: knobs that were swept while one tile counter capped a 1080p frame at 0.42 ms, swept again with the XCD-local queues.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b17; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'streamed_moving', s.get('streamed_moving_camera_ms'))" 2>/dev/null || echo "$2 FAILED"; }
for v in default r3f8 r2f8 r3f16 r2f24 spec4 shadestep tb24 tb40 default; do
  [ $v = default ] && cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so || cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium; do timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$v /"; done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${v}_atrium.json 2> $O/${v}_atrium.err; one $O/${v}_atrium.json "$v atrium"
  timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${v}_s256.json 2> $O/${v}_s256.err; one $O/${v}_s256.json "$v s256"
done 2>&1 | grep -v "d876fd8fde00ef83 74966856"
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
for d in 2 3 6 8; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-extras --steps 30 --warmup 3 --in-flight $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $d atrium', d['ms_per_step'])"; done
