#!/bin/bash
# Round 4, batch 21: issue priority (s_setprio) for waves that host rays far along -- does the frame's critical path (its longest rays) get shorter?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b21; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'streamed_moving', s.get('streamed_moving_camera_ms'), 'kernel_warm', s.get('kernel_ms_warm'))" 2>/dev/null || echo "$2 FAILED"; }
for v in default prio5 prio6 prio7 prio8 default; do
  [ $v = default ] && cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so || cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium s256; do timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$v /"; done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${v}_atrium.json 2> $O/${v}_atrium.err; one $O/${v}_atrium.json "$v atrium"
  timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${v}_s256.json 2> $O/${v}_s256.err; one $O/${v}_s256.json "$v s256"
done 2>&1 | grep -v "d876fd8fde00ef83 74966856\|7912c59103550713 734379842"
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
