#!/bin/bash
# Round 4, batch 19: does waiting on the frame's event (instead of the whole stream) cost the exchange loop anything? (no profiler attached)
cd ${GRAFT_REPO_ROOT:-/root/repo}
g() { env $2 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-secondary --gather-at-one $3 --steps 30 --warmup 3 --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
  g "event device-handoff" "" ""; g "event host-handoff" "" "--host-handoff"
  g "stream device-handoff" "AIC_WAIT_WHOLE_STREAM=1" ""; g "stream host-handoff" "AIC_WAIT_WHOLE_STREAM=1" "--host-handoff"
done
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['single_frame']; print('event', d['ms_per_step'], s['single_frame_warm_ms'], s['single_frame_cold_ms'], s['single_frame_moving_camera_ms'])"
  AIC_WAIT_WHOLE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['single_frame']; print('stream', d['ms_per_step'], s['single_frame_warm_ms'], s['single_frame_cold_ms'], s['single_frame_moving_camera_ms'])"
done
