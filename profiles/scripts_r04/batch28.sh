#!/bin/bash
# Round 4, batch 28: frames without a usable cost record: index order inside each XCD queue against a scrambled one (AIC_STATIC_SCRAMBLE=1).
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'streamed_moving', s.get('streamed_moving_camera_ms'), 'kernel_cold', s.get('kernel_ms_cold'))" 2>/dev/null || echo "$2 FAILED"; }
O=gpurun_out/r04b28; mkdir -p $O
for cfg in "index:AIC_NOP=1" "scramble:AIC_STATIC_SCRAMBLE=1" "index:AIC_NOP=1" "scramble:AIC_STATIC_SCRAMBLE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for wl in atrium s256; do env $envs timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$name /"; done | grep -v "d876fd8fde00ef83 74966856\|7912c59103550713 734379842"
  env $envs timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${name}_atrium.json 2>/dev/null; one $O/${name}_atrium.json "$name atrium"
  env $envs timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${name}_s256.json 2>/dev/null; one $O/${name}_s256.json "$name s256"
done
