#!/bin/bash
# The experiment round 4 built and could not measure (csrc/aic_trace.hip AIC_HURRY_STEPS): waves serve rays that are far along ahead of their batching.
# Here (no GPU needed):  AIC_PATCH=profiles/scripts_r04/experiments_r01_r04.patch tools/build_variants.sh "hurry96:-DAIC_HURRY_STEPS=96" "hurry128:-DAIC_HURRY_STEPS=128" "hurry160:-DAIC_HURRY_STEPS=160" "hurry400:-DAIC_HURRY_STEPS=400" "hurry700:-DAIC_HURRY_STEPS=700"
# On the GPU box:        gpurun --timeout 900 -- 'bash profiles/scripts_r04/exp_hurry.sh'
# Frames must keep their hashes (scheduling only); what to read: one frame warm / cold against the default (C2's longest rays are 154-205 steps, C3's reach the
# 1000-step cap: 96-160 suit C2, 400-700 C3), and that the streamed figure does not pay for it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/hurry; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'kernel_warm', s.get('kernel_ms_warm'))" 2>/dev/null || echo "$2 FAILED"; }
for v in default hurry96 hurry128 hurry160 hurry400 hurry700 default; do
  if [ $v = default ]; then cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so; elif [ -f variants/libaic_hip_$v.so ]; then cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so; else continue; fi
  for wl in atrium s256; do timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$v /"; done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/${v}_atrium.json 2>/dev/null; one $O/${v}_atrium.json "$v atrium"
  timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/${v}_s256.json 2>/dev/null; one $O/${v}_s256.json "$v s256"
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
