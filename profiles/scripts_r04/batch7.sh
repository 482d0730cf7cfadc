#!/bin/bash
# Round 4, GPU batch 7: the light updater beside frames in flight (tests, relight, light-bench), trip shapes (2 passes per trip x fast steps).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest7.log 2>&1; tail -3 $O/pytest7.log )
( AIC_LIGHT_FUZZ_N=200 timeout 900 python -m pytest tests/test_gpu_light_update.py -x -q -k fuzz > $O/lfuzz7.log 2>&1; tail -2 $O/lfuzz7.log )
timeout 400 python bench.py --no-cpu-baseline --workload relight --steps 60 --warmup 5 > $O/b7_relight.json 2> $O/b7_relight.err
python -c "
import json
d=json.loads(open('$O/b7_relight.json').read().strip().splitlines()[-1]); print('relight', d['ms_per_step'], d.get('relight'))"
timeout 400 python bench.py --no-cpu-baseline --workload light-bench > $O/b7_lightbench.json 2> $O/b7_lightbench.err
python -c "
import json
d=json.loads(open('$O/b7_lightbench.json').read().strip().splitlines()[-1]); l=d['light_update']; print('light-bench', l['total_ms'], l['device_ms'], l['launches'], l['throughput_mode']['total_ms'])"
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'kernel_warm', s.get('kernel_ms_warm'))" 2>/dev/null || echo "$2 FAILED"; }
for v in default r2 r2f10 r2f12 r2f16 default; do
  [ $v = default ] && cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so || cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium s256; do timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$v /"; done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/b7_${v}_atrium.json 2> $O/b7_${v}_atrium.err; one $O/b7_${v}_atrium.json "$v atrium"
  timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/b7_${v}_s256.json 2> $O/b7_${v}_s256.err; one $O/b7_${v}_s256.json "$v s256"
done 2>&1 | tee $O/variants7.txt
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
