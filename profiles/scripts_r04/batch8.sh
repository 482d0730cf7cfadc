#!/bin/bash
# Round 4, GPU batch 8: the parity suite on the new trip shape (2 passes x 16 fast steps), a sweep around it, profile counters.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest8.log 2>&1; tail -3 $O/pytest8.log )
( AIC_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz8.log 2>&1; tail -2 $O/fuzz8.log )
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
one() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
print('$2', 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'moving', s.get('single_frame_moving_camera_ms'), 'kernel_warm', s.get('kernel_ms_warm'))" 2>/dev/null || echo "$2 FAILED"; }
for v in default r2f20 r2f24 r1f16 r1f24 r3f16 fm12 fm20 tb28 default; do
  [ $v = default ] && cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so || cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  for wl in atrium s256; do timeout 200 python tools/check_frame_hash.py $wl 2 2>&1 | tail -1 | sed "s/^/$v /"; done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 > $O/b8_${v}_atrium.json 2> $O/b8_${v}_atrium.err; one $O/b8_${v}_atrium.json "$v atrium"
  timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/b8_${v}_s256.json 2> $O/b8_${v}_s256.err; one $O/b8_${v}_s256.json "$v s256"
done 2>&1 | grep -v "d876fd8fde00ef83 74966856\|7912c59103550713 734379842" | tee $O/variants8.txt
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
