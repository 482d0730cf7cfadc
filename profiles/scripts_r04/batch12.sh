#!/bin/bash
# Round 4, batch 12: the chain walk fetching CHUNK bundles at a time (1 = round 3's walk, 2, 4 = default, 8), with and without the session kernel.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py -q 2>&1 | tail -4
AIC_LIGHT_NO_SESSION=1 timeout 300 python -m pytest tests/test_gpu_light_update.py -q -k beside_frames 2>&1 | tail -2
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
lb() { env $2 timeout 300 python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2> $O/lb_$1.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lu=d['light_update']; print('$1 $2', {k:lu[k] for k in ('updates','launches','device_ms','total_ms')}, lu.get('throughput_mode',{}).get('device_ms'), lu.get('throughput_mode',{}).get('total_ms'))"; }
lb chunk4 ""; lb chunk4 AIC_LIGHT_NO_SESSION=1
for v in chunk1 chunk2 chunk8; do cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so; lb $v ""; lb $v AIC_LIGHT_NO_SESSION=1; done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
