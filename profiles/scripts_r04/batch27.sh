#!/bin/bash
# Round 4, batch 27: the walk's hot job fields held in vector registers (no scalar re-loads from the kernel-argument segment inside the loop).
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py -q 2>&1 | tail -1
AIC_LIGHT_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_light_update.py -q -k fuzz 2>&1 | tail -1
for i in 1 2; do python bench.py --workload light-bench --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lu=d['light_update']; print('lightbench', lu['total_ms'], lu['device_ms'], lu['throughput_mode']['total_ms'], lu['throughput_mode']['device_ms'])"; done
python bench.py --workload relight --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('relight', d['ms_per_step'], d['relight']['light_ms_per_frame'])"
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_lighttiming.so all_is_cubes_amd/libaic_hip.so
python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2>&1 >/dev/null | grep "light host us\|light timing" | head -2
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
