#!/bin/bash
# Round 4, batch 11: session kernel -- the failing test's traceback, and the host laps of a session (-DAIC_LIGHT_TIMING).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b11; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_light_update.py -x -q -k beside_frames 2>&1 | grep -v "^$" | tail -40
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_lighttiming.so all_is_cubes_amd/libaic_hip.so
for env in "" "AIC_LIGHT_NO_SESSION=1"; do
  echo "== light-bench $env"
  env $env timeout 300 python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2>&1 >/dev/null | grep "light host us\|light timing"
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
