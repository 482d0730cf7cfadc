#!/bin/bash
# What does the GPU clock at while the light updater's chain of 32-cube batches runs (32 of 1024 SIMDs busy), and while frames stream (all busy)?
# rocm-smi sampled from a second shell a few seconds into each loop. usage (GPU box): bash tools/light_clock_probe.sh
cd "$(dirname "$0")/.."
python - <<'PY' &
import time, numpy as np
from all_is_cubes_amd import abi, workloads as scenes
sp = scenes.light_bench_space(); sp.light[...] = 0
c = abi.Context(0)
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    c.upload_space(abi.LAYER_WORLD, sp)
    info = c.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=32, queue_order=16)
    n += 1
print("light loop:", n, "runs of", info.updates, "updates;", round((time.time() - t0) / n * 1e3, 1), "ms each, device", round(info.device_ms, 1))
c.close()
PY
sleep 7; echo "-- during the light chain:"; rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4; rocm-smi --showpower 2>&1 | grep -i "power" | head -2; wait
(python bench.py --steps 3000 --warmup 3 --no-cpu-baseline --no-extras --no-secondary > /dev/null 2>&1 &); sleep 9; echo "-- while frames stream:"; rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4; rocm-smi --showpower 2>&1 | grep -i "power" | head -2; sleep 6
