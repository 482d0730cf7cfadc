#!/bin/bash
# Round 4, batch 10: the light updater's session kernel (one launch per call for small batches) -- parity, then light-bench both ways.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py -x -q 2>&1 | tail -8
for env in "" "AIC_LIGHT_NO_SESSION=1"; do
  echo "== light-bench $env"
  env $env timeout 300 python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2> $O/lb$env.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lu=d['light_update']; print({k:lu[k] for k in ('updates','launches','device_ms','total_ms')}, lu.get('throughput_mode',{}).get('total_ms'))"
done
