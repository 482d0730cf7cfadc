#!/bin/bash
# Round 4, batch 13: light suites on the final light code (launch per batch by default; AIC_LIGHT_SESSION=1: the session kernel), the light file on its own
# (the library loaded before torch used to leave the process with two HIP runtimes), light fuzz both ways.
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_light_update.py -q 2>&1 | tail -3
AIC_LIGHT_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_light_update.py -q -k fuzz 2>&1 | tail -2
AIC_LIGHT_SESSION=1 AIC_LIGHT_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py tests/test_gpu_goldens2.py -q 2>&1 | tail -3
AIC_LIGHT_SESSION=1 timeout 300 python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lu=d['light_update']; print('session', {k:lu[k] for k in ('updates','launches','device_ms','total_ms')})"
