#!/bin/bash
# quick experiment loop on the GPU box: parity tests, then the headline bench under a few settings
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for sm in 0 1 2 3 7; do
for wl in atrium s256; do
echo "== AIC_SKIP_MAX=$sm $wl"
AIC_SKIP_MAX=$sm python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['gsteps_per_s'], d['config']['steps_per_ray'])"
done; done
