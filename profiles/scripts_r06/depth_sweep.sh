#!/bin/bash
# the streamed loop at the driver's --steps 20 --warmup 5 against the frames in flight (and a long region for the steady state)
for d in 2 3 4 5 6 8; do
  for s in 20 200; do
    python bench.py --steps $s --warmup 5 --in-flight $d --no-cpu-baseline --no-extras --no-secondary --min-seconds 1.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $d steps $s: ms_per_step', d['ms_per_step'], 'min', d.get('ms_per_step_min'), 'max', d.get('ms_per_step_max'), 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done
for fpl in 2 4; do
  for d in 8 16; do
    python bench.py --steps 20 --warmup 5 --frames-per-launch $fpl --in-flight $d --no-cpu-baseline --no-extras --no-secondary --min-seconds 1.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-launch $fpl in-flight $d steps 20: ms_per_step', d['ms_per_step'], 'min', d.get('ms_per_step_min'))"
  done
done
