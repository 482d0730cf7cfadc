#!/bin/bash
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head
rocm-smi --showperflevel --showpower 2>&1 | grep -v "^=\|^$" | head
for n in 30 300 2000; do
python bench.py --steps $n --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steps', d['steps'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
(python bench.py --steps 3000 --warmup 3 --no-cpu-baseline >/dev/null 2>&1 &) ; sleep 6; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power" | head; sleep 8
