#!/bin/bash
# Streamed frames on a fraction of the chip each (AIC_TILES_PER_WAVE sizes a streamed frame's grid: 1080p has 7.9 tiles per wave of the resident grid,
# 4K 31.6), with 4 / 8 frames in flight: do several part-grid launches side by side stream better than full-grid launches one behind the other?
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-secondary --no-extras --steps 40 --warmup 5"
one() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || echo "$1 FAILED"; }
for fl in 4 8; do for t in 0 12 16 24 32 48; do AIC_TILES_PER_WAVE=$t timeout 200 $B --in-flight $fl 2>/dev/null | one "atrium in-flight $fl tiles/wave $t"; done; done
for fl in 4 8; do for t in 0 64 128; do AIC_TILES_PER_WAVE=$t timeout 300 $B --workload s256 --steps 8 --warmup 2 --in-flight $fl 2>/dev/null | one "s256 in-flight $fl tiles/wave $t"; done; done
