#!/usr/bin/env python3
"""Summarise an AIC_WAVE_PROF file (-DAIC_PROFILE build): per wave start, first-saw-the-queue-dry and end clocks, pixels taken.
    python tools/wave_tail.py <file> [label]"""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1], dtype=np.int64)
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
a = a[(a[:, 2] != 0) | (a[:, 0] != 0)]
start, dry, end, px = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
u32 = 1 << 32
# (each XCD has its own clock: only differences within a wave mean anything; all resident waves start together)
d, e = (dry - start) % u32, (end - start) % u32
q = lambda v: " ".join(str(int(np.percentile(v, p))) for p in (0, 10, 50, 90, 100))
print(f"{label}: waves recorded {len(a)}")
print(f"{label}: first saw the queue dry after  min/p10/median/p90/max {q(d)}")
print(f"{label}: ended after                    min/p10/median/p90/max {q(e)}")
print(f"{label}: tail (end - dry)               min/p10/median/p90/max {q(e - d)}")
print(f"{label}: longest wave {int(e.max())} cycles; median dry / longest = {np.median(d) / e.max():.3f}; wave-time after dry / all wave-time = {(e - d).sum() / e.sum():.3f}; pixels per wave min/median/max {int(px.min())} {int(np.median(px))} {int(px.max())}")
