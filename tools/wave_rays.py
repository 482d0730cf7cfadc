#!/usr/bin/env python3
"""AIC_WAVE_PROF file of a -DAIC_PROFILE -DAIC_RAY_PROF build: per wave, its longest ray (duration in clocks, steps) beside its own lifetime."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
a = a[(a[:, 2] != 0) | (a[:, 0] != 0)]
u32 = 1 << 32
life = (a[:, 2] - a[:, 0]) % u32
dry = (a[:, 1] - a[:, 0]) % u32
dur = a[:, 3] & ~1023
steps = a[:, 3] & 1023
q = lambda v: " ".join(str(int(np.percentile(v, p))) for p in (0, 10, 50, 90, 100))
print(f"{label}: wave lifetime min/p10/median/p90/max {q(life)}; first saw the queue dry {q(dry)}")
print(f"{label}: the wave's longest ray: duration {q(dur)}; its steps {q(steps)}; duration / lifetime {np.median(dur / life):.3f} (median)")
o = np.argsort(-dur)[:8]
print(f"{label}: the eight longest rays: " + ", ".join(f"{int(dur[i])} clocks for {int(steps[i])} steps" for i in o))
