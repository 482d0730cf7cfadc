"""Cross-batch speculation in the light updater: hit rates on light_bench_space in the reference's order (tools/light_spec/light_spec.cpp). CPU only.
    python tools/light_spec/run.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402  (the Space struct the oracle's C API takes)
from all_is_cubes_amd import workloads  # noqa: E402

so = os.path.join(tempfile.gettempdir(), "liblight_spec.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(HERE, "light_spec.cpp")])
lib = C.CDLL(so)
sp = workloads.light_bench_space()
sp.light[...] = 0
osp = oracle.Space(sp)
print("light_bench_space 54x16x54, fast_evaluate_light + evaluate_light(1), batches of 32 in the reference's (hashbrown) order")
print("depth = batches computed per launch (the batch at hand + 32 x (depth - 1) queue entries ahead, against the volume before the batch is applied)")
for depth in (1, 2, 4, 8):
    out = np.zeros(8)
    lib.light_spec_stats(C.byref(osp.c), C.c_int32(30), C.c_int32(1), C.c_int32(32), C.c_int32(16), C.c_int32(depth), out.ctypes.data_as(C.c_void_p))
    b, n, w, h, whole, spec, nearly = out[:7]
    print(f"depth {depth}: {int(b)} batches, {int(n)} updates; in the speculated window {w / n:6.1%}, usable (read set clean) {h / n:6.1%}; batches served whole "
          f"{whole / b:6.1%} (all but <= 2 cubes: {nearly / b:6.1%}); speculative computations per update {spec / n:.2f}")
