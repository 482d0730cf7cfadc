"""Context-lifecycle stress: N x (create a context, upload a small scene, submit 1..8 frames across render slots -- sometimes past the
eighth --, wait for some, none or all of them, sometimes run the light updater beside them, destroy). Run under `python -X faulthandler`; every iteration's
plan is appended to a log that is flushed BEFORE the iteration runs, so that a fatal exit leaves its signal (faulthandler's dump on stderr) and its place
(the log's last line) behind -- round 4 lost both to `| tail -1`.
usage: python -X faulthandler tools/lifecycle_stress.py [iterations] [log file]"""
import faulthandler
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
faulthandler.enable(all_threads=True)
import numpy as np
import torch

import oracle
from all_is_cubes_amd import abi, workloads

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
log_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/lifecycle_stress.log"
os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
rng = random.Random(20250927)
sp = workloads.synthetic_space(n=16, resolution=4, n_blocks=4, seed=3)
w, h = 96, 64
_, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up((8.5, 12.5, 24.0), (8.0, 4.0, 8.0)), (8.5, 12.5, 24.0))
bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(12)]
ref = None
t0 = time.time()
with open(log_path, "w") as log:
    for it in range(n_iter):
        n_frames = rng.randint(1, 8)
        slots = rng.sample(range(12 if rng.random() < 0.2 else 8), n_frames)
        wait = rng.choice(("all", "none", "some"))
        light = rng.random() < 0.1
        log.write(f"{it} frames={n_frames} slots={slots} wait={wait} light={int(light)}\n")
        log.flush()
        os.fsync(log.fileno())
        ctx = abi.Context(0)
        try:
            ctx.upload_space(abi.LAYER_WORLD, sp)
            ctx.set_options(abi.LAYER_WORLD, abi.make_options(view_distance=200.0))
            fr = ctx.make_frame(w, h, world_inv=inv)
            for k, s in enumerate(slots):
                ctx.render_submit(fr, bufs[s].data_ptr(), s)
            if light:
                ctx.evaluate_light(abi.LAYER_WORLD, 8, True, 1, 64)
            if wait == "all":
                for s in slots:
                    ctx.render_wait(s)
                got = bufs[slots[0]].cpu().numpy()
                if ref is None:
                    ref = got.copy()
                elif not light:
                    assert (got == ref).all(), f"iteration {it}: frame differs"
            elif wait == "some":
                for s in slots[::2]:
                    ctx.render_wait(s)
        finally:
            ctx.close()
        if it % 2000 == 0:
            print(f"iteration {it} ({time.time() - t0:.0f} s)", flush=True)
torch.cuda.synchronize()
print(f"lifecycle stress: {n_iter} iterations in {time.time() - t0:.0f} s, no fault")
