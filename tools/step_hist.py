"""Distribution of per-pixel step counts for a bench workload (from the device's aux records)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench, oracle
from all_is_cubes_amd import abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
_, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
with abi.Context(0) as ctx:
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(view_distance=vd))
    aux = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)["aux"]
c = aux["cubes_traced"].astype(np.int64)
print(wl, "mean", c.mean(), "pcts 50/90/99/99.9/max", [int(np.percentile(c, p)) for p in (50, 90, 99, 99.9)], c.max())
print("pixels with >=500 steps:", int((c >= 500).sum()), ">=900:", int((c >= 900).sum()))
rows = c.max(axis=1)
print("row-wise max steps (every 60th row):", rows[::60].tolist())
t = c[: h // 8 * 8, : w // 8 * 8].reshape(h // 8, 8, w // 8, 8)
tmax = t.max(axis=(1, 3)); tsum = t.sum(axis=(1, 3))
print("8x8 tiles: max-steps pcts 50/90/99/max", [int(np.percentile(tmax, p)) for p in (50, 90, 99)], tmax.max(), " sum pcts", [int(np.percentile(tsum, p)) for p in (50, 90, 99)], tsum.max())
