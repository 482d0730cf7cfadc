"""How long does ONE rank's share of the frame take on one GPU? (estimates the N-GPU trace rate without N GPUs)
usage: python tools/rank_share.py [n_parts] [in_flight] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench, oracle
from all_is_cubes_amd import abi
n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = sys.argv[3] if len(sys.argv) > 3 else "atrium"
sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
_, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
with abi.Context(0) as ctx:
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(view_distance=vd))
    for part in (0, n_parts // 2):
        fr = ctx.make_frame(w, h, world_inv=inv, partition=(16, n_parts, part))
        rows = ctx.partition_rows(h, (16, n_parts, part))
        bufs = [torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        def run(n):
            fl = []
            kms = []
            for i in range(n):
                if len(fl) == depth:
                    kms.append(ctx.render_wait(fl.pop(0)).kernel_ms)
                ctx.render_submit(fr, bufs[i % depth].data_ptr(), i % depth)
                fl.append(i % depth)
            while fl:
                kms.append(ctx.render_wait(fl.pop(0)).kernel_ms)
            return kms
        run(8)
        t = time.perf_counter(); k = run(100); dt = time.perf_counter() - t
        print(f"{wl}: part {part}/{n_parts} ({rows} rows), {depth} in flight: {dt / 100 * 1e3:.4f} ms/frame, mean kernel {np.mean(k):.4f} ms")
