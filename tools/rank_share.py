"""How long does ONE rank's share of the frame take on one GPU? (estimates the N-GPU trace rate without N GPUs)
usage: python tools/rank_share.py [n_parts] [in_flight] [workload] [frames_per_launch] [strip_rows]
frames_per_launch k > 1 (1, 2, 4, 8): the rank's shares of k consecutive frames are traced by ONE launch (aic_render_submit_batch), `in_flight` such
launches overlapping; the figure printed is still ms per FRAME (share)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench, oracle
from all_is_cubes_amd import abi
n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = sys.argv[3] if len(sys.argv) > 3 else "atrium"
per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
from all_is_cubes_amd import distributed as D
strip = int(sys.argv[5]) if len(sys.argv) > 5 else D.STRIP_ROWS  # rows per strip
sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
_, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
with abi.Context(0) as ctx:
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(view_distance=vd))
    for part in (0, n_parts // 2):
        fr = ctx.make_frame(w, h, world_inv=inv, partition=(strip, n_parts, part))
        rows = ctx.partition_rows(h, (strip, n_parts, part))
        bufs = [[torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda") for _ in range(per_launch)] for _ in range(depth)]
        def run(n):  # n launches of per_launch frames each
            fl = []
            kms = []
            for i in range(n):
                if len(fl) == depth:
                    kms.append(ctx.render_wait(fl.pop(0)).kernel_ms)
                if per_launch == 1:
                    ctx.render_submit(fr, bufs[i % depth][0].data_ptr(), i % depth)
                else:
                    ctx.render_submit_batch([fr] * per_launch, [b.data_ptr() for b in bufs[i % depth]], i % depth)
                fl.append(i % depth)
            while fl:
                kms.append(ctx.render_wait(fl.pop(0)).kernel_ms)
            return kms
        run(8)
        n = max(8, 100 // per_launch)
        t = time.perf_counter(); k = run(n); dt = time.perf_counter() - t
        ref = ctx.render(fr)
        same = all(bool((b.cpu().numpy() == ref["rgba8"]).all()) for bb in bufs for b in bb)
        print(f"{wl}: part {part}/{n_parts} ({rows} rows), {depth} launches in flight x {per_launch} frame(s) per launch: {dt / (n * per_launch) * 1e3:.4f} ms/frame, "
              f"mean kernel {np.mean(k):.4f} ms per launch; shares equal the share traced alone: {same}")
