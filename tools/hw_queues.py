#!/usr/bin/env python3
"""Streamed frames against the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES), as a caller of the C ABI sees them.

bench.py sets GPU_MAX_HW_QUEUES=8 for itself before the runtime starts; a host that links libaic_hip.so (the Rust shim, a C
program) sets nothing. The runtime deals HIP streams onto that many hardware queues, and streams that share a queue run their
kernels one behind the other -- which is the whole of what aic_render_submit's slots are for. This tool runs the streamed loop of
bench.py's secondary leg (the host mirror's HipRtRenderer, `depth` frames in flight) in one child process per setting:

  runtime   the variable unset and the library told to leave it alone (AIC_KEEP_HW_QUEUES=1): the runtime's own default
  library   the variable unset: what libaic_hip.so's load-time default makes of it (csrc/aic_abi.cpp aic_default_hw_queues)
  2 4 8 16  set by the caller

with HIP buffers from hipMalloc (a caller without torch) or, with --torch, torch imported first and its buffers used.

    python tools/hw_queues.py [--workload atrium] [--frames 400] [--depth 4] [--torch]
"""
from __future__ import annotations

import argparse
import ctypes
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def child(args) -> None:
    keep = os.environ.get("GPU_MAX_HW_QUEUES")
    import bench  # its import sets a default for itself; put the variable back as this child was given it (nothing has touched HIP yet)
    if keep is None:
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = keep
    torch = None
    if args.torch:
        import torch
        torch.zeros(1, device="cuda:0")
    import all_is_cubes_amd as A
    from all_is_cubes_amd import distributed as D
    H = A.host
    flat_space, (w, h), eye, target, view_distance, _ = bench.build_workload(args.workload)
    cams = H.StandardCameras()
    opts = H.GraphicsOptions()
    opts.bloom_intensity = 0.0
    opts.view_distance = view_distance
    opts.debug_info_text = False
    cams.graphics_options = opts
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    cams.world_space = A.space_from_flat(flat_space)
    cams.world_view_transform = H.look_at_y_up(eye, target)
    r = H.HipRtRenderer(cams, None, 0)
    r.update()
    depth = args.depth
    if torch is not None:
        bufs = [torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(depth)]
        ptrs = [b.data_ptr() for b in bufs]
    else:
        # the HIP runtime this process has mapped already (libaic_hip.so's: abi.load keeps it to one), not a second copy
        paths = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
        assert len(paths) == 1, paths
        hip = ctypes.CDLL(paths[0])
        ptrs = []
        for _ in range(depth):
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(w * h * 4)) == 0
            ptrs.append(p.value)

    def streamed(n):
        for i in range(n):
            if i >= depth:
                r.wait_rows(i % depth)
            r.submit_rows_to_device(ptrs[i % depth], D.STRIP_ROWS, 1, 0, i % depth)
        for i in range(max(0, n - depth), n):
            r.wait_rows(i % depth)
        r.synchronize()

    streamed(3 * depth)
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        streamed(args.frames)
        out.append((time.perf_counter() - t0) / args.frames * 1e3)
    alone = []
    for _ in range(12):
        t0 = time.perf_counter()
        r.draw_rows_to_device(ptrs[0], D.STRIP_ROWS, 1, 0)
        alone.append((time.perf_counter() - t0) * 1e3)
    alone.sort()
    print(f"streamed ms/frame {min(out):.4f} (runs {' '.join(f'{v:.4f}' for v in out)}); one frame alone {alone[len(alone) // 2]:.4f}", flush=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="atrium")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--torch", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--settings", default="runtime,library,2,4,8,16")
    args = ap.parse_args()
    if args.child:
        child(args)
        return 0
    for setting in args.settings.split(","):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        env.pop("AIC_KEEP_HW_QUEUES", None)
        if setting == "runtime":
            env["AIC_KEEP_HW_QUEUES"] = "1"
        elif setting != "library":
            env["GPU_MAX_HW_QUEUES"] = setting
        cmd = [sys.executable, __file__, "--child", "--workload", args.workload, "--frames", str(args.frames), "--depth", str(args.depth)]
        if args.torch:
            cmd.append("--torch")
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        tail = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else f"rc {p.returncode}: {p.stderr.strip()[-300:]}"
        print(f"{args.workload} depth {args.depth} {'torch' if args.torch else 'hipMalloc'} GPU_MAX_HW_QUEUES={setting}: {tail}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
