"""Where a cold frame's time goes with and without the cost probe (profiles/r06_experiments.txt B): renders warm and cold (AIC_FRAME_NO_FEEDBACK)
frames of a bench workload one at a time; run under `rocprofv3 --kernel-trace --output-format csv -d <dir>` and then with `--parse <dir>` to
print the per-kernel durations, the probe launch (a grid of <= 128 workgroups) apart from the frame's.
usage: python tools/cold_probe_timing.py [workload] [frames]      |      python tools/cold_probe_timing.py --parse <dir>"""
import csv, glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        name = "trace" if "trace_image_kernel" in name else ("order_tiles" if "order_tiles" in name else name[:40])
        key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid), v in sorted(groups.items()):
        print(f"{name:14s} workgroups {grid:5d}  launches {len(v):4d}  median {np.median(v):9.1f} us  min {min(v):9.1f}  max {max(v):9.1f}")
    sys.exit(0)

import bench
import oracle
from all_is_cubes_amd import abi

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sp, (w, h), eye, target, vd, _ = bench.build_workload(wl)
_, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
import torch
buf = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
with abi.Context(0) as c:
    c.upload_space(abi.LAYER_WORLD, sp)
    c.set_options(abi.LAYER_WORLD, abi.make_options(bloom_intensity=0.0, view_distance=vd))
    for label, flags in (("warm", 0), ("cold", abi.FRAME_NO_FEEDBACK)):
        ms = []
        for _ in range(n):
            ms.append(c.render_to_device(c.make_frame(w, h, world_inv=inv, flags=flags), buf.data_ptr()).kernel_ms)
        print(wl, label, "kernel_ms (HIP events around probe + order + frame) median", round(float(np.median(ms[2:])), 4), "probe cap", os.environ.get("AIC_PROBE_CAP", "default"))
