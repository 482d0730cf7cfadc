"""Writes a call recording (the AIC_DUMP format, all_is_cubes_amd/replay.py) of one of bench.py's own workloads: upload, options,
one frame. `bench.py --workload replay:<file>` on it must reproduce `--workload <name>` (same steps per ray): the check of the
replay path while no recording of the reference's real scenes exists (tests/golden/README.md).
usage: python tools/make_recording.py atrium /tmp/atrium_like.aic"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from all_is_cubes_amd import _host as H, abi, replay

name, path = sys.argv[1], sys.argv[2]
sp, (w, h), eye, target, vd, _ = bench.build_workload(name)
# the camera matrix exactly as the host mirror derives it (so that the replayed frame is the same frame)
o = H.GraphicsOptions()
o.bloom_intensity = 0.0
o.view_distance = vd
cam = H.Camera(o, H.Viewport.with_scale(1.0, w, h))
cam.set_view_transform(H.look_at_y_up(eye, target))
inv = np.array(cam.inverse_projection_view(), np.float64).reshape(4, 4)
with replay.DumpWriter(path) as wr:
    wr.upload_space(abi.LAYER_WORLD, sp)
    wr.set_options(abi.LAYER_WORLD, abi.make_options(view_distance=vd))
    wr.frame(abi.Context.make_frame(w, h, world_inv=inv))
print("wrote", path, os.path.getsize(path), "bytes")
