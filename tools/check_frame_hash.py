"""Prints a hash of one production-variant frame of a bench workload and its step count: run under different settings
(e.g. AIC_MIGRATE_K=0 and =16, separate processes: the switch is read once) the lines must be identical.
usage: python tools/check_frame_hash.py [workload] [frames]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from all_is_cubes_amd import _host as H, space_from_flat

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
cams = H.StandardCameras()
opts = H.GraphicsOptions()
opts.bloom_intensity = 0.0
opts.view_distance = vd
opts.debug_info_text = False
cams.graphics_options = opts
cams.viewport = H.Viewport.with_scale(1.0, w, h)
cams.world_space = space_from_flat(sp)
cams.world_view_transform = H.look_at_y_up(eye, target)
r = H.HipRtRenderer(cams, None, 0)
r.update()
out = []
for k in range(frames):  # the first frame is cold (index order), the rest use the cost feedback: all must be the same image
    rend = r.draw_rgba("")
    out.append((hashlib.sha1(np.asarray(rend.data).tobytes()).hexdigest()[:16], int(rend.info.cubes_traced)))
assert len(set(out)) == 1, out
print(wl, w, h, out[0][0], out[0][1])
