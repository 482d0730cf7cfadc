"""How much of a frame is the tile dispenser? Renders the atrium stand-in from its usual eye towards a chosen target (default: straight
up into the sky -- every ray leaves the space after a few steps, so a wave asks for a new tile as fast as it can) one frame at a
time and prints the kernel's duration. Run under AIC_TILE_QUEUES=1 (one counter for the chip) and the default (one per XCD).
usage: python tools/dispenser_probe.py [width height] [tx ty tz] [ex ey ez]   (an eye above the space looking up: rays that never enter it)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from all_is_cubes_amd import _host as H, space_from_flat

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sp, _, eye, _, vd, _ = bench.build_workload("atrium")
target = tuple(float(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (eye[0], eye[1] + 100.0, eye[2] - 1.0)
if len(sys.argv) > 8:
    eye = tuple(float(v) for v in sys.argv[6:9])
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
ms, steps = [], 0
for k in range(12):
    rend = r.draw_rgba("")
    ms.append(rend.info.kernel_ms)
    steps = int(rend.info.cubes_traced)
print(f"{w}x{h} eye {eye} target {target} queues {os.environ.get('AIC_TILE_QUEUES', 'default')}: kernel ms median {np.median(ms[2:]):.4f} min {min(ms[2:]):.4f}; "
      f"{steps / (w * h):.2f} steps per ray; {(w // 8) * ((h + 7) // 8)} tiles -> {(w // 8) * ((h + 7) // 8) / np.median(ms[2:]) / 1e3:.1f} M tiles/s")
