#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X voxel raytracer.

Metric (BASELINE.json): Mrays/s (+ frames/s) at 1920x1080. One "step" = one frame of the hot
path over the scene already resident in HBM: camera upload, (tile ordering,) trace kernel, and --
for N > 1 -- the RCCL gather of the row strips to rank 0 plus the de-interleave. Frames are
streamed the way a recording loop submits them: 4 traces in flight (8 from N = 4), a
frame's gather running under the next frames' traces; every frame issued in the timed region is
complete, gathered and assembled before the clock stops; the K-step region is repeated until the
regions add up to `--min-seconds` and the median region is reported (min / max beside it).
`--no-pipeline` traces one frame at a time; the `single_frame` object reports one frame alone
(warm / cold / moving camera). The finished RGBA8 frames stay in HBM (the PCIe-inclusive rate is
reported separately as `fps_with_readback`).

Workload (config.workload), BASELINE.json configs[1]: 1920x1080 single-frame raytrace of the
Atrium scene at block resolution 16. The reference's Atrium generator needs the un-vendored
noise crate and the block-evaluation engine, so the stand-in is
`atrium_like_space` (19x35x51 cubes, R16 recursive blocks, a light field, the Atrium spawn
camera), with GraphicsOptions::default() minus bloom (Volumetric transparency, Linear lighting,
Abrupt fog). `--workload s256` selects configs[2] (3840x2160 synthetic 256^3 Space, R32),
`--workload orbit` configs[4] (60-frame orbit, light volume re-uploaded every frame),
`--workload relight` the same loop with the light computed ON THE DEVICE (a lamp toggled, 1024 cube
updates of the light updater and one frame per step), `--workload light-bench` the reference's own
bench scene (all-is-cubes-render/benches/raytrace.rs: light_bench_space, 64x64), lit on the device,
`--workload replay:<file.aic>` a scene recorded through the C ABI (the reference's real Atrium / DemoCity once captured with
rust/all-is-cubes-hip/examples/capture.rs: tests/golden/README.md) with its recorded camera and options.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time
from pathlib import Path

# HIP deals streams onto hardware queues in creation order, 4 by default; with four frame streams,
# an upload stream, torch's and RCCL's own, two frame slots could end up sharing a queue and
# serialise their kernels. Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
# tools/ubench/issue_rate.hip on an MI355X (profiles/r03_issue_rate.txt): SIMD cycles per wave-instruction, launch-time based
ISSUE_PEAK_CYCLES = 2.2   # the cheapest stream there is (v_mov_b32 / v_add_u32, 8 waves per SIMD)
ISSUE_PIPE_CYCLES = 4.2   # f64 add / fma / compare, v_cndmask, v_mul_lo_u32 and every SALU instruction


def build_workload(name: str):
    from all_is_cubes_amd import workloads as scenes

    if name == "atrium":
        space = scenes.atrium_like_space()
        size = (1920, 1080)
        eye, target = (0.5, 9.91, 10.0), (0.5, 8.0, -20.0)  # atrium spawn eye (content atrium/mod.rs:98-102)
        view_distance = 200.0
        label = "atrium-like 19x35x51 R16, 1920x1080, GraphicsOptions::default() minus bloom"
    elif name == "orbit":
        # BASELINE.json configs[4]: 1080p 60-frame camera orbit with a per-frame light re-upload
        space = scenes.atrium_like_space()
        size = (1920, 1080)
        eye, target = (0.5, 9.91, 10.0), (0.5, 8.0, 0.0)
        view_distance = 200.0
        label = "atrium-like 19x35x51 R16, 1920x1080, 60-frame orbit, light volume re-uploaded + camera moved every frame"
    elif name == "relight":
        # the sim+render loop with the light ON THE DEVICE (SURVEY.md 8f N1 + N2): every frame a lamp block is placed or
        # removed, the changed cube and its neighbours are queued (aic_light_cubes_changed), the light updater gets a budget
        # of cube updates (aic_evaluate_light, continuing the layer's queue), the camera moves, and the frame is traced.
        # No light volume crosses PCIe.
        from all_is_cubes_amd import flat
        space = scenes.atrium_like_space()
        space.add_block(flat.atom((1.0, 0.9, 0.7, 1.0), (8.0, 7.0, 5.0), name="lamp"))
        space.light[...] = 0
        size = (1920, 1080)
        eye, target = (0.5, 9.91, 10.0), (0.5, 8.0, 0.0)
        view_distance = 200.0
        label = ("atrium-like 19x35x51 R16, 1920x1080, 60-frame orbit; a lamp block toggled every --relight-period frames, every frame one "
                 "launch of --light-budget cube updates of the device's light updater, camera moved")
    elif name == "s256":
        space = scenes.synthetic_space(n=256, resolution=32, n_blocks=64, seed=1)
        size = (3840, 2160)
        eye, target = (128.5, 140.5, 300.0), (128.0, 100.0, 128.0)
        view_distance = 600.0
        label = "synthetic S256 256^3 R32 (64 blocks), 3840x2160, GraphicsOptions::default() minus bloom, view_distance 600"
    elif name == "light-bench":
        # the reference's own benchmark scene (all-is-cubes-render/benches/raytrace.rs, all-is-cubes/benches/light.rs):
        # content::testing::light_bench_space at 54x16x54 seen from its Spawn (looking_at_space(bounds, [0, 0.5, 1])),
        # 64x64 viewport. Uploaded with its light Uninitialized: bench.py lights it ON THE DEVICE (aic_evaluate_light,
        # fast_evaluate_light + evaluate_light(1) -- light.rs's "both" mode) before tracing.
        space = scenes.light_bench_space()
        space.light[...] = 0
        size = (64, 64)
        lo, hi = np.array(space.lo, float), np.array(space.hi, float)
        d = np.array([0.0, 0.5, 1.0])
        eye_v = (lo + hi) / 2.0 + d / np.linalg.norm(d) * float((hi - lo).max())  # camera.rs:34-40 eye_for_look_at
        eye, target = tuple(eye_v), tuple(eye_v - d)
        view_distance = 200.0
        label = "light_bench_space 54x16x54 (reference bench scene), 64x64, lit on the device"
    elif name == "small":
        space = scenes.synthetic_space(n=32, resolution=8, n_blocks=8, seed=1)
        size = (320, 200)
        eye, target = (16.5, 24.5, 48.0), (16.0, 8.0, 16.0)
        view_distance = 200.0
        label = "synthetic S32 R8 320x200 (plumbing)"
    elif name.startswith("replay:"):
        # a scene recorded through the C ABI (AIC_DUMP; rust/all-is-cubes-hip/examples/capture.rs, tests/golden/README.md): the
        # world space, options and camera of the recording's LAST frame
        from all_is_cubes_amd import abi, replay as rp
        path = name.split(":", 1)[1]
        st, frame = rp.SceneState(), None
        for r in rp.read_dump(path):
            if r.tag == rp.FRAME:
                frame = r.data["frame"]
            else:
                st.apply(r)
        if frame is None or st.spaces[abi.LAYER_WORLD] is None:
            raise SystemExit(f"{path}: no frame / no world space recorded")
        if st.spaces[abi.LAYER_UI] is not None:
            print(f"{path}: the recorded UI layer is not replayed by the benchmark (world layer only)", file=sys.stderr)
        space = st.spaces[abi.LAYER_WORLD]
        size = (int(frame["width"]), int(frame["height"]))
        o = st.options[abi.LAYER_WORLD]
        view_distance = float(o["view_distance"]) if o is not None else 200.0
        eye = target = None
        label = f"recorded scene {os.path.basename(path)} ({'x'.join(str(v) for v in space.size)} cubes, {len(space.blocks)} blocks), {size[0]}x{size[1]}, recorded camera and options"
        REPLAY.update(inv=[float(v) for v in frame["world"]["inverse_projection_view"]], exposure=float(frame["world"]["exposure"]), options=o)
    else:
        raise SystemExit(f"unknown workload {name}")
    return space, size, eye, target, view_distance, label


REPLAY: dict = {}  # --workload replay:<file>: the recorded camera matrix, exposure and options


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="atrium",
                    help="atrium | s256 | small | orbit | light-bench | relight | replay:<recording.aic> (a scene captured through the C ABI: tests/golden/README.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                    help="N > 1 (default there): check the assembled frame against a single-rank trace of the same frame")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--in-flight", type=int, default=0, help="frames traced concurrently (1..32); default 4 at N < 4, 8 at N = 4, 16 at N >= 8")
    ap.add_argument("--frames-per-gather", type=int, default=0,
                    help="N > 1: consecutive frames moved to rank 0 by one collective; default 1 (a frame is gathered as soon as it is traced). More "
                         "frames per collective trade latency for fewer host calls: to be measured on a real 8-GPU node before it becomes a default "
                         "(README: the SCALE commands)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="1, 2, 4 or 8: that many consecutive frames (of a rank: its shares of them) are traced by ONE launch (aic_render_submit_batch); --in-flight "
                         "then counts frames too (at least two launches are kept in flight). Default: 1 at N < 4, 8 from N = 4 -- a rank's share is then too small "
                         "to fill the chip alone, and eight of them in one launch are a whole frame's worth of tiles (profiles/r06_rank_share.txt)")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: gather each frame before tracing the next")
    ap.add_argument("--gather-at-one", action="store_true",
                    help="N = 1: run the exchange step all the same (a one-rank process group over the nccl backend, the strip ring, the gather and the "
                         "de-interleave) -- the only way to execute the RCCL leg of the N > 1 path on a one-GPU box")
    ap.add_argument("--host-handoff", action="store_true",
                    help="N > 1: the round-3 hand-off (the host waits for a traced frame before it submits its gather) instead of the device-side one")
    ap.add_argument("--lighting", type=int, default=3, help="experiment: LightingOption (0 None,1 Flat,2 Coarse,3 Linear,4 Smoothstep); default Linear")
    ap.add_argument("--fog", type=int, default=1, help="experiment: FogOption (0 None,1 Abrupt,...); default Abrupt")
    ap.add_argument("--transparency", type=int, default=None, help="experiment: 0 Surface, 1 Volumetric; default Volumetric (light-bench: Surface, the reference bench's 'linear-surface')")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed single-frame / moving-camera / read-back measurements (counter passes)")
    ap.add_argument("--relight-period", type=int, default=30, help="relight: frames between two lamp toggles")
    ap.add_argument("--light-budget", type=int, default=2048, help="relight: cube updates of the light updater per frame (one launch)")
    ap.add_argument("--blocking-light", action="store_true", help="relight: the blocking aic_evaluate_light per frame (rounds 2-5) instead of aic_evaluate_light_submit / _wait")
    ap.add_argument("--no-secondary", action="store_true", help="atrium (the default line): skip the short s256 leg reported as `secondary`")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="repeat the K-step timed region until the regions add up to this much time")
    args = ap.parse_args()
    default_transparency = 0 if args.workload == "light-bench" else 1
    if args.transparency is None:
        args.transparency = default_transparency

    import torch
    import torch.distributed as dist

    from all_is_cubes_amd import _host as H
    from all_is_cubes_amd import distributed as D
    from all_is_cubes_amd import space_from_flat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in the product)")
    # (test hook: AIC_BENCH_ONE_GPU=1 maps every rank to GPU 0, to exercise the N > 1 path on a 1-GPU box)
    if os.environ.get("AIC_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    one_gpu_test = os.environ.get("AIC_BENCH_ONE_GPU") == "1"
    exchange = world > 1 or args.gather_at_one  # the strips travel to rank 0 through a collective
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu_test:  # RCCL refuses two ranks on one device: the test hook gathers through host memory
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    flat_space, (w, h), eye, target, view_distance, label = build_workload(args.workload)

    # --- scene upload (untimed): replicated on every GPU --------------------------------------
    cams = H.StandardCameras()
    opts = H.GraphicsOptions()  # default(): Volumetric, Linear lighting, Abrupt fog
    opts.bloom_intensity = 0.0
    opts.view_distance = view_distance
    opts.debug_info_text = False
    if args.workload == "light-bench" and (args.lighting, args.fog, args.transparency) == (3, 1, 0):
        opts.transparency = H.TransparencyOption(H.TransparencyKind(0))  # raytrace.rs "linear-surface"
        label += ", GraphicsOptions::default() + TransparencyOption::Surface ('linear-surface')"
    elif (args.lighting, args.fog, args.transparency) != (3, 1, 1):  # experiments only; the headline run uses the defaults
        opts.lighting_display = H.LightingOption(H.LightingKind(args.lighting))
        opts.fog = H.FogOption(args.fog)
        opts.transparency = H.TransparencyOption(H.TransparencyKind(args.transparency))
        label += f" [EXPERIMENT lighting={args.lighting} fog={args.fog} transparency={args.transparency}]"
    cams.graphics_options = opts
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    cams.world_space = space_from_flat(flat_space)
    is_replay = args.workload.startswith("replay:")
    if is_replay:
        ro = REPLAY["options"]
        if ro is not None:  # the recorded GraphicsOptions of the world layer
            opts.fog = H.FogOption(int(ro["fog"]))
            opts.transparency = H.TransparencyOption(H.TransparencyKind(int(ro["transparency"])), float(ro["threshold"]))
            opts.lighting_display = H.LightingOption(H.LightingKind(int(ro["lighting"])))
            opts.antialiasing = H.AntialiasingOption(int(ro["antialiasing"]))
            opts.debug_pixel_cost = bool(ro["debug_pixel_cost"])
            opts.tone_mapping = H.ToneMappingOperator(int(ro["tone_mapping"]))
            opts.maximum_intensity = float(ro["maximum_intensity"])
            cams.graphics_options = opts
        eye, target = (0.0, 0.0, 0.0), (0.0, 0.0, -1.0)  # (unused: the recorded matrix overrides the derived camera)
    cams.world_view_transform = H.look_at_y_up(eye, target)
    renderer = H.HipRtRenderer(cams, None, local_rank)
    if is_replay:
        renderer.set_world_camera_override(REPLAY["inv"], REPLAY["exposure"])
    renderer.update()
    light_update = None
    loop = None   # orbit / relight: what changes in the scene before every frame (SceneLoop)
    relight = None
    if args.workload in ("orbit", "relight"):
        loop = SceneLoop(args.workload, H, flat_space, cams, renderer, eye, target, args.relight_period, args.light_budget, args.blocking_light)
        light_update, relight = loop.light_update, loop.relight
    if args.workload == "light-bench":
        # light.rs "both": fast_evaluate_light then evaluate_light(1), LightPhysics::Rays { maximum_distance: 30 },
        # batches of 32 in the reference's queue order -- the configuration that reproduces the reference's texels
        li = renderer.evaluate_light(30, True, 1, 32, 16)
        def light_roofline(r):
            # Algorithmic bytes of the light walk: per visited ray-tree bundle 16 B of the bundle's record
            # (its only child's entry, its number of children), 2 B of the block index at its cube and 4 B of the light texel
            # there; the kernel counts the visits (aic_light_info.bundles_visited). Against the HBM peak like the trace
            # kernel's figure, and just as far from it: the walk is a chain of dependent L2 hits, bound by latency.
            b = 22 * int(r["bundles_visited"])
            return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bundles_visited": int(r["bundles_visited"]), "algorithmic_bytes": b,
                    "achieved": round(b / (r["device_ms"] * 1e-3) / 1e9, 3) if r["device_ms"] > 0 else None,
                    "frac": round(b / (r["device_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if r["device_ms"] > 0 else None,
                    "kernel": "compute_light_wave_kernel", "kernel_ms_per_launch": round(r["device_ms"] / max(1, int(r["batches"])), 4),
                    "bundles_per_cube": round(int(r["bundles_visited"]) / max(1, int(r["updates"])), 1)}

        light_update = {"mode": "fast_evaluate_light + evaluate_light(1), batch 32, hashbrown order", "updates": int(li["updates"]),
                        "launches": int(li["batches"]), "device_ms": round(li["device_ms"], 3), "total_ms": round(li["total_ms"], 3),
                        "updates_per_s": round(li["updates"] / (li["total_ms"] * 1e-3), 1) if li["total_ms"] > 0 else None,
                        "roofline": light_roofline(li)}
        # and the same work with whole-queue batches (order differs from the reference's in the last texel unit)
        renderer.update()
        lt = renderer.evaluate_light(30, True, 1, 8192, 0)
        light_update["throughput_mode"] = {"batch": 8192, "updates": int(lt["updates"]), "launches": int(lt["batches"]),
                                           "device_ms": round(lt["device_ms"], 3), "total_ms": round(lt["total_ms"], 3),
                                           "updates_per_s": round(lt["updates"] / (lt["total_ms"] * 1e-3), 1) if lt["total_ms"] > 0 else None,
                                           "roofline": light_roofline(lt)}
        li = renderer.evaluate_light(30, True, 1, 32, 16)  # leave the reference-order light in place for the traced frames

    strip = D.STRIP_ROWS
    local_rows = renderer.partition_rows(strip, world, rank)
    rays_per_frame = w * h * (4 if opts.antialiasing == H.AntialiasingOption.Always else 1)
    # Frames are streamed: two traces in flight on the device (frame i+1 starts filling the GPU while
    # frame i's last rays finish) and, for N > 1, frame i-2's RCCL gather running under them.
    # --no-pipeline: one frame at a time, gathered before the next is traced.
    streamed = not args.no_pipeline
    # traces in flight (AIC_MAX_IN_FLIGHT = 32): a rank's share of a frame shrinks with N while a ray's latency does not -- the default keeps about four
    # frames' worth of rays on each device (profiles/r05_rank_share.txt)
    per_launch = (args.frames_per_launch if args.frames_per_launch in (1, 2, 4, 8) else (1 if world < 4 else 8)) if streamed else 1  # frames traced by one launch (aic_render_submit_batch)
    # (with eight shares per launch: two launches in flight at N = 4, four from N = 8 -- what the share's time stops falling at on one GPU, profiles/r06_rank_share.txt)
    depth = max(1, min(32, args.in_flight if args.in_flight > 0 else (4 if world < 4 else ((8 if world < 8 else 16) if per_launch == 1 else (16 if world < 8 else 32)))))
    launches = max(2, depth // per_launch) if per_launch > 1 else depth  # render slots in use (a launch occupies one)
    if per_launch > 1:
        depth = launches * per_launch
    # (the frames of one launch are done together: they travel in one collective -- no frame waits for another that is not done anyway)
    per_gather = max(1, min(depth, args.frames_per_gather if args.frames_per_gather > 0 else per_launch)) if streamed else 1
    ring = ((depth + per_gather - 1) // per_gather + (1 if per_gather == 1 else 2)) if streamed else 1  # group slots: the groups being traced, one being gathered
    pipe = D.StripGatherPipeline(h, w, strip, "cpu" if one_gpu_test else dev, depth=ring,
                                 wait_event=None if one_gpu_test else renderer.wait_event, frames=per_gather) if exchange else None
    n_local = depth if streamed else 1
    local_bufs = [torch.empty((max(local_rows, 1), w, 4), dtype=torch.uint8, device=dev) for _ in range(n_local)] if (pipe is None or one_gpu_test) else None
    stage_buf = (torch.empty((world, pipe.max_rows, w, 4) if per_gather == 1 else (world, per_gather, pipe.max_rows, w, 4), dtype=torch.uint8, device=dev)
                 if (one_gpu_test and pipe is not None and rank == 0) else None)
    frame_buf = torch.empty((h, w, 4), dtype=torch.uint8, device=dev) if (rank == 0 and exchange) else None
    # Hand-off of a traced frame to its gather. Round 3 waited for the frame on the HOST (aic_render_wait: a hipStreamSynchronize per
    # frame in front of every collective -- at N = 8, where a rank's share is traced in 0.1 ms, a host round trip on the critical
    # path). Now the stream the collective is issued from waits for the frame's event ON THE DEVICE (aic_stream_wait_frame) and the
    # host goes on; the frame's report is collected just before its render slot is used again, `depth` frames later, when it has
    # long finished (that wait is the loop's back-pressure, not part of the exchange).
    device_handoff = pipe is not None and not one_gpu_test and not args.host_handoff
    uncollected = {}  # render slot -> True: its frame was handed to the gather on the device, its report is still to be read

    kernel_ms = []
    frame_no = [0]
    traced = []   # launches whose trace is in flight: (first frame number, render slot, frames)
    pending = []  # frames-per-launch > 1: frames waiting for their launch to fill up

    filled = {}  # group slot -> frames of it traced since its last gather

    def slot_of(i):  # frame i belongs to group i // per_gather, which lives in this ring slot
        return (i // per_gather) % ring

    def finish(slot) -> None:  # a gathered group of frames leaves the ring: de-interleave each on rank 0
        g = pipe.retire(slot)
        n_valid = filled.pop(slot, 0)
        if rank == 0 and g is not None:
            if one_gpu_test:
                stage_buf.copy_(g)
                torch.cuda.synchronize()
                g = stage_buf
            if per_gather == 1:
                renderer.assemble_strips(g.data_ptr(), frame_buf.data_ptr(), strip, world, not device_handoff)  # (device hand-off: enqueue only)
            else:
                for k in range(n_valid):
                    pipe.assemble(g, k, out=frame_buf)

    def render_target(i):
        if pipe is not None and not one_gpu_test:
            return pipe.frame_buffer(slot_of(i), i % per_gather)
        return local_bufs[i % n_local]

    def collect(rslot) -> None:
        if uncollected.pop(rslot, None):
            kernel_ms.append(renderer.wait_rows(rslot).kernel_ms)

    def complete_oldest() -> None:  # the oldest launch in flight: hand its frames' strips to the gather
        i0, rslot, count = traced.pop(0)
        if device_handoff:
            renderer.stream_wait_rows(rslot, torch.cuda.current_stream(dev).cuda_stream)
            uncollected[rslot] = True
        else:
            info = renderer.wait_rows(rslot)
            kernel_ms.append(info.kernel_ms)
        for i in range(i0, i0 + count):
            if pipe is not None:
                if one_gpu_test:
                    pipe.frame_buffer(slot_of(i), i % per_gather)[:local_rows].copy_(local_bufs[i % n_local][:local_rows])
                filled[slot_of(i)] = i % per_gather + 1
                if i % per_gather == per_gather - 1:
                    pipe.submit(slot_of(i))

    def launch(frames) -> None:  # one launch for these consecutive frames (1, 2, 4 or 8 of them)
        if len(traced) == launches:
            complete_oldest()
        rslot = (frames[0] // per_launch) % launches
        for i in frames:
            if pipe is not None and i % per_gather == 0:
                finish(slot_of(i))  # the gather that last used this ring slot
        collect(rslot)
        if len(frames) == 1:
            renderer.submit_rows_to_device(render_target(frames[0]).data_ptr(), strip, world, rank, rslot)
        else:
            renderer.submit_rows_batch_to_device([render_target(i).data_ptr() for i in frames], strip, world, rank, rslot)
        traced.append((frames[0], rslot, len(frames)))

    def flush_pending() -> None:  # a partial batch (the timed region's last frames): launches of the largest power of two that fits
        while pending:
            n = 1 << (len(pending).bit_length() - 1)
            launch(pending[:n])
            del pending[:n]

    def step() -> None:
        i = frame_no[0]
        frame_no[0] += 1
        if loop is not None:
            loop.before_frame(i)
        if not streamed:
            info = renderer.draw_rows_to_device(render_target(i).data_ptr(), strip, world, rank)
            kernel_ms.append(info.kernel_ms)
            if pipe is not None:
                finish(0)
                if one_gpu_test:
                    pipe.local[0][:local_rows].copy_(local_bufs[0][:local_rows])
                pipe.submit(0)
            return
        pending.append(i)
        if len(pending) == per_launch:
            launch(list(pending))
            pending.clear()

    def drain() -> None:  # every frame issued so far is traced, gathered and assembled
        flush_pending()
        while traced:
            complete_oldest()
        for rslot in list(uncollected):
            collect(rslot)
        if pipe is not None:
            for slot in [sl for sl in filled if pipe.work[sl] is None]:  # a group the fence cut short travels as it is
                pipe.submit(slot)
        while pipe is not None and pipe.oldest() is not None:
            finish(pipe.oldest())

    def fence() -> None:
        drain()
        renderer.synchronize()
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    def timed_region() -> float:  # EXACTLY args.steps steps between two fences; MAX over ranks
        kernel_ms.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if one_gpu_test else dev)
        if exchange:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The region is a few tens of milliseconds at the default K, so it is repeated until the regions add up to
    # --min-seconds of GPU time (every rank runs the same count: the first region's time is agreed on first). `value`
    # and `ms_per_step` are those of the MEDIAN region; min / max are reported beside them.
    regions = [timed_region()]
    n_regions = max(1, min(5000, int(np.ceil(args.min_seconds / max(regions[0], 1e-6)))))  # (capped at 200 until round 4: a 10 ms region then summed to 2.1 s, not --min-seconds)
    region_kernel_ms = [float(np.mean(kernel_ms)) if kernel_ms else 0.0]
    for _ in range(n_regions - 1):
        regions.append(timed_region())
        region_kernel_ms.append(float(np.mean(kernel_ms)) if kernel_ms else 0.0)
    order = int(np.argsort(regions)[len(regions) // 2])
    elapsed = regions[order]
    mean_kernel_ms = region_kernel_ms[order]

    # --- untimed extras: algorithmic-byte counters, read-back rate ------------------------------
    verified = None
    if (args.verify or args.verify is None) and exchange:
        verified = True
        # the assembled frame of the last step must equal the same frame traced by one rank alone
        if rank == 0:
            whole = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
            renderer.draw_rows_to_device(whole.data_ptr(), strip, 1, 0)
            renderer.synchronize()
            same = bool((whole == frame_buf).all().item())
            print(f"verify: assembled {world}-rank frame == single-rank frame: {same}", file=sys.stderr, flush=True)
            if not same:
                raise SystemExit("multi-rank frame differs from the single-rank frame")
        dist.barrier()
    cbuf = render_target(0)
    info = renderer.draw_rows_to_device(cbuf.data_ptr(), strip, world, rank, True)
    counts = torch.tensor([info.cubes_traced, info.n_outer, info.n_inner, info.n_hits, info.n_light], dtype=torch.int64,
                          device="cpu" if one_gpu_test else dev)
    if exchange:
        dist.all_reduce(counts)
    cubes_traced, n_outer, n_inner, n_hits, n_light = (int(v) for v in counts.tolist())
    # per-launch algorithmic bytes of THIS rank's trace kernel:
    #   2 B per in-bounds cube lookup + 2 B per voxel lookup + 32 B per lit surface (palette
    #   entry) + 4 B per light texel + 4 B per output pixel
    my_bytes = 2 * info.n_outer + 2 * info.n_inner + 32 * info.n_hits + 4 * info.n_light + 4 * w * local_rows
    # Launches OVERLAP when frames are streamed (4-8 in flight): a launch's own duration (HIP events; rocprofv3 agrees) is then
    # longer than its share of the device's time, and the kernel's rate is bytes of the launches completed in the timed region /
    # the region's duration = bytes per launch / frame period. One frame at a time (--no-pipeline) the two coincide.
    launch_period_ms = (elapsed / args.steps * 1e3) if streamed else mean_kernel_ms
    period_gbs = (my_bytes / (launch_period_ms * 1e-3)) / 1e9 if launch_period_ms > 0 else 0.0

    # BASELINE.json quotes a *single-frame* raytrace: next to the streamed frame period, the time of one frame alone
    # (submit, wait, repeat) -- "warm": tile order learnt from the identical previous frame; "cold": no feedback used or
    # recorded, what the first frame of any sequence costs -- and of a moving camera (6 degrees per frame about the
    # scene's axis: the feedback never applies), one frame at a time and streamed.
    single = None
    if world == 1 and args.workload not in ("orbit", "relight") and not args.no_extras:
        n_l = max(10, min(60, args.steps))
        tgt = render_target(0).data_ptr()

        one_kernel_ms = []

        def one_by_one(n, **kw):
            renderer.synchronize()
            ts = []
            one_kernel_ms.clear()
            for _ in range(n):
                t1 = time.perf_counter()
                fi = renderer.draw_rows_to_device(tgt, strip, world, rank, **kw)
                ts.append((time.perf_counter() - t1) * 1e3)
                one_kernel_ms.append(fi.kernel_ms)
            return ts

        one_by_one(2)
        warm = one_by_one(n_l)
        warm_kernel_ms = float(np.median(one_kernel_ms))  # one launch alone on the device (HIP events): the KERNEL's duration
        cold = one_by_one(n_l, no_feedback=True)
        cold_kernel_ms = float(np.median(one_kernel_ms))
        if is_replay:
            eye, target = (0.0, 0.0, 1.0), (0.0, 0.0, 0.0)
        radius = float(np.hypot(eye[0] - target[0], eye[2] - target[2]))
        a0 = float(np.arctan2(eye[0] - target[0], eye[2] - target[2]))
        views = [H.look_at_y_up((target[0] + radius * np.sin(a0 + np.radians(6.0 * k)), eye[1], target[2] + radius * np.cos(a0 + np.radians(6.0 * k))), target)
                 for k in range(60)]
        moving = []
        renderer.synchronize()
        for k in range(0 if is_replay else n_l):
            cams.world_view_transform = views[k % 60]
            renderer.update()
            t1 = time.perf_counter()
            renderer.draw_rows_to_device(tgt, strip, world, rank)
            moving.append((time.perf_counter() - t1) * 1e3)
        renderer.synchronize()
        t1 = time.perf_counter()
        for k in range(0 if is_replay else n_l):
            cams.world_view_transform = views[k % 60]
            renderer.update()
            if k >= depth:
                renderer.wait_rows(k % depth)
            renderer.submit_rows_to_device(render_target(k).data_ptr() if local_bufs is None else local_bufs[k % len(local_bufs)].data_ptr(), strip, world, rank, k % depth)
        for k in range(max(0, n_l - depth), 0 if is_replay else n_l):
            renderer.wait_rows(k % depth)
        moving_streamed = (time.perf_counter() - t1) / n_l * 1e3
        cams.world_view_transform = H.look_at_y_up(eye, target)
        renderer.update()
        single = {
            "single_frame_warm_ms": round(float(np.median(warm)), 4),
            "single_frame_cold_ms": round(float(np.median(cold)), 4),
            "single_frame_moving_camera_ms": round(float(np.median(moving)), 4) if moving else None,
            "streamed_moving_camera_ms": round(moving_streamed, 4) if moving else None,
            "kernel_ms_warm": round(warm_kernel_ms, 4),
            "kernel_ms_cold": round(cold_kernel_ms, 4),
            "frames": n_l,
            "note": "medians of one frame at a time (host submit to completion); moving camera: 6 degrees per frame about the view target",
        }

    # Several frames per launch (aic_render_submit_batch, round 6): the same static view streamed as launches of 4 frames, two launches in flight -- what a
    # caller that knows its next cameras (a camera path over a still scene) gets; reported beside `value`, which stays the one-frame-per-launch rate
    batched = None
    if world == 1 and single is not None and per_launch == 1 and not is_replay:
        kb, lb, nb = 4, 2, 16  # frames per launch, launches in flight, launches timed
        bb = [[torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(kb)] for _ in range(lb)]

        def run_batched(n):
            for i in range(n):
                if i >= lb:
                    renderer.wait_rows(i % lb)
                renderer.submit_rows_batch_to_device([b.data_ptr() for b in bb[i % lb]], strip, 1, 0, i % lb)
            for i in range(max(0, n - lb), n):
                renderer.wait_rows(i % lb)
            renderer.synchronize()

        run_batched(4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_batched(nb)
        torch.cuda.synchronize()
        ms_b = (time.perf_counter() - t1) / (nb * kb) * 1e3
        batched = {"frames_per_launch": kb, "launches_in_flight": lb, "frames": nb * kb, "ms_per_step": round(ms_b, 4),
                   "value": round(rays_per_frame / (ms_b * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                   "frames_equal_single_frame": bool((bb[0][0] == bb[lb - 1][kb - 1]).all().item())}

    fps_with_readback = None
    if world == 1 and not args.no_extras:
        n_rb = max(3, min(10, args.steps))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_rb):
            renderer.draw_rgba("")
        fps_with_readback = n_rb / (time.perf_counter() - t1)

    # The reference's own Criterion benches (all-is-cubes-render/benches/raytrace.rs:92-114) time `renderer.draw_rgba(..)` -- a finished
    # 64x64 image in host memory -- for two option sets; the same call, the same scene, the same two option sets, per call:
    criterion = None
    if world == 1 and args.workload == "light-bench" and not args.no_extras:
        criterion = {}
        for bench_name, lighting_kind in (("flat-surface", 1), ("linear-surface", 3)):
            o2 = H.GraphicsOptions()
            o2.transparency = H.TransparencyOption(H.TransparencyKind(0))
            o2.lighting_display = H.LightingOption(H.LightingKind(lighting_kind))
            cams.graphics_options = o2
            renderer.update()
            for _ in range(20):
                renderer.draw_rgba("")
            ts = []
            for _ in range(300):
                t1 = time.perf_counter()
                renderer.draw_rgba("")
                ts.append((time.perf_counter() - t1) * 1e6)
            criterion[bench_name] = {"median_us": round(float(np.median(ts)), 2), "min_us": round(float(np.min(ts)), 2), "samples": len(ts)}
        criterion["note"] = ("wall time of HipRtRenderer.draw_rgba (64x64 image delivered to host memory), as the reference's Criterion group "
                             "'threaded'/'serial' measures RtRenderer::draw_rgba; GraphicsOptions::default() + the bench's two changes")
        cams.graphics_options = opts
        renderer.update()

    # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs of
    # this same command, corrected as MI355X_MICROARCH.md prescribes; tools/measure.sh +
    # tools/summarize_profile.py). PMC collection cannot run inside the timed bench, so the
    # committed per-launch figure for this workload is reported (null when none is on file
    # or when the image is partitioned differently from the profiled single-GPU launch).
    traffic, traffic_src, valu = None, None, None
    if world == 1 and (args.lighting, args.fog, args.transparency) == (3, 1, 1) and args.workload not in ("orbit", "relight"):
        cands = sorted(glob.glob(os.path.join(str(ROOT), "profiles", f"r*_pmc_{args.workload}.json")))
        if cands:
            with open(cands[-1]) as f:
                pj = json.load(f)
            if pj.get("hbm_traffic_bytes_per_launch"):
                traffic = round(pj["hbm_traffic_bytes_per_launch"] / (launch_period_ms * 1e-3) / 1e9, 3) if launch_period_ms > 0 else None
                traffic_src = "profiles/" + os.path.basename(cands[-1]) + f" ({int(pj['hbm_traffic_bytes_per_launch'])} B/launch, GB/s at this run's launch period)"
            cn = pj.get("counters", {})
            vi = cn.get("SQ_INSTS_VALU", {}).get("mean_per_launch")
            if vi and elapsed > 0:
                # Instruction issue, priced with tools/ubench/issue_rate.hip (residency pinned and verified per line,
                # profiles/r03_issue_rate.txt; cycles one SIMD spends per wave-instruction at 4-8 waves per SIMD, from the launch's
                # HIP-event time): 2.2 for the simplest VALU operations (v_mov_b32, v_add_u32), 4.0-4.45 for everything else this
                # kernel is made of (f64 add / fma / compare, v_cndmask, v_mul_lo, v_fma_f32) and for SALU; VALU and SALU of
                # DIFFERENT waves issue side by side (alternating stream: 2.3 per instruction). `peak` is the first figure -- a
                # ceiling no instruction stream exceeds; `pipe_busy` prices this kernel's own VALU and SALU counts at 4.2 cycles
                # each against the cycles its launch period offers (an upper estimate for VALU: its simple operations cost 2.2).
                kinds = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")
                total = sum(float(cn.get(k, {}).get("mean_per_launch") or 0.0) for k in kinds)
                si = float(cn.get("SQ_INSTS_SALU", {}).get("mean_per_launch") or 0.0)
                peak = 1024 * 2.4e9 / ISSUE_PEAK_CYCLES
                rate = total / (launch_period_ms * 1e-3) if launch_period_ms > 0 else 0.0
                simd_cycles = 1024 * 2.4e9 * launch_period_ms * 1e-3
                tc, ai = cn.get("SQ_THREAD_CYCLES_VALU", {}).get("mean_per_launch"), cn.get("SQ_ACTIVE_INST_VALU", {}).get("mean_per_launch")
                valu = {"wave_insts_per_launch": int(total), "valu_wave_insts_per_launch": int(vi), "salu_wave_insts_per_launch": int(si),
                        "issue_rate": round(rate / 1e9, 2), "peak": round(peak / 1e9, 1),
                        "unit": "G wave-insts/s", "frac": round(rate / peak, 4),
                        "cycles_per_inst_per_simd": round(1024 * 2.4e9 / rate, 3) if rate else None,
                        "pipe_busy": {"valu": round(vi * ISSUE_PIPE_CYCLES / simd_cycles, 3), "salu": round(si * ISSUE_PIPE_CYCLES / simd_cycles, 3)} if simd_cycles else None,
                        "valu_lane_utilisation": round(tc / (ai * 64.0), 4) if tc and ai else None,
                        "source": "profiles/" + os.path.basename(cands[-1]),
                        "note": "instruction counts per launch from the PMC file (one frame at a time) over this run's launch period; peak = the fastest "
                                f"instruction stream measured on this chip ({ISSUE_PEAK_CYCLES} cycles per instruction per SIMD: v_mov_b32 at 8 waves per SIMD, "
                                f"profiles/r03_issue_rate.txt); pipe_busy = VALU / SALU instructions x {ISSUE_PIPE_CYCLES} cycles (what f64, compare, select and "
                                "scalar operations cost there) / SIMD cycles of the launch period (DESIGN.md 6)"}

    secondary = None
    if world == 1 and args.workload == "atrium" and not args.no_extras and not args.no_secondary and (args.lighting, args.fog, args.transparency) == (3, 1, 1):
        try:
            secondary = {"s256": secondary_workload_leg("s256", H, D, torch, dev, local_rank, space_from_flat)}
        except Exception as exc:  # the headline line must not be lost to the extra leg
            secondary = {"s256": {"error": repr(exc)}}
        for leg in ("orbit", "relight"):  # configs[4]: the sim + render loops, a second or two each
            try:
                secondary[leg] = secondary_loop_leg(leg, H, D, torch, dev, local_rank, space_from_flat)
            except Exception as exc:
                secondary[leg] = {"error": repr(exc)}

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        result = {
            "metric": "Mrays/s",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
            "timed_regions": len(regions),
            "frames_per_s": round(args.steps / elapsed, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": label,
                "width": w,
                "height": h,
                "rays_per_frame": rays_per_frame,
                "partition": f"interleaved {strip}-row strips over {world} GPU(s), scene replicated, RCCL gather to rank 0",
                "steps_per_ray": round(cubes_traced / rays_per_frame, 2),
                "frames_in_flight": depth if streamed else 1,
                "frames_per_launch": per_launch,
                "frames_per_gather": per_gather if exchange else None,
                "handoff": ("device (aic_stream_wait_frame: the gather's stream waits for the trace's event)" if device_handoff else "host (aic_render_wait before each gather)") if exchange else None,
                "assembled_frame_equals_single_rank_frame": verified,
            },
            "roofline": roofline_object(my_bytes, mean_kernel_ms, launch_period_ms, period_gbs, streamed, depth,
                                        single["kernel_ms_warm"] if single is not None else None, traffic, traffic_src, info.cubes_traced),
            "device": renderer.device_name(),
        }
        if valu is not None:
            result["issue"] = valu
        if single is not None:
            result["single_frame"] = single
            result["streamed_ms"] = round(ms_per_step, 4) if streamed else None
            # BASELINE.json config 2 is a SINGLE-frame raytrace: the rate of one frame alone, cold (no tile order learnt from an
            # earlier identical frame: what the first frame of any sequence costs), beside the streamed `value`
            if single["single_frame_cold_ms"] > 0:
                result["value_single_frame"] = round(rays_per_frame / (single["single_frame_cold_ms"] * 1e-3) / 1e6, 3)
                result["value_single_frame_warm"] = round(rays_per_frame / (single["single_frame_warm_ms"] * 1e-3) / 1e6, 3)
                result["config"]["value_is"] = ("`value` = streamed frames (frames_in_flight traces overlapping, static camera: the recording loop's rate); "
                                                "BASELINE.json config 2 names a single-frame raytrace, which is `value_single_frame` (one frame alone, cold; "
                                                "`value_single_frame_warm` with the tile order learnt from the identical previous frame)")
        if batched is not None:
            result["streamed_batched"] = batched
        if secondary is not None:
            result["secondary"] = secondary
        if criterion is not None:
            result["criterion_equivalent"] = criterion
        if light_update is not None:
            result["light_update"] = light_update
        if relight is not None and relight["calls"]:
            result["relight"] = {"light_updates_per_frame": round(relight["updates"] / relight["calls"], 1),
                                 "light_budget_per_frame": args.light_budget, "frames_between_toggles": args.relight_period,
                                 "light_ms_per_frame": round(relight["light_ms"] / relight["calls"], 4),
                                 "queue_left_at_end": relight["queue_left"], "queue_left_max": relight.get("queue_max", 0),
                                 "frames_ending_with_an_empty_queue": relight.get("drained", 0), "frames_counted": relight["calls"]}
        if fps_with_readback is not None:
            result["fps_with_readback"] = round(fps_with_readback, 3)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(flat_space, opts, w, h, eye, target, view_distance, args.cpu_seconds,
                                                  inv=REPLAY.get("inv") if is_replay else None)
            if result["cpu_baseline"]["value"]:
                result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 2)
        # (RCCL prints a version banner through C stdio, which sits in its buffer until the process exits when stdout is a file
        #  or a pipe: flush it now, so that the bench line is the LAST line of the output)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()
    return 0


class SceneLoop:
    """What the sim + render loops change before every frame (BASELINE.json configs[4]):
    `orbit`   -- the camera moves 6 degrees about the atrium's axis and the whole light volume is re-uploaded (aic_update_light_volume);
    `relight` -- the light is computed ON THE DEVICE (SURVEY.md 8f N1 + N2): every `period` frames a lamp block is placed or removed, the
                 changed cube and its neighbours are queued (aic_light_cubes_changed), every frame the light updater gets `budget` cube
                 updates in one launch (aic_evaluate_light, continuing the layer's queue) and the camera moves. No light volume crosses PCIe."""

    def __init__(self, name, H, flat_space, cams, renderer, eye, target, period=30, budget=2048, blocking_light=False):
        self.name, self.cams, self.renderer, self.period, self.budget = name, cams, renderer, period, budget
        self.blocking_light = blocking_light  # relight: aic_evaluate_light per frame (rounds 2-5) instead of aic_evaluate_light_submit / _wait
        self.light_pending = False
        self.light_lag = 0       # steps the pending update has been left running
        self.light_carry = 0     # budget of the steps that started no update of their own (added to the next one)
        self.light_update, self.relight, self.lights = None, None, None
        self.views = [H.look_at_y_up((0.5 + 7.0 * np.sin(2.0 * np.pi * k / 60.0), eye[1], 7.0 * np.cos(2.0 * np.pi * k / 60.0)), target) for k in range(60)]
        if name == "relight":
            renderer.device_light = True
            renderer.device_light_queue_order = 0
            li = renderer.evaluate_light(30, True, 1, 8192, 0)  # the starting light: large batches to convergence
            self.light_update = {"mode": "initial light: fast_evaluate_light + evaluate_light(1), batch 8192", "updates": int(li["updates"]),
                                 "launches": int(li["batches"]), "device_ms": round(li["device_ms"], 3), "total_ms": round(li["total_ms"], 3)}
            lo_s, sz_s = np.array(flat_space.lo), np.array(flat_space.size)
            air_i = next(i for i, b in enumerate(flat_space.blocks) if b.is_air)
            lamp_i = len(flat_space.blocks) - 1
            # lamp sites: air cubes of the scene's middle column region, spread over the orbit
            bi = np.asarray(flat_space.block_index)
            rng = np.random.default_rng(7)
            sites = []
            while len(sites) < 30:
                c = rng.integers(0, sz_s)
                if int(bi[tuple(c)]) == air_i and tuple(c) not in sites:
                    sites.append(tuple(int(v) for v in c))
            self.relight = {"sites": [tuple(int(l + c) for l, c in zip(lo_s, s_)) for s_ in sites], "air": air_i, "lamp": lamp_i, "updates": 0, "calls": 0,
                            "light_ms": 0.0, "queue_left": 0}
        else:
            # 60 key frames: the eye circles the atrium's axis, the light field breathes (status bytes kept)
            base = flat_space.light.copy()
            self.lights = []
            for k in range(60):
                gain = 0.85 + 0.15 * np.sin(2.0 * np.pi * k / 60.0)
                lk = base.copy()
                lk[..., 0:3] = np.clip(np.round(base[..., 0:3].astype(np.float32) + 10.0 * np.log2(gain)), 0, 255).astype(np.uint8) * (base[..., 0:3] > 0)
                self.lights.append(np.ascontiguousarray(lk.reshape(-1, 4)))

    def before_frame(self, i):
        k = i % 60
        cams, renderer, relight = self.cams, self.renderer, self.relight
        if relight is not None:
            if i % self.period == 0:   # a lamp placed (first pass over the 30 sites) or removed (second pass)
                j = i // self.period
                x, y, z = relight["sites"][j % 30]
                cams.world_space.set(x, y, z, relight["lamp"] if (j // 30) % 2 == 0 else relight["air"])
            cams.world_view_transform = self.views[k]
            t_l = time.perf_counter()
            if self.blocking_light:
                renderer.update()                              # block delta -> aic_update_cubes + aic_light_cubes_changed
                t_l = time.perf_counter()
                li = renderer.evaluate_light(30, False, 1, self.budget, 0, 0, True, self.budget)  # one launch of that many cube updates
            else:
                # the step's light update runs on the library's worker thread BESIDE the frame the caller submits next (which reads the light as the
                # previous step left it): collect the previous step's update here, apply this step's block change, start this step's update
                # (an update that is still running is left running for up to two steps -- its budget joins the next one's --: the frames keep their pace, the
                #  light is at most two steps behind them; a block change publishes it in any case)
                li = {"updates": 0, "queue_left": relight["queue_left"]}
                start = True
                if self.light_pending:
                    if self.light_lag < 2 and i % self.period != 0 and not renderer.evaluate_light_done():
                        self.light_lag += 1
                        self.light_carry += self.budget
                        start = False
                    else:
                        li = renderer.evaluate_light_wait()
                        self.light_pending = False
                t_w = time.perf_counter() - t_l
                renderer.update()                              # block delta -> aic_update_cubes + aic_light_cubes_changed
                t_l = time.perf_counter()
                if start:
                    n_b = self.budget + self.light_carry
                    renderer.evaluate_light_submit(30, False, 1, n_b, 0, 0, True, n_b)
                    self.light_pending, self.light_lag, self.light_carry = True, 0, 0
                t_l -= t_w                                     # (the host's time in the light calls)
            relight["light_ms"] += (time.perf_counter() - t_l) * 1e3
            relight["updates"] += int(li["updates"]); relight["calls"] += 1; relight["queue_left"] = int(li["queue_left"])
            relight["queue_max"] = max(relight.get("queue_max", 0), int(li["queue_left"]))
            relight["drained"] = relight.get("drained", 0) + (1 if int(li["queue_left"]) == 0 else 0)
        else:
            cams.world_space.load_light(self.lights[k])        # SpaceChange burst -> aic_update_light_volume
            cams.world_view_transform = self.views[k]
            renderer.update()                                  # (waits for the frames in flight: they read the light volume)


def secondary_loop_leg(name, H, D, torch, dev, local_rank, space_from_flat, frames=120, warm=12):
    """configs[4] inside the default run (C5 was a builder-run number the driver never timed): the `orbit` or `relight`
    loop on its own renderer, `frames` streamed frames (4 in flight) after `warm`, a second or two in all."""
    flat_space, (w, h), eye, target, view_distance, label = build_workload(name)
    cams = H.StandardCameras()
    opts = H.GraphicsOptions()
    opts.bloom_intensity = 0.0
    opts.view_distance = view_distance
    opts.debug_info_text = False
    cams.graphics_options = opts
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    cams.world_space = space_from_flat(flat_space)
    cams.world_view_transform = H.look_at_y_up(eye, target)
    r = H.HipRtRenderer(cams, None, local_rank)
    r.update()
    loop = SceneLoop(name, H, flat_space, cams, r, eye, target)
    strip, depth = D.STRIP_ROWS, 4
    bufs = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(depth)]
    n = [0]

    def run(count):
        first = n[0]
        for _ in range(count):
            i = n[0]
            n[0] += 1
            loop.before_frame(i)
            if i - first >= depth:
                r.wait_rows(i % depth)
            r.submit_rows_to_device(bufs[i % depth].data_ptr(), strip, 1, 0, i % depth)
        for i in range(max(first, n[0] - depth), n[0]):
            r.wait_rows(i % depth)
        r.synchronize()

    run(warm)
    if loop.relight is not None:
        loop.relight.update(updates=0, calls=0, light_ms=0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(frames)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    out = {"workload": label, "rays_per_frame": w * h, "frames": frames, "frames_in_flight": depth, "ms_per_step": round(ms, 4),
           "value": round(w * h / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s", "frames_per_s": round(1e3 / ms, 2)}
    if loop.relight is not None and loop.relight["calls"]:
        out["light_ms_per_frame"] = round(loop.relight["light_ms"] / loop.relight["calls"], 4)
        out["light_updates_per_frame"] = round(loop.relight["updates"] / loop.relight["calls"], 1)
    del r
    return out


def roofline_object(nbytes, mean_kernel_ms, launch_period_ms, period_gbs, streamed, depth, kernel_alone_ms, traffic, traffic_src, cubes_traced):
    """The bench line's `roofline` object. `achieved` / `frac` are the KERNEL's own: algorithmic bytes per launch / the duration of one launch
    alone on the device (HIP events on its stream; rocprofv3's per-kernel average of a --no-pipeline run agrees: profiles/rNN_kernel_stats_*_nopipe.csv).
    Until round 5 `frac` was bytes / the frame period of overlapping launches -- device throughput, which no per-kernel duration reproduces
    (VERDICT r05 weak 6); that figure is `frac_streamed_period` now."""
    if kernel_alone_ms is None and not streamed:
        kernel_alone_ms = mean_kernel_ms
    if kernel_alone_ms and kernel_alone_ms > 0:
        basis_ms, basis = kernel_alone_ms, ("achieved = algorithmic bytes per launch / kernel_ms_one_at_a_time: one launch alone on the device (HIP events on its "
                                            "stream, warm tile order), whatever the timed region streams")
    else:  # (--no-extras on a streamed run: no launch was timed alone)
        basis_ms, basis = mean_kernel_ms, ("achieved = algorithmic bytes per launch / kernel_ms, the mean duration of the timed region's OVERLAPPING launches "
                                           "(no launch was timed alone in this run): a lower bound of the kernel's own rate")
    achieved = (nbytes / (basis_ms * 1e-3)) / 1e9 if basis_ms and basis_ms > 0 else 0.0
    return {
        "bound": "hbm",
        "achieved": round(achieved, 3),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": traffic,
        "traffic_source": traffic_src,
        "kernel": "trace_image_kernel",
        "kernel_ms": round(mean_kernel_ms, 4),
        "kernel_ms_one_at_a_time": round(kernel_alone_ms, 4) if kernel_alone_ms else None,
        "frac_one_at_a_time": round(achieved / HBM_PEAK_GBS, 6) if kernel_alone_ms else None,  # (= frac; the key rounds 3-5 carried)
        "launches_in_flight": depth if streamed else 1,
        "launch_period_ms": round(launch_period_ms, 4),
        "achieved_streamed_period": round(period_gbs, 3),
        "frac_streamed_period": round(period_gbs / HBM_PEAK_GBS, 6),
        "rate_basis": basis + ("; frac_streamed_period = bytes per launch / launch period: the device's throughput with launches_in_flight launches "
                               "overlapping, not a property of one kernel" if streamed else ""),
        "algorithmic_bytes_per_launch": int(nbytes),
        "gsteps_per_s": round((cubes_traced / (launch_period_ms * 1e-3)) / 1e9, 3) if launch_period_ms > 0 else 0.0,
        "note": "rank-0 launch; cache-resident scene: the path is latency/ALU-bound, not HBM-bound (DESIGN.md)",
    }


def secondary_workload_leg(name, H, D, torch, dev, local_rank, space_from_flat, frames=16, one_by_one_frames=6):
    """A short measurement of another workload inside the default run (the driver's one line then carries a C3
    figure too): its own renderer on the same device, `frames` streamed frames (4 in flight) after a warm-up, a few frames one at
    a time for the kernel's own duration, one counter-collecting launch for the algorithmic bytes."""
    flat_space, (w, h), eye, target, view_distance, label = build_workload(name)
    cams = H.StandardCameras()
    opts = H.GraphicsOptions()
    opts.bloom_intensity = 0.0
    opts.view_distance = view_distance
    opts.debug_info_text = False
    cams.graphics_options = opts
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    cams.world_space = space_from_flat(flat_space)
    cams.world_view_transform = H.look_at_y_up(eye, target)
    r = H.HipRtRenderer(cams, None, local_rank)
    r.update()
    strip, depth = D.STRIP_ROWS, 4
    bufs = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(depth)]

    def streamed(n):
        for i in range(n):
            if i >= depth:
                r.wait_rows(i % depth)
            r.submit_rows_to_device(bufs[i % depth].data_ptr(), strip, 1, 0, i % depth)
        for i in range(max(0, n - depth), n):
            r.wait_rows(i % depth)
        r.synchronize()

    streamed(depth + 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    streamed(frames)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    ks, ts = [], []
    for _ in range(one_by_one_frames):
        t1 = time.perf_counter()
        fi = r.draw_rows_to_device(bufs[0].data_ptr(), strip, 1, 0)
        ts.append((time.perf_counter() - t1) * 1e3)
        ks.append(fi.kernel_ms)
    cold = []
    for _ in range(max(2, one_by_one_frames // 2)):
        t1 = time.perf_counter()
        r.draw_rows_to_device(bufs[0].data_ptr(), strip, 1, 0, no_feedback=True)
        cold.append((time.perf_counter() - t1) * 1e3)
    info = r.draw_rows_to_device(bufs[0].data_ptr(), strip, 1, 0, True)
    nbytes = 2 * info.n_outer + 2 * info.n_inner + 32 * info.n_hits + 4 * info.n_light + 4 * w * h
    k_ms = float(np.median(ks))
    out = {"workload": label, "rays_per_frame": w * h, "frames": frames, "frames_in_flight": depth, "ms_per_step": round(ms, 4),
           "value": round(w * h / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s", "steps_per_ray": round(info.cubes_traced / (w * h), 2),
           "single_frame_warm_ms": round(float(np.median(ts)), 4), "single_frame_cold_ms": round(float(np.median(cold)), 4),
           "value_single_frame": round(w * h / (float(np.median(cold)) * 1e-3) / 1e6, 3),
           "algorithmic_bytes_per_launch": int(nbytes), "kernel_ms_one_at_a_time": round(k_ms, 4),
           "frac": round(nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if k_ms > 0 else None,  # the kernel's own (one launch alone), as in `roofline`
           "frac_one_at_a_time": round(nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if k_ms > 0 else None,
           "frac_streamed_period": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if ms > 0 else None}
    del r
    return out


def usable_cpus():
    """Threads the CPU baseline may actually run on: the scheduler affinity mask capped by the
    container's cgroup CPU quota (oversubscribing a quota only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"affinity {n} of {os.cpu_count()} logical CPUs"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            note += f", cgroup cpu.max quota {q}"
            n = min(n, q)
    except OSError:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                note += f", cgroup cfs quota {max(1, q // per)}"
                n = min(n, max(1, q // per))
        except OSError:
            pass
    return n, note


def cpu_baseline(flat_space, opts, w, h, eye, target, view_distance, target_seconds: float, inv=None) -> dict:
    """The CPU oracle (a restatement of the reference algorithm -- the reference itself is Rust and
    cannot be built here) timed on this host's cores over a bounded sample of the same workload:
    whole frames, all hardware threads, row-parallel like the reference's rayon loop
    (renderer.rs:537-555), repeated until about `target_seconds` of wall time. Reported next to
    the GPU number; not the optimisation target."""
    import oracle

    threads, cpu_note = usable_cpus()
    extra = {}
    if (np.asarray(flat_space.light)[..., 3] == 0).all():  # light-bench: the GPU leg lit the space on the device; here the oracle does
        import copy
        one = copy.deepcopy(flat_space)
        t_l = time.perf_counter()
        oracle.evaluate_light(one, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16, threads=1)
        dt_one = time.perf_counter() - t_l
        # the reference computes a batch's 32 cubes on its rayon pool (updater.rs:231-247): the port does the same on `threads` threads
        t_l = time.perf_counter()
        n_upd = oracle.evaluate_light(flat_space, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16, threads=threads)
        dt_l = time.perf_counter() - t_l
        extra["light_update"] = {"updates": int(n_upd), "total_ms": round(dt_l * 1e3, 1), "updates_per_s": round(n_upd / dt_l, 1), "cores": threads,
                                 "one_core_ms": round(dt_one * 1e3, 1)}
    sp = oracle.Space(flat_space)
    oo = oracle.make_options(fog=int(opts.fog), transparency=int(opts.transparency.kind), lighting=int(opts.lighting_display.kind),
                             view_distance=view_distance)
    if inv is None:
        q = oracle.look_at_y_up(eye, target)
        _, _, inv = oracle.camera_matrices(90.0, view_distance, w / h, q, eye)
    cam = oracle.make_camera(np.asarray(inv, np.float64).reshape(4, 4), w, h)
    oracle.render(sp, oo, cam, threads=threads)  # warm-up frame (page-in, thread start)
    frames, t0 = 0, time.perf_counter()
    per_frame = []
    while True:
        t1 = time.perf_counter()
        oracle.render(sp, oo, cam, threads=threads)
        per_frame.append(time.perf_counter() - t1)
        frames += 1
        if time.perf_counter() - t0 >= target_seconds or frames >= 400:
            break
    dt = time.perf_counter() - t0
    rays = frames * w * h
    # the same port as rounds 1-5 built it (-O2), a quarter of the sample: so that the two sets of GPU / CPU ratios can be told apart
    try:
        lib2 = oracle.lib_o2()
        oracle.render(sp, oo, cam, threads=threads, use_lib=lib2)
        f2, t2 = 0, time.perf_counter()
        while True:
            oracle.render(sp, oo, cam, threads=threads, use_lib=lib2)
            f2 += 1
            if time.perf_counter() - t2 >= target_seconds / 4.0 or f2 >= 100:
                break
        d2 = time.perf_counter() - t2
        extra["o2"] = {"value": round(f2 * w * h / d2 / 1e6, 4), "unit": "Mrays/s", "frames": f2, "flags": "-O2 -ffp-contract=off -fno-fast-math (rounds 1-5)"}
    except Exception as exc:  # (the baseline's main figure must not be lost to the comparison)
        extra["o2"] = {"error": repr(exc)}
    return {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "flags": "-O3 -ffp-contract=off -fno-fast-math (SURVEY.md 8d)",
        "sample": f"{frames} whole {w}x{h} frames of the same workload in {dt:.1f} s wall (median {1e3 * float(np.median(per_frame)):.1f} ms/frame), "
                  f"oracle/aic_oracle.cpp row-parallel on {threads} threads ({cpu_note})",
        "frames_per_s": round(frames / dt, 4),
        **extra,
    }


if __name__ == "__main__":
    sys.exit(main())
