"""Randomised test of the boundary's STATE MACHINE (not of the arithmetic, which tests/test_gpu_fuzz.py holds to the oracle): a
random sequence of the calls a host makes -- frames submitted on random slots (alone or 2 / 4 per launch, whole or as one part of a
row partition), waits in random order, cube updates, block replacements, light-volume re-uploads, option changes, light updates on
the context's worker thread (aic_evaluate_light_submit: frames submitted before its wait read the light as it stood) -- against a second
context that receives the same scene calls and draws every frame synchronously at the moment it is submitted. A frame collected
later, after any number of scene changes, must be byte for byte the frame the scene stood for when it was submitted (the
reference's `update()` / `draw()` pair: renderer.rs:96-141, 282-308), with its step total; nothing of another slot's frame, cost
record, tile order, counters or light volume may leak.

Seeds: AIC_PROTOCOL_FUZZ_N (default 6), AIC_PROTOCOL_FUZZ_CALLS calls each (default 160)."""
import os

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat
from all_is_cubes_amd import workloads as scenes

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("AIC_PROTOCOL_FUZZ_N", "6"))
N_CALLS = int(os.environ.get("AIC_PROTOCOL_FUZZ_CALLS", "160"))
SIZES = [(96, 64), (130, 70), (64, 33), (200, 120)]
SLOTS = 10  # past the eight made with the context: slots 8 and 9 get their stream on first use


def cameras(rng, w, h, view_distance):
    eye = (8.5 + float(rng.uniform(-5, 5)), 14.5 + float(rng.uniform(-3, 4)), 26.0 + float(rng.uniform(-6, 6)))
    _, _, inv = oracle.camera_matrices(float(rng.choice([60.0, 90.0])), view_distance, w / h, oracle.look_at_y_up(eye, (8.0, 5.0, 8.0)), eye)
    return inv


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_call_sequences_against_a_synchronous_twin(seed):
    import torch

    rng = np.random.default_rng(1000 + seed)
    space = scenes.synthetic_space(n=16, resolution=8, n_blocks=6, seed=20 + seed, light="field")
    n_blocks = len(space.blocks)
    opts = dict(fog=int(rng.integers(0, 4)), transparency=int(rng.integers(0, 3)), lighting=int(rng.integers(0, 5)), view_distance=float(rng.choice([30.0, 200.0])))
    a, b = abi.Context(0), abi.Context(0)  # a: the context under test; b: its synchronous twin
    try:
        for c in (a, b):
            c.upload_space(abi.LAYER_WORLD, space)
            c.set_options(abi.LAYER_WORLD, abi.make_options(**opts))
        in_flight = {}  # slot -> list of (device buffer, expected rows, expected steps)
        checked = 0
        pending_light = []  # the parameters of a light update submitted on `a` and not yet collected: the twin makes the same update, blocking, when it is

        def settle_light():
            """The wait publishes the worker's update on `a`; the twin makes the same update now. Called in front of every scene or light call
            (which would publish it by itself: INTEGRATION.md "Threads") so that both contexts change at the same point of the sequence."""
            if pending_light:
                kw = pending_light.pop()
                ia = a.evaluate_light_wait(abi.LAYER_WORLD)
                ib = b.evaluate_light(abi.LAYER_WORLD, **kw)
                assert (ia.updates, ia.queue_left) == (ib.updates, ib.queue_left), f"seed {seed}: the worker's light update differs from the blocking one"

        def collect(slot):
            nonlocal checked
            items = in_flight.pop(slot)
            if len(items) == 1:
                infos = [a.render_wait(slot)]
            else:
                infos = a.render_wait_batch(slot, len(items))
            for (buf, want, steps), info in zip(items, infos):
                got = buf.cpu().numpy()
                assert (got[: want.shape[0]] == want).all(), f"seed {seed}: slot {slot} differs from the frame at submit time"
                assert int(info.cubes_traced) == steps
                checked += 1

        def make_frames(count):
            w, h = SIZES[int(rng.integers(0, len(SIZES)))]
            part = None
            if rng.random() < 0.35:
                n_parts = int(rng.integers(2, 4))
                part = (8, n_parts, int(rng.integers(0, n_parts)))
            flags = abi.FRAME_NO_FEEDBACK if rng.random() < 0.2 else 0
            tuning = abi.tuning(variant=abi.VARIANT_EXCHANGING) if rng.random() < 0.3 else 0
            return [abi.Context.make_frame(w, h, world_inv=cameras(rng, w, h, opts["view_distance"]), partition=part, flags=flags, tuning=tuning) for _ in range(count)], w, h

        for _ in range(N_CALLS):
            op = rng.random()
            if op < 0.45:  # submit: one frame, or 2 / 4 in one launch
                slot = int(rng.integers(0, SLOTS))
                if slot in in_flight:
                    collect(slot)
                count = int(rng.choice([1, 1, 1, 2, 4]))
                frames, w, h = make_frames(count)
                items = []
                for f in frames:
                    ref = b.render(f)  # the twin draws it now: what the scene stands for at this moment
                    buf = torch.zeros((max(ref["rgba8"].shape[0], 1), w, 4), dtype=torch.uint8, device="cuda")
                    items.append((buf, ref["rgba8"].copy(), int(ref["info"].cubes_traced)))
                torch.cuda.synchronize()  # (torch clears the buffers on ITS stream: the clear must not land on a frame the library's stream has written)
                if count == 1:
                    a.render_submit(frames[0], items[0][0].data_ptr(), slot)
                else:
                    a.render_submit_batch(frames, [it[0].data_ptr() for it in items], slot)
                in_flight[slot] = items
            elif op < 0.60 and in_flight:  # collect a random slot
                collect(int(rng.choice(list(in_flight))))
            elif op < 0.66:  # blocks placed, the light updater told, an update started on the worker thread; frames go on reading the light as it stands
                settle_light()
                n = int(rng.integers(1, 6))
                xyz = np.stack([rng.integers(1, 15, n), rng.integers(1, 15, n), rng.integers(1, 15, n)], 1).astype(np.int32)
                blocks = rng.integers(0, n_blocks, n).astype(np.uint16)
                kw = dict(maximum_distance=int(rng.choice([4, 8])), fast=False, queue=[], max_updates=int(rng.choice([0, 40, 200])), batch=int(rng.choice([32, 256])))
                for c in (a, b):
                    c.update_cubes(abi.LAYER_WORLD, xyz, blocks)
                    c.light_cubes_changed(abi.LAYER_WORLD, xyz)
                a.evaluate_light_submit(abi.LAYER_WORLD, **kw)
                pending_light.append(kw)
            elif op < 0.72:  # cubes change (waits for the frames in flight, which keep what they were given)
                settle_light()
                n = int(rng.integers(1, 12))
                xyz = np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n), rng.integers(0, 16, n)], 1).astype(np.int32)
                blocks = rng.integers(0, n_blocks, n).astype(np.uint16)
                light = rng.integers(0, 256, (n, 4)).astype(np.uint8)
                light[:, 3] = rng.choice([1, 128, 255], n)
                for c in (a, b):
                    c.update_cubes(abi.LAYER_WORLD, xyz, blocks, light)
            elif op < 0.80:  # a block's palette changes: its colours, or it becomes an atom and back
                settle_light()
                idx = int(rng.integers(1, n_blocks))
                if rng.random() < 0.5:
                    blk = flat.atom((float(rng.random()), float(rng.random()), float(rng.random()), float(rng.choice([1.0, 0.5]))))
                else:
                    blk = scenes.synthetic_blocks(8, 1, seed=int(rng.integers(1, 1 << 30)), translucent=bool(rng.random() < 0.5))[0]
                for c in (a, b):
                    c.replace_block(abi.LAYER_WORLD, idx, blk)
            elif op < 0.90:  # the whole light volume again (beside the frames in flight: they keep the volume they were given)
                settle_light()
                light = rng.integers(0, 256, tuple(space.size) + (4,)).astype(np.uint8)
                light[..., 3] = rng.choice([1, 128, 255, 255], tuple(space.size))
                for c in (a, b):
                    c.update_light_volume(abi.LAYER_WORLD, light)
            elif op < 0.95:  # options
                settle_light()
                opts.update(fog=int(rng.integers(0, 4)), lighting=int(rng.integers(0, 5)))
                for c in (a, b):
                    c.set_options(abi.LAYER_WORLD, abi.make_options(**opts))
            else:  # a synchronous frame on slot 0 between the streamed ones
                if 0 in in_flight:
                    collect(0)
                frames, w, h = make_frames(1)
                ra, rb = a.render(frames[0]), b.render(frames[0])
                assert (ra["rgba8"] == rb["rgba8"]).all() and ra["info"].cubes_traced == rb["info"].cubes_traced
                checked += 1
        for slot in list(in_flight):
            collect(slot)
        settle_light()
        a.synchronize()
        assert (a.read_light_volume(abi.LAYER_WORLD, tuple(space.size)) == b.read_light_volume(abi.LAYER_WORLD, tuple(space.size))).all()
        assert checked > 20
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("seed", range(max(2, N_SEEDS // 2)))
def test_random_call_sequences_on_the_multi_device_context(seed):
    """The same for `aic_multi` (csrc/aic_multi.cpp; the form the Rust shim's `with_devices` holds): 2 or 3 device entries -- the one GPU,
    repeated -- with frames streamed through `aic_multi_render_submit / _wait` (host and device targets, up to eight slots), scene calls
    replicated, blocking light updates on the device that runs the updater; the twin is a single-device context drawing synchronously."""
    import torch

    rng = np.random.default_rng(5000 + seed)
    space = scenes.synthetic_space(n=16, resolution=8, n_blocks=6, seed=40 + seed, light="field")
    n_blocks = len(space.blocks)
    opts = dict(fog=int(rng.integers(0, 4)), transparency=int(rng.integers(0, 3)), lighting=int(rng.integers(0, 5)), view_distance=200.0)
    m, b = abi.MultiContext([0] * int(rng.integers(2, 4))), abi.Context(0)
    try:
        for c in (m, b):
            c.upload_space(abi.LAYER_WORLD, space)
            c.set_options(abi.LAYER_WORLD, abi.make_options(**opts))
        in_flight = {}  # slot -> (host array or device tensor, expected frame, expected steps)
        checked = 0

        def collect(slot):
            nonlocal checked
            out, want, steps = in_flight.pop(slot)
            info = m.render_wait(slot)
            got = out if isinstance(out, np.ndarray) else out.cpu().numpy()
            assert (got == want).all(), f"seed {seed}: multi slot {slot} differs from the frame at submit time"
            assert int(info.cubes_traced) == steps
            checked += 1

        for _ in range(N_CALLS // 2):
            op = rng.random()
            if op < 0.5:
                slot = int(rng.integers(0, 8))
                if slot in in_flight:
                    collect(slot)
                w, h = SIZES[int(rng.integers(0, len(SIZES)))]
                f = abi.Context.make_frame(w, h, world_inv=cameras(rng, w, h, opts["view_distance"]))
                ref = b.render(f)
                if rng.random() < 0.5:
                    buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    m.render_submit(f, slot, buf.data_ptr())
                else:
                    buf = m.render_submit(f, slot)
                in_flight[slot] = (buf, ref["rgba8"].copy(), int(ref["info"].cubes_traced))
            elif op < 0.65 and in_flight:
                collect(int(rng.choice(list(in_flight))))
            elif op < 0.78:
                n = int(rng.integers(1, 12))
                xyz = np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n), rng.integers(0, 16, n)], 1).astype(np.int32)
                blocks = rng.integers(0, n_blocks, n).astype(np.uint16)
                light = rng.integers(0, 256, (n, 4)).astype(np.uint8)
                light[:, 3] = rng.choice([1, 128, 255], n)
                for c in (m, b):
                    c.update_cubes(abi.LAYER_WORLD, xyz, blocks, light)
            elif op < 0.86:
                light = rng.integers(0, 256, tuple(space.size) + (4,)).astype(np.uint8)
                light[..., 3] = rng.choice([1, 128, 255, 255], tuple(space.size))
                for c in (m, b):
                    c.update_light_volume(abi.LAYER_WORLD, light)
            elif op < 0.93:  # blocks placed and the light brought up to date on the device (blocking; the texels reach the other devices)
                n = int(rng.integers(1, 5))
                xyz = np.stack([rng.integers(1, 15, n), rng.integers(1, 15, n), rng.integers(1, 15, n)], 1).astype(np.int32)
                blocks = rng.integers(0, n_blocks, n).astype(np.uint16)
                for c in (m, b):
                    c.update_cubes(abi.LAYER_WORLD, xyz, blocks)
                    c.light_cubes_changed(abi.LAYER_WORLD, xyz)
                im = m.evaluate_light(abi.LAYER_WORLD, 6, fast=False)
                ib = b.evaluate_light(abi.LAYER_WORLD, 6, fast=False)
                assert im.updates == ib.updates
            else:
                opts.update(fog=int(rng.integers(0, 4)), lighting=int(rng.integers(0, 5)))
                for c in (m, b):
                    c.set_options(abi.LAYER_WORLD, abi.make_options(**opts))
        for slot in list(in_flight):
            collect(slot)
        assert checked > 5
    finally:
        m.close()
        b.close()
