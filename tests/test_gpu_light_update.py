"""GPU tests (-m gpu) of the light updater on the device: `aic_evaluate_light` against the oracle's
restatement of `Mutation::{fast_evaluate_light, evaluate_light}` -- the converged light volume byte for byte, the update
count and the summed cost -- and, end to end (uninitialised space -> device light propagation -> device raytrace), against
the reference's golden PNGs of its lighting image tests."""
import copy
import ctypes
import ctypes.util

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi
from all_is_cubes_amd import flat, workloads
from tests import scenes
from tests.test_gpu_parity import to_abi_options
from tests.test_oracle_goldens import COMMON_VIEWPORT
from tests.test_oracle_light import EXACT, LIGHTING, image_diff, lit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


SCENES = {
    "light_spread": scenes.light_spread_space,
    "light_on_slab": scenes.light_on_slab_space,
    "fog": scenes.fog_test_space,
    "tone_mapping": scenes.tone_mapping_space,
}


def test_device_log2f_is_libm_log2f(ctx):
    """PackedLight::scalar_in (data.rs:214-218) takes f32::log2 = libm's log2f: the device's restatement equals it bit for bit."""
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.log2f.restype = ctypes.c_float
    libm.log2f.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(5)
    bits = np.concatenate([rng.integers(1, 0x7f800000, 200000, dtype=np.uint32),
                           np.array([0, 1, 0x007fffff, 0x00800000, 0x3f7fffff, 0x3f800000, 0x3f800001, 0x7f7fffff, 0x7f800000], np.uint32),
                           (np.float32(2.0) ** np.arange(-30, 30, dtype=np.float32)).view(np.uint32)])
    x = bits.view(np.float32)
    got = ctx.probe_log2f(x)
    want = np.array([libm.log2f(float(v)) for v in x[:20000]], np.float32)
    assert (got[:20000].view(np.uint32) == want.view(np.uint32)).all()
    # and the whole sample against numpy where numpy's own log2 agrees with libm on the checked prefix
    tail = np.array([libm.log2f(float(v)) for v in x[-69:]], np.float32)
    assert (got[-69:].view(np.uint32) == tail.view(np.uint32)).all()


@pytest.mark.parametrize("name", list(SCENES) + ["synthetic"])
def test_derived_matches_oracle(ctx, name):
    sp = SCENES[name]() if name in SCENES else workloads.synthetic_space(n=8, resolution=8, n_blocks=12, seed=4)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    got, got_opaque = ctx.probe_derived(abi.LAYER_WORLD, len(sp.blocks))
    want = oracle.compute_derived(oracle.Space(sp))
    n = len(sp.blocks)
    assert (got[:, 0:4].view(np.uint32) == want["color"].view(np.uint32)).all()
    assert (got[:, 4:28].reshape(n, 6, 4).view(np.uint32) == want["face_colors"].view(np.uint32)).all()
    assert (got[:, 28:31].view(np.uint32) == want["emission"].view(np.uint32)).all()
    assert ((got[:, 31] != 0) == want["visible"]).all()
    assert ((got_opaque != 0) == want["opaque"]).all()


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("batch,order,lanes", [(32, 16, 64), (1, 16, 64), (32, 8, 64), (7, 0, 64), (32, 16, 1), (32, 16, 256)])
def test_evaluate_light_matches_oracle(ctx, name, batch, order, lanes):
    """`lanes`: one wave per cube (64 lanes walk slices of the tree, contributions added in the reference's order) or the
    plain one-lane-per-cube restatement: both must give the oracle's bytes."""
    if name in ("fog", "tone_mapping") and (batch, order) != (32, 16):
        pytest.skip("the larger scenes run in the reference configuration only")
    sp = SCENES[name]()
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=30, fast=True, epsilon=1, batch=batch, hb_width=order)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=batch, queue_order=order, lanes_per_cube=lanes)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref
    assert (got == np.asarray(ref.light).reshape(got.shape)).all(), f"{(got != np.asarray(ref.light).reshape(got.shape)).any(axis=-1).sum()} texels differ"
    assert info.queue_left >= 0 and info.batches >= (n_ref + batch - 1) // batch


def test_relight_after_a_change(ctx):
    """Not `fast`: start from a converged volume, change a block, queue what modified_cube_needs_update would
    (updater.rs:135-173: the cube at NEWLY_VISIBLE and its neighbours), and run to convergence again."""
    sp = copy.deepcopy(lit(scenes.light_spread_space))
    lo, size = np.array(sp.lo), np.array(sp.size)
    cube = tuple(int(v) for v in lo + size // 2 + np.array([1, 0, 1]))
    rel = tuple(c - l for c, l in zip(cube, sp.lo))
    old = int(sp.block_index[rel])
    new = next(i for i in range(len(sp.blocks)) if i != old)
    sp.block_index[rel] = new
    queue = [(cube, 250)]
    for f in range(6):
        nb = list(cube)
        nb[f % 3] += 1 if f >= 3 else -1
        queue.append((tuple(nb), 250))
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=30, fast=False, epsilon=1, batch=32, queue=queue, hb_width=16)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=32, queue_order=16, queue=queue)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref and n_ref > 0
    assert (got == np.asarray(ref.light).reshape(got.shape)).all()


def test_uninitialized_queue_and_update_limit(ctx):
    """queue=None enqueues every Uninitialized texel at Priority::UNINIT; max_updates cuts the run short at the same place."""
    sp = scenes.light_on_slab_space()
    sp.light[...] = 0  # PackedLight::UNINITIALIZED_AND_BLACK everywhere
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=12, fast=False, epsilon=1, batch=32, queue=None, max_updates=200, hb_width=16)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 12, fast=False, epsilon=1, batch=32, queue_order=16, queue=None, max_updates=200)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref == 224  # whole batches of 32
    assert (got == np.asarray(ref.light).reshape(got.shape)).all()


@pytest.mark.parametrize("batch", [32, 100])
def test_dependency_pool_grows_when_a_batch_overflows_it(batch):
    """The dependency lists go through a chunk pool on the device; a batch that runs out of chunks is computed again with a
    larger pool (aic_evaluate_light). A fresh context started with a four-chunk pool (test hook) must still give the
    oracle's bytes -- with the small-batch path (cubes and texels in the launch arguments) and the copying path."""
    sp = scenes.light_spread_space()
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=30, fast=True, epsilon=1, batch=batch, hb_width=16)
    c = abi.Context(0)
    try:
        c.upload_space(abi.LAYER_WORLD, sp)
        info = c.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=batch, queue_order=16, dep_pool_chunks=4)
        got = c.read_light_volume(abi.LAYER_WORLD, sp.size)
    finally:
        c.close()
    assert info.updates == n_ref
    assert (got == np.asarray(ref.light).reshape(got.shape)).all()


@pytest.mark.parametrize("name,batch,lanes,pool", [("light_spread", 32, 256, 0), ("light_on_slab", 7, 64, 0), ("fog", 32, 256, 0), ("light_spread", 32, 256, 4)])
def test_session_kernel_gives_the_same_bytes(name, batch, lanes, pool):
    """aic_light_params.hooks bit 0 (AIC_LIGHT_SESSION=1 until round 5): one launch serves every small batch of a call, fed through pinned host memory (an experiment kept
    reproducible, csrc/aic_light.h LightMailbox). Same texels, same update count -- also when the dependency pool overflows
    and the session has to end, grow the pool and start again, and in a relight that continues the queue."""
    sp = SCENES[name]()
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=30, fast=True, epsilon=1, batch=batch, hb_width=16)
    c = abi.Context(0)
    try:
        c.upload_space(abi.LAYER_WORLD, sp)
        info = c.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=batch, queue_order=16, lanes_per_cube=lanes, session=True, dep_pool_chunks=pool)
        got = c.read_light_volume(abi.LAYER_WORLD, sp.size)
        assert info.updates == n_ref and info.device_ms > 0
        assert (got == np.asarray(ref.light).reshape(got.shape)).all()
        # a second call on the same context: another session, the queue continued (nothing left: no updates, no hang)
        again = c.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=batch, queue_order=16, queue=[], lanes_per_cube=lanes, session=True)
        assert again.updates == 0
        assert (c.read_light_volume(abi.LAYER_WORLD, sp.size) == got).all()
    finally:
        c.close()


def test_large_batches_converge_to_the_same_light(ctx):
    """Throughput mode: thousands of queue entries per launch. The order of updates differs from the reference's, so the
    converged texels may differ in their last log-scale unit (the reference's own order is unspecified, queue.rs:236-243)."""
    sp = scenes.fog_test_space()
    ref = lit(scenes.fog_test_space)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=4096, queue_order=0)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size).astype(int)
    want = np.asarray(ref.light).reshape(got.shape).astype(int)
    assert (got[..., 3] == want[..., 3]).all()  # status
    d = np.abs(got[..., :3] - want[..., :3])
    print(f"batch 4096: {info.updates} updates in {info.batches} launches, device {info.device_ms:.1f} ms, total {info.total_ms:.1f} ms; "
          f"texel difference histogram {np.bincount(d.max(axis=-1).ravel())}")
    assert d.max() <= 3 and (d > 1).mean() < 0.01


@pytest.mark.parametrize("option", ["Flat", "Linear"])
def test_end_to_end_light_on_slab_golden(ctx, golden_dir, option):
    """Uninitialised space -> aic_evaluate_light -> aic_render, all on the device, against the reference's golden image."""
    sp = scenes.light_on_slab_space()
    opt = oracle.unaltered_colors(lighting=LIGHTING[option])
    w, h = COMMON_VIEWPORT
    eye, look = (0.5, -6.0, 6.0), (0.0, 1.0, -1.0)
    q = oracle.look_at_y_up(eye, tuple(e + l for e, l in zip(eye, look)))
    _, _, inv = oracle.camera_matrices(45.0, opt.view_distance, w / h, q, eye)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=32, queue_order=16)
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv))
    name = f"light_on_slab-{option}-all"
    assert name in EXACT
    assert image_diff(golden_dir, name, got["rgba8"]).max() <= 7
    assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / f"png_{name}.npy").astype(int)).max() <= 1


def test_end_to_end_light_spread_golden(ctx, golden_dir):
    sp = scenes.light_spread_space()
    opt = oracle.unaltered_colors(lighting=3)
    w, h = COMMON_VIEWPORT
    eye, look = (0.0, 0.0, 8.0), (0.0, 0.0, -1.0)
    q = oracle.look_at_y_up(eye, tuple(e + l for e, l in zip(eye, look)))
    _, _, inv = oracle.camera_matrices(45.0, opt.view_distance, w / h, q, eye)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=True, epsilon=1, batch=32, queue_order=16)
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv))
    name = "light_spread-Linear-all"
    assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / f"png_{name}.npy").astype(int)).max() <= 1


def test_evaluate_light_rejects_bad_arguments(ctx):
    sp = scenes.light_on_slab_space()
    ctx.upload_space(abi.LAYER_WORLD, sp)
    with pytest.raises(abi.AicError):
        ctx.evaluate_light(abi.LAYER_WORLD, 300)
    with pytest.raises(abi.AicError):
        ctx.evaluate_light(abi.LAYER_WORLD, 30, queue_order=5)
    with pytest.raises(abi.AicError):
        ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, queue=[((0, 0, 0), 999)])
    ctx.clear_space(abi.LAYER_UI)
    with pytest.raises(abi.AicError):
        ctx.evaluate_light(abi.LAYER_UI, 30)


@pytest.mark.parametrize("maximum_distance", [3, 127, 255])
def test_other_maximum_distances(ctx, maximum_distance):
    """The effective tree, its LDS bitmaps (86 KB at the full chart) and the slot arrays are sized per maximum_distance."""
    sp = scenes.light_on_slab_space()
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=maximum_distance, fast=True, epsilon=1, batch=32, max_updates=600, hb_width=16)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, maximum_distance, fast=True, epsilon=1, batch=32, queue_order=16, max_updates=600)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref
    assert (got == np.asarray(ref.light).reshape(got.shape)).all()


def test_multi_device_light_evaluation(ctx, golden_dir):
    """aic_multi_evaluate_light: the updater runs on the first device and the volume is handed to the others, so that the
    strips every device traces carry the same light: the assembled frame equals the single-context frame and the golden."""
    sp = scenes.light_on_slab_space()
    opt = oracle.unaltered_colors(lighting=3)
    w, h = COMMON_VIEWPORT
    eye, look = (0.5, -6.0, 6.0), (0.0, 1.0, -1.0)
    q = oracle.look_at_y_up(eye, tuple(e + l for e, l in zip(eye, look)))
    _, _, inv = oracle.camera_matrices(45.0, opt.view_distance, w / h, q, eye)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    one_info = ctx.evaluate_light(abi.LAYER_WORLD, 30)
    one = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    with abi.MultiContext([0, 0, 0]) as m:
        m.upload_space(abi.LAYER_WORLD, sp)
        m.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        info = m.evaluate_light(abi.LAYER_WORLD, 30)
        many = m.render(abi.Context.make_frame(w, h, world_inv=inv))["rgba8"]
    assert info.updates == one_info.updates
    assert (one == many).all()
    assert np.abs(many.astype(int) - np.load(golden_dir / "png_light_on_slab-Linear-all.npy").astype(int)).max() <= 1


def test_multi_device_cubes_changed_reaches_every_device(ctx):
    """aic_multi_light_cubes_changed: the OPAQUE texels written on the device that runs the updater are scattered into the other
    devices' volumes (aic_read_light_cubes + aic_update_cubes), so the strips of the next frame carry the same light -- with
    no evaluate call in between -- and equal the single-context frame. aic_read_light_cubes itself against the volume."""
    sp = scenes.light_on_slab_space()
    opt = oracle.unaltered_colors(lighting=3)
    w, h = 96, 72
    eye, look = (0.5, -6.0, 6.0), (0.0, 1.0, -1.0)
    q = oracle.look_at_y_up(eye, tuple(e + l for e, l in zip(eye, look)))
    _, _, inv = oracle.camera_matrices(45.0, opt.view_distance, w / h, q, eye)
    d = oracle.compute_derived(oracle.Space(sp))
    wall = next(i for i in range(len(d["opaque"])) if d["opaque"][i].all() and not d["emission"][i].any())
    air = [tuple(int(v) + sp.lo[a] for a, v in enumerate(c)) for c in np.argwhere(sp.block_index == 0)]
    rng = np.random.default_rng(5)
    near = [c for c in air if abs(c[0]) <= 5 and abs(c[1]) <= 5 and c[2] <= 2]  # in view
    xyz = np.array([near[i] for i in rng.choice(len(near), 12, replace=False)], np.int32)
    bi = np.full(len(xyz), wall, np.uint16)

    def run(c):
        c.upload_space(abi.LAYER_WORLD, sp)
        c.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        c.evaluate_light(abi.LAYER_WORLD, 30)
        before = c.render(abi.Context.make_frame(w, h, world_inv=inv))["rgba8"]
        c.update_cubes(abi.LAYER_WORLD, xyz, block_index=bi)
        c.light_cubes_changed(abi.LAYER_WORLD, xyz)
        return before, c.render(abi.Context.make_frame(w, h, world_inv=inv))["rgba8"]

    ctx.clear_space(abi.LAYER_UI)
    one_before, one_after = run(ctx)
    volume = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    probe = np.concatenate([xyz, np.array(air[:40], np.int32)])
    rel = probe - np.array(sp.lo, np.int32)
    assert (ctx.read_light_cubes(abi.LAYER_WORLD, probe) == volume[rel[:, 0], rel[:, 1], rel[:, 2]]).all()
    assert (ctx.read_light_cubes(abi.LAYER_WORLD, xyz) == np.array([0, 0, 0, 128], np.uint8)).all()  # PackedLight::OPAQUE
    with pytest.raises(abi.AicError):
        ctx.read_light_cubes(abi.LAYER_WORLD, [[sp.lo[0] - 1, sp.lo[1], sp.lo[2]]])
    with abi.MultiContext([0, 0, 0]) as m:
        many_before, many_after = run(m)
    assert (one_before == many_before).all() and (one_after == many_after).all()
    assert (one_after != one_before).any()


def _modified_cubes_queue(sp, changes):
    """modified_cube_needs_update (updater.rs:135-173) for [(cube, new block index)] applied to `sp` in order, in Python:
    patches sp.block_index / sp.light and returns the queue entries [(cube, priority)] in insertion order."""
    d = oracle.compute_derived(oracle.Space(sp))
    opaque, emission = d["opaque"], d["emission"]
    lo, size = sp.lo, sp.size

    def inb(c):
        return all(lo[a] <= c[a] < lo[a] + size[a] for a in range(3))

    queue = []
    for cube, index in changes:
        sp.set(cube, index)
        rel = tuple(cube[a] - lo[a] for a in range(3))
        if opaque[index].all() and not emission[index].any():  # opaque_for_light_computation
            sp.light[rel] = (0, 0, 0, 128)
            queue = [(c, p) for c, p in queue if c != tuple(cube)]
        else:
            queue.append((tuple(cube), 250))
        for f in range(6):
            nb = list(cube)
            nb[f % 3] += 1 if f >= 3 else -1
            if not inb(nb):
                continue
            nb_block = int(sp.block_index[tuple(nb[a] - lo[a] for a in range(3))])
            if not opaque[nb_block][(f + 3) % 6]:  # the neighbour's face towards the cube
                queue.append((tuple(nb), 250))
    return queue


def test_cubes_changed_queue_and_budgeted_relight(ctx):
    """The layer's own update queue: aic_update_cubes + aic_light_cubes_changed queue what Mutation::set would, and
    aic_evaluate_light(fast=0, no new entries) drains it -- in one call, or in budgeted slices (a per-frame light budget)
    with the same result, which is the oracle's for the same queue."""
    base = copy.deepcopy(lit(scenes.light_spread_space))
    lo, size = np.array(base.lo), np.array(base.size)
    centre = lo + size // 2
    air = next(i for i, b in enumerate(base.blocks) if b.is_air)
    solid = next(i for i, b in enumerate(base.blocks) if not b.is_air)
    c1 = tuple(int(v) for v in centre + np.array([1, 0, 1]))
    c2 = tuple(int(v) for v in centre + np.array([-1, 0, 1]))
    changes = []
    for cube in (c1, c2):
        old = int(base.block_index[tuple(cube[a] - base.lo[a] for a in range(3))])
        changes.append((cube, solid if old == air else air))
    ref = copy.deepcopy(base)
    queue = _modified_cubes_queue(ref, changes)
    n_ref = oracle.evaluate_light(ref, maximum_distance=30, fast=False, epsilon=1, batch=32, queue=queue, hb_width=16)
    want = np.asarray(ref.light)

    for budget in (0, 48):
        ctx.upload_space(abi.LAYER_WORLD, base)
        xyz = np.array([c for c, _ in changes], np.int32)
        ctx.update_cubes(abi.LAYER_WORLD, xyz, block_index=np.array([i for _, i in changes], np.uint16))
        ctx.light_cubes_changed(abi.LAYER_WORLD, xyz)
        total, calls = 0, 0
        while True:
            info = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=32, queue_order=16, queue=[], max_updates=budget)
            total += info.updates
            calls += 1
            if info.queue_left == 0 or calls > 10000:
                break
        got = ctx.read_light_volume(abi.LAYER_WORLD, base.size)
        assert total == n_ref and n_ref > 0
        assert (got == want.reshape(got.shape)).all()
        if budget:
            assert calls > 1


def test_host_mirror_device_light_mode(ctx):
    """HipRtRenderer.device_light: update() forwards block changes only and queues them for relighting; evaluate_light
    (fresh, then continuing the queue) gives the frames the C ABI path gives for the same operations."""
    import all_is_cubes_amd as A
    from all_is_cubes_amd import _host as H

    sp = scenes.light_on_slab_space()
    w, h = 128, 96
    eye, look = (0.5, -6.0, 6.0), (0.0, 1.0, -1.0)
    target = tuple(e + l for e, l in zip(eye, look))
    cams = H.StandardCameras()
    o = H.GraphicsOptions()  # default(): Volumetric, Linear lighting, Abrupt fog, fov 90, view distance 200
    o.bloom_intensity = 0.0
    cams.graphics_options = o
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    space = A.space_from_flat(sp)
    cams.world_space = space
    cams.world_view_transform = H.look_at_y_up(eye, target)
    r = H.HipRtRenderer(cams)
    r.device_light = True
    r.update()
    r.evaluate_light(30)
    img_a = np.array(r.draw("").data)

    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, target), eye)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(bloom_intensity=0.0))
    ctx.evaluate_light(abi.LAYER_WORLD, 30)
    img_b = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    assert (img_a == img_b).all()

    # a block change: through Space.set + update() on one side, update_cubes + light_cubes_changed on the other
    cube = (0, 0, 2)
    wall = 1
    space.set(cube[0], cube[1], cube[2], wall)
    r.update()
    r.evaluate_light(30, fast=False, continue_queue=True)
    img_a2 = np.array(r.draw("").data)
    xyz = np.array([cube], np.int32)
    ctx.update_cubes(abi.LAYER_WORLD, xyz, block_index=np.array([wall], np.uint16))
    ctx.light_cubes_changed(abi.LAYER_WORLD, xyz)
    ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, queue=[])
    img_b2 = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    assert (img_a2 == img_b2).all()
    assert (img_a2 != img_a).any()


def _random_light_scene(seed):
    """A small space with every kind of block the updater distinguishes: air, opaque and translucent atoms, an opaque
    emitter, a translucent emitter, recursive blocks (full, partial, translucent, emissive voxels), an octant sky, at random."""
    rng = np.random.default_rng(1000 + seed)
    size = tuple(int(v) for v in rng.integers(3, 9, 3))
    lo = tuple(int(v) for v in rng.integers(-5, 5, 3))
    sp = flat.FlatSpace(lo, size)
    if seed % 3 == 0:
        sp.set_sky_octants(rng.uniform(0.0, 2.0, (8, 3)).astype(np.float32))
    else:
        sp.set_sky_uniform(tuple(float(v) for v in rng.uniform(0.0, 1.5, 3)))
    ids = [sp.add_block(flat.air())]
    ids.append(sp.add_block(flat.atom((0.8, 0.7, 0.6, 1.0))))
    ids.append(sp.add_block(flat.atom((0.2, 0.5, 0.9, 0.5))))
    ids.append(sp.add_block(flat.atom((1.0, 1.0, 1.0, 1.0), (4.0, 2.0, 0.5))))
    ids.append(sp.add_block(flat.atom((0.5, 0.5, 0.5, 0.25), (0.0, 1.5, 3.0))))
    for b in workloads.synthetic_blocks(int(rng.choice([2, 4, 8])), 8, seed=seed + 1, palette_size=6):
        ids.append(sp.add_block(b))
    fill = rng.uniform(0.15, 0.6)
    choice = rng.integers(1, len(ids), size)
    sp.block_index[...] = np.where(rng.random(size) < fill, choice, 0).astype(np.uint16)
    sp.light[...] = 0
    return sp


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AIC_LIGHT_FUZZ_N", "16"))))
def test_light_updater_fuzz(ctx, seed):
    """Randomised differential test of the whole updater (derived block properties, fast_evaluate_light, queue order,
    compute_light, apply): device == oracle, texel for texel, with a random distance, batch and order."""
    sp = _random_light_scene(seed)
    rng = np.random.default_rng(seed)
    maxd = int(rng.choice([2, 5, 12, 30]))
    batch = int(rng.choice([1, 5, 32, 100]))
    order = int(rng.choice([0, 8, 16]))
    lanes = int(rng.choice([1, 64, 256]))
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=maxd, fast=True, epsilon=1, batch=batch, hb_width=order)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    got_d, got_o = ctx.probe_derived(abi.LAYER_WORLD, len(sp.blocks))
    want = oracle.compute_derived(oracle.Space(sp))
    assert (got_d[:, 0:4].view(np.uint32) == want["color"].view(np.uint32)).all() and ((got_o != 0) == want["opaque"]).all()
    info = ctx.evaluate_light(abi.LAYER_WORLD, maxd, fast=True, epsilon=1, batch=batch, queue_order=order, lanes_per_cube=lanes)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref
    assert (got == np.asarray(ref.light).reshape(got.shape)).all(), f"seed {seed}: {(got != np.asarray(ref.light).reshape(got.shape)).any(axis=-1).sum()} texels differ"


def test_light_update_on_the_worker_thread_equals_the_blocking_call():
    """aic_evaluate_light_submit / aic_evaluate_light_wait (ABI 3): a sim + render loop of 120 steps -- a lamp block placed or removed every ten steps, a budget of
    cube updates per step, frames streamed on four slots -- run twice: with the blocking aic_evaluate_light in front of each frame, and with the update on the
    library's worker thread BESIDE the frame. At every step the two light volumes are byte for byte equal once the update is collected (same update
    counts, same queue left), a frame submitted while the update runs shows the light as it stood (the blocking run's frame of the step before the
    update), and scene calls made while an update is pending (aic_update_cubes, aic_light_cubes_changed, aic_read_light_volume) publish it first."""
    import torch

    sp = copy.deepcopy(lit(scenes.light_spread_space))
    lo, size = np.array(sp.lo), np.array(sp.size)
    rng = np.random.default_rng(3)
    air = next(i for i, b in enumerate(sp.blocks) if b.is_air)
    lamp = next(i for i, b in enumerate(sp.blocks) if not b.is_air)
    sites = []
    bi = np.asarray(sp.block_index)
    while len(sites) < 6:
        c = rng.integers(1, size - 1)
        if int(bi[tuple(c)]) == air and tuple(c) not in sites:
            sites.append(tuple(int(v) for v in c))
    w, h = 320, 200
    eye = tuple(float(v) for v in lo + size * np.array([0.5, 0.5, 1.6]))
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, tuple(float(v) for v in lo + size / 2)), eye)
    budget = 64
    steps = 120

    def run(blocking):
        volumes, infos, frames = [], [], []
        with abi.Context(0) as c:
            c.upload_space(abi.LAYER_WORLD, sp)
            c.set_options(abi.LAYER_WORLD, abi.make_options())
            fr = c.make_frame(w, h, world_inv=inv)
            bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
            in_flight = {}
            pending = False
            for i in range(steps):
                if not blocking and pending:
                    infos.append(c.evaluate_light_wait(abi.LAYER_WORLD))      # the previous step's update: collected, published
                    volumes.append(c.read_light_volume(abi.LAYER_WORLD, sp.size))
                if i % 10 == 0:
                    x, y, z = sites[(i // 10) % 6]
                    cube = [(int(lo[0] + x), int(lo[1] + y), int(lo[2] + z))]
                    c.update_cubes(abi.LAYER_WORLD, cube, [lamp if (i // 60) % 2 == 0 else air])
                    c.light_cubes_changed(abi.LAYER_WORLD, cube, queue_order=0)
                if blocking:
                    slot = i % 4
                    if slot in in_flight:
                        c.render_wait(slot)
                        frames.append(in_flight.pop(slot).cpu().numpy().copy())
                    # (the frame of the step BEFORE this step's update: what the other run's frame, submitted beside the update, must show)
                    c.render_submit(fr, bufs[slot].data_ptr(), slot)
                    in_flight[slot] = bufs[slot]
                    infos.append(c.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=budget, queue_order=0, queue=[], max_updates=budget))
                    volumes.append(c.read_light_volume(abi.LAYER_WORLD, sp.size))
                else:
                    c.evaluate_light_submit(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=budget, queue_order=0, queue=[], max_updates=budget)
                    pending = True
                    slot = i % 4
                    if slot in in_flight:
                        c.render_wait(slot)
                        frames.append(in_flight.pop(slot).cpu().numpy().copy())
                    c.render_submit(fr, bufs[slot].data_ptr(), slot)           # beside the update: the light as it stood
                    in_flight[slot] = bufs[slot]
            if not blocking:
                infos.append(c.evaluate_light_wait(abi.LAYER_WORLD))
                volumes.append(c.read_light_volume(abi.LAYER_WORLD, sp.size))
                assert c.evaluate_light_wait(abi.LAYER_WORLD).updates == 0   # nothing submitted: zeros
            for slot in sorted(in_flight):
                c.render_wait(slot)
        return volumes, infos, frames

    v0, i0, f0 = run(True)
    v1, i1, f1 = run(False)
    assert len(v0) == len(v1) == steps and sum(i.updates for i in i0) > 100
    for k in range(steps):
        assert i0[k].updates == i1[k].updates and i0[k].queue_left == i1[k].queue_left, k
        assert (v0[k] == v1[k]).all(), f"light volume after step {k}"
    assert len(f0) == len(f1) == steps - 4
    for k, (a, b) in enumerate(zip(f0, f1)):
        assert (a == b).all(), f"frame {k}: a frame submitted beside the update did not see the light as it stood"
    assert any((v0[k] != v0[k - 1]).any() for k in range(1, steps)), "the loop never changed the light"


def test_light_update_beside_frames_in_flight(ctx):
    """aic_evaluate_light no longer waits for the frames in flight (round 4): it works on the other half of the light double buffer.
    A frame submitted BEFORE the update shows the old light whatever the update does meanwhile; a frame rendered after it shows the
    new light; and the new volume is byte for byte what the same update gives with nothing in flight."""
    import torch

    sp = copy.deepcopy(lit(scenes.light_spread_space))
    lo, size = np.array(sp.lo), np.array(sp.size)
    cube = tuple(int(v) for v in lo + size // 2 + np.array([1, 0, 1]))
    rel = tuple(c - l for c, l in zip(cube, sp.lo))
    new = next(i for i in range(len(sp.blocks)) if i != int(sp.block_index[rel]))
    queue = [(cube, 250)] + [(tuple(int(cube[a] + (d if a == k else 0)) for a in range(3)), 250) for k in range(3) for d in (-1, 1)]
    w, h = 480, 360
    eye = tuple(float(v) for v in lo + size * np.array([0.5, 0.5, 1.6]))
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, tuple(float(v) for v in lo + size / 2)), eye)
    # reference run: nothing in flight
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options())
    fr = ctx.make_frame(w, h, world_inv=inv)
    before = ctx.render(fr)["rgba8"]
    ctx.update_cubes(abi.LAYER_WORLD, [cube], [new])
    info0 = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=32, queue_order=16, queue=queue)
    want_volume = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    after = ctx.render(fr)["rgba8"]
    assert info0.updates > 0 and (after != before).any()
    # the same with two frames in flight while the update runs
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.update_cubes(abi.LAYER_WORLD, [cube], [new])
    mid = ctx.render(fr)["rgba8"]  # the block has changed, the light has not
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    ctx.render_submit(fr, bufs[0].data_ptr(), 1)
    ctx.render_submit(fr, bufs[1].data_ptr(), 2)
    info1 = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=32, queue_order=16, queue=queue)
    ctx.render_submit(fr, bufs[2].data_ptr(), 3)
    for slot in (1, 2, 3):
        ctx.render_wait(slot)
    assert info1.updates == info0.updates
    assert (bufs[0].cpu().numpy() == mid).all() and (bufs[1].cpu().numpy() == mid).all(), "a frame in flight saw the update"
    assert (bufs[2].cpu().numpy() == after).all()
    assert (ctx.read_light_volume(abi.LAYER_WORLD, sp.size) == want_volume).all()
    # and once more: the halves swap back and forth
    info2 = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=1, batch=32, queue_order=16, queue=[])
    assert info2.updates == 0 and (ctx.render(fr)["rgba8"] == after).all()
