"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, and against the reference's golden vectors.

Bar: bit-exact for integer results (hit cube / voxel / face / block index, step counts,
f64 t-distances); RGBA8 within +-1 LSB (f32 colour math goes through powf/exp whose last bit
is libm-dependent; the reference itself tolerates 1-2 levels: cases/src/lib.rs:347,1233)."""
from pathlib import Path

import os

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat
from tests import scenes
from tests.test_oracle_goldens import COMMON_VIEWPORT, camera_for, diff_to, neighbourhood_diff

pytestmark = pytest.mark.gpu

RGBA_TOL = 1  # stated float tolerance, in 8-bit levels


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


def to_abi_options(o: oracle.OrcOptions) -> abi.Options:
    return abi.make_options(fog=o.fog, transparency=o.transparency, threshold=o.threshold, lighting=o.lighting, bounce_samples=o.bounce_samples,
                            antialiasing=o.antialiasing, debug_pixel_cost=bool(o.debug_pixel_cost), tone_mapping=o.tone_mapping,
                            maximum_intensity=o.maximum_intensity, bloom_intensity=0.0, view_distance=o.view_distance)


def assert_production_variants(ctx, frame, got):
    """The production kernel variants of `frame` -- the plain one an image this small gets and the lane-exchanging one (asked for through
    aic_frame_desc.tuning) -- must give the aux-recording variant's bytes and step total; aic_frame_info.variant says which ran."""
    keep_flags, keep_tuning = frame.flags, frame.tuning
    frame.flags &= ~(abi.FRAME_AUX | abi.FRAME_COUNTERS)
    try:
        fast = ctx.render(frame)
        frame.tuning = keep_tuning | abi.tuning(variant=abi.VARIANT_EXCHANGING)
        exchanged = ctx.render(frame)
    finally:
        frame.flags, frame.tuning = keep_flags, keep_tuning
    if got["info"].rows_rendered and frame.width:
        assert got["info"].variant == abi.VARIANT_RECORDING or not (keep_flags & (abi.FRAME_AUX | abi.FRAME_COUNTERS))
        assert exchanged["info"].variant in (abi.VARIANT_EXCHANGING, abi.VARIANT_PLAIN)  # (Bounce lighting has no exchanging variant)
        assert fast["info"].variant in (abi.VARIANT_PLAIN, abi.VARIANT_EXCHANGING)
    for name, other in (("plain", fast), ("exchanging", exchanged)):
        assert (other["rgba8"] == got["rgba8"]).all() and other["info"].cubes_traced == got["info"].cubes_traced, f"production variant ({name}) differs from the aux-recording one"


def render_both(ctx, space, opt: oracle.OrcOptions, size, eye, quat=(0, 0, 0, 1), fov=90.0, ui=None, ui_eye=(0, 0, 0), backdrop=(0, 0, 0, 0), tuning=0):
    w, h = size
    _, _, inv = oracle.camera_matrices(fov, opt.view_distance, w / h, quat, eye)
    ui_inv = None
    if space is not None:
        ctx.upload_space(abi.LAYER_WORLD, space)
    else:
        ctx.clear_space(abi.LAYER_WORLD)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    if ui is not None:
        _, _, ui_inv = oracle.camera_matrices(fov, opt.view_distance, w / h, (0, 0, 0, 1), ui_eye)
        ctx.upload_space(abi.LAYER_UI, ui)
        ctx.set_options(abi.LAYER_UI, to_abi_options(opt))
    else:
        ctx.clear_space(abi.LAYER_UI)
    opt.exposure = 1.0
    frame = ctx.make_frame(w, h, world_inv=inv, ui_inv=ui_inv, backdrop=backdrop, tuning=tuning)
    got = ctx.render(frame, want_aux=True)
    # every scene that goes through this helper (goldens, options matrix, layers, antialiasing, UI pre-pass + world pass) covers the production variants too
    assert_production_variants(ctx, frame, got)
    ref = oracle.render(
        oracle.Space(space) if space is not None else None, opt, oracle.make_camera(inv, w, h),
        ui=oracle.Space(ui) if ui is not None else None, ui_opt=opt if ui is not None else None,
        ui_cam=oracle.make_camera(ui_inv, w, h) if ui is not None else None, backdrop=backdrop, want_aux=True,
    )
    return got, ref


def assert_parity(got, ref, tol=RGBA_TOL):
    ga, ra = got["aux"], ref["aux"]
    assert (ga["cubes_traced"] == ra["cubes_traced"]).all(), "per-pixel step counts differ"
    assert got["info"].cubes_traced == int(ref["info"]["cubes_traced"])
    for k in ("hit", "cube", "voxel", "resolution", "face", "block_index"):
        assert (ga[k] == ra[k]).all(), f"first-hit {k} differs"
    hit = ra["hit"] == 1
    assert (ga["t_distance"][hit].view(np.uint64) == ra["t_distance"][hit].view(np.uint64)).all(), "hit t_distance bits differ"
    for k in ("n_outer", "n_inner", "n_hits", "n_light"):
        assert getattr(got["info"], k) == int(ref["info"][k]), k
    d = np.abs(got["rgba8"].astype(np.int16) - ref["rgba8"].astype(np.int16))
    assert d.max() <= tol, f"RGBA8 max diff {d.max()}"
    return int(d.max())


# --- device Raycaster vs the reference's known answers (raycast/tests.rs) -----------------
def test_probe_raycast_kats(ctx):
    steps, ended = ctx.probe_raycast((0.0, -0.25, -0.5), (1.0, 1.0, 1.0), bounds=((2, -10, -10), (4, 10, 10)))
    assert ended
    assert [(tuple(s["cube"]), int(s["face"]), float(s["t_distance"])) for s in steps] == [
        ((2, 1, 1), 1, 2.0), ((2, 2, 1), 2, 2.25), ((2, 2, 2), 3, 2.5), ((3, 2, 2), 1, 3.0), ((3, 3, 2), 2, 3.25),
        ((3, 3, 3), 3, 3.5), ((4, 3, 3), 1, 4.0)]
    steps, _ = ctx.probe_raycast((6.749300603672869e-67, 6.750109954921438e-67, -85891558.96000093),
                                 (1.1036366354256313e-305, 0.0, 8589152896.000092), bounds=((-10, -20, -30), (10, 20, 30)), max_steps=1)
    assert tuple(steps[0]["cube"]) == (0, 0, -30) and steps[0]["face"] == 3 and steps[0]["t_distance"] == 0.010000000000000002
    steps, ended = ctx.probe_raycast((10.5, 20.5, 30.5), (1.0, 2.0, float("nan")), max_steps=3)
    assert ended and len(steps) == 1 and tuple(steps[0]["cube"]) == (10, 20, 30)
    steps, ended = ctx.probe_raycast((0.5, 0.5, float(2**31 - 2) - 0.5), (0.0, 0.0, 1.0), max_steps=5)
    assert ended and [tuple(s["cube"]) for s in steps] == [(0, 0, 2**31 - 3), (0, 0, 2**31 - 2)]


def test_probe_raycast_random_vs_oracle(ctx):
    rng = np.random.default_rng(99)
    for case in range(200):
        origin = rng.uniform(-30, 30, 3)
        direction = rng.normal(size=3) * (10.0 ** rng.uniform(-2, 3))
        if case % 5 == 0:
            direction[rng.integers(3)] = 0.0
        if case % 11 == 0:
            origin = np.round(origin)  # start exactly on cube edges
        lo = rng.integers(-10, 5, 3)
        hi = lo + rng.integers(1, 16, 3)
        bounds = None if case % 13 == 0 else (lo, hi)
        g, ge = ctx.probe_raycast(origin, direction, bounds=bounds, max_steps=48)
        r, re_ = oracle.raycast(origin, direction, bounds=bounds, max_steps=48)
        assert len(g) == len(r) and ge == re_, case
        assert (g["cube"] == r["cube"]).all() and (g["face"] == r["face"]).all(), case
        assert (g["t_distance"].view(np.uint64) == r["t_distance"].view(np.uint64)).all(), case
        assert (g["intersection_point"].view(np.uint64) == r["intersection_point"].view(np.uint64)).all(), case


def test_probe_raycast_extreme_exponents_vs_oracle(ctx):
    """Rays whose operands leave the exponent windows of the cheap divisions (aic_trace.hip div_known_recip: quotients by a direction through its t_delta, taken only for
    exponents far from the ends of the range) -- tiny and huge direction components, origins a denormal's width from a bounding plane or exactly on it, origins far away --
    so that the generic division behind the guard, and the guard itself, are compared with the oracle bit for bit like the ordinary rays above."""
    rng = np.random.default_rng(4242)
    for case in range(300):
        origin = rng.uniform(-30, 30, 3)
        direction = rng.normal(size=3)
        k = case % 6
        if k == 0:
            direction *= 10.0 ** rng.uniform(-300, -80)          # every component tiny
        elif k == 1:
            direction[rng.integers(3)] *= 10.0 ** rng.uniform(-300, -80)  # one tiny
        elif k == 2:
            direction *= 10.0 ** rng.uniform(60, 99)             # huge (but below the 1e100 cut-off of Parameters::new)
        elif k == 3:
            direction[rng.integers(3)] *= 10.0 ** rng.uniform(60, 99)
        elif k == 4:
            origin = origin * 10.0 ** rng.uniform(3, 9)          # far from the bounds: a long fast-forward
        lo = rng.integers(-10, 5, 3)
        hi = lo + rng.integers(1, 16, 3)
        if k == 5:
            a = rng.integers(3)
            lo[a] = 0
            origin[a] = [0.0, 5e-324, -5e-324, 1e-200, -1e-200, 1e-160][case // 6 % 6]  # on the plane, or next to it by less than the window allows
        if case % 7 == 0:
            direction[rng.integers(3)] = 0.0
        g, ge = ctx.probe_raycast(origin, direction, bounds=(lo, hi), max_steps=40)
        r, re_ = oracle.raycast(origin, direction, bounds=(lo, hi), max_steps=40)
        assert len(g) == len(r) and ge == re_, case
        assert (g["cube"] == r["cube"]).all() and (g["face"] == r["face"]).all(), case
        assert (g["t_distance"].view(np.uint64) == r["t_distance"].view(np.uint64)).all(), case
        assert (g["intersection_point"].view(np.uint64) == r["intersection_point"].view(np.uint64)).all(), case


def test_light_lut_matches_reference_table(ctx, golden_dir):
    ref = np.load(golden_dir / "packed_light_lut.npy")
    assert (ctx.probe_light_lut().view(np.uint32) == ref.view(np.uint32)).all()


# --- the reference's golden images through the HIP path -----------------------------------
def gpu_case(ctx, space, opt, size=COMMON_VIEWPORT, eye=(0.5, 0.5, 2.0), **kw):
    got, ref = render_both(ctx, space, opt, size, eye, **kw)
    assert_parity(got, ref)
    return got["rgba8"]


def test_golden_transparent_one(ctx, golden_dir):
    img = gpu_case(ctx, scenes.transparent_one_space(), oracle.unaltered_colors(transparency=0))
    assert diff_to(golden_dir, "transparent_one-surf-all", img).max() <= RGBA_TOL
    img = gpu_case(ctx, scenes.transparent_one_space(), oracle.unaltered_colors(transparency=1))
    assert diff_to(golden_dir, "transparent_one-vol-all", img).max() <= 2


def test_golden_emission_and_ramp(ctx, golden_dir):
    img = gpu_case(ctx, scenes.emission_space(), oracle.unaltered_colors())
    assert diff_to(golden_dir, "emission-all", img).max() <= RGBA_TOL
    img = gpu_case(ctx, scenes.color_srgb_ramp_space(), oracle.unaltered_colors(), size=(128, 128), eye=(16.0, 16.0, 17.0))
    assert diff_to(golden_dir, "color_srgb_ramp-all", img).max() <= RGBA_TOL
    img = gpu_case(ctx, scenes.one_cube_space(), oracle.unaltered_colors(), size=(101, 37))
    assert diff_to(golden_dir, "viewport_prime-all", img).max() <= 2


@pytest.mark.parametrize("kind", ["only", "semi"])
@pytest.mark.parametrize("mode,transparency", [("surf", 0), ("vol", 1)])
def test_golden_voxel_shape(ctx, golden_dir, kind, mode, transparency):
    img = gpu_case(ctx, scenes.voxel_shape_space(kind), oracle.unaltered_colors(transparency=transparency))
    ref = np.load(golden_dir / f"png_emission_{kind}-{mode}-all.npy")
    assert neighbourhood_diff(img, ref).max() <= RGBA_TOL


@pytest.mark.parametrize("name,with_world,with_ui", [("layers_all-all", True, True), ("layers_hidden_ui-all", True, False), ("layers_ui_only-all", False, True)])
def test_golden_layers(ctx, golden_dir, name, with_world, with_ui):
    opt = oracle.unaltered_colors(lighting=1) if with_world else oracle.unaltered_colors()
    got, ref = render_both(ctx, scenes.one_cube_space() if with_world else None, opt, COMMON_VIEWPORT, (0.5, 0.5, 2.0),
                           ui=scenes.ui_space() if with_ui else None)
    assert_parity(got, ref)
    from tests.test_oracle_goldens import with_info_text  # the "hello world" overlay (renderer.rs:659-683)
    assert diff_to(golden_dir, name, with_info_text(got["rgba8"])).max() <= RGBA_TOL


def test_viewport_zero(ctx):
    ctx.upload_space(abi.LAYER_WORLD, scenes.one_cube_space())
    got = ctx.render(ctx.make_frame(0, 0, world_inv=np.eye(4)))
    assert got["rgba8"].shape == (0, 0, 4) and got["info"].cubes_traced == 0


# --- seeded synthetic scenes: every option of the path vs the oracle ------------------------
SYNTH_EYE = (14.5, 22.5, 44.0)


def synth_quat():
    return oracle.look_at_y_up(SYNTH_EYE, (14.0, 8.0, 12.0))


@pytest.fixture(scope="module")
def synth_space():
    return scenes.synthetic_space(n=28, resolution=8, n_blocks=12, seed=2, light="field")


@pytest.mark.parametrize("transparency", [0, 1, 2])
@pytest.mark.parametrize("lighting", [0, 1, 2, 3, 4])
def test_synthetic_options_matrix(ctx, synth_space, transparency, lighting):
    opt = oracle.make_options(fog=1, transparency=transparency, threshold=0.6, lighting=lighting)
    got, ref = render_both(ctx, synth_space, opt, (160, 96), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)


@pytest.mark.parametrize("scale", [1e-200, 1e-90, 1e120])
def test_scaled_camera_matrix_takes_the_generic_division(ctx, synth_space, scale):
    """A camera matrix times a constant unprojects to the same rays up to rounding, with homogeneous coordinates far outside the exponent window in which the
    unprojection's three quotients share one reciprocal (aic_trace.hip unproject): the guard sends every lane to the generic division, which must give the oracle's
    bits for that matrix (first-hit t, per-pixel step counts, pixels), on all three kernel variants."""
    opt = oracle.make_options(fog=1, lighting=3)
    w, h = 96, 64
    _, _, inv = oracle.camera_matrices(90.0, opt.view_distance, w / h, synth_quat(), SYNTH_EYE)
    inv = inv * scale
    ctx.upload_space(abi.LAYER_WORLD, synth_space)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    opt.exposure = 1.0
    frame = ctx.make_frame(w, h, world_inv=inv)
    got = ctx.render(frame, want_aux=True)
    assert_production_variants(ctx, frame, got)
    ref = oracle.render(oracle.Space(synth_space), opt, oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(got, ref)
    assert got["info"].cubes_traced > w * h  # (the scaled camera still sees the scene)


@pytest.mark.parametrize("fog", [0, 1, 2, 3])
def test_synthetic_fog_modes(ctx, synth_space, fog):
    opt = oracle.make_options(fog=fog, view_distance=60.0)
    got, ref = render_both(ctx, synth_space, opt, (128, 80), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)


def test_synthetic_antialias_tonemap_debug(ctx, synth_space):
    opt = oracle.make_options(antialiasing=2)
    got, ref = render_both(ctx, synth_space, opt, (96, 64), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)
    opt = oracle.make_options(tone_mapping=1, maximum_intensity=1.0)
    opt.exposure = 1.0
    got, ref = render_both(ctx, synth_space, opt, (96, 64), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)
    opt = oracle.make_options(debug_pixel_cost=True)
    got, ref = render_both(ctx, synth_space, opt, (96, 64), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)


def test_synthetic_octant_sky_backdrop_and_ui(ctx, synth_space):
    sp = scenes.synthetic_space(n=20, resolution=4, n_blocks=6, seed=5)
    sp.set_sky_octants(np.random.default_rng(1).uniform(0.1, 1.5, (8, 3)))
    opt = oracle.make_options(lighting=3)
    got, ref = render_both(ctx, sp, opt, (120, 72), (10.5, 16.5, 32.0), oracle.look_at_y_up((10.5, 16.5, 32.0), (10, 6, 10)),
                           ui=scenes.ui_space(), backdrop=(0.2, 0.4, 0.6, 0.5))
    assert_parity(got, ref)


def test_bounce_with_octant_sky_ui_layer_and_antialiasing(ctx):
    """Bounce through the layer stack: the UI layer's primary rays are traced without the sky but THEIR secondary rays with it
    (trace_ray_impl(ray, .., true, false)); each antialiasing sample seeds its own RNG from its own direction; the octant sky is sampled
    by the secondary ray's direction."""
    sp = scenes.synthetic_space(n=20, resolution=4, n_blocks=6, seed=5)
    sp.set_sky_octants(np.random.default_rng(1).uniform(0.1, 1.5, (8, 3)))
    opt = oracle.make_options(lighting=5, bounce_samples=2, antialiasing=2)
    got, ref = render_both(ctx, sp, opt, (96, 60), (10.5, 16.5, 32.0), oracle.look_at_y_up((10.5, 16.5, 32.0), (10, 6, 10)),
                           ui=scenes.ui_space(), backdrop=(0.2, 0.4, 0.6, 0.5))
    assert_parity(got, ref)


def test_xcd_local_tile_queues_trace_every_pixel_once(ctx, synth_space):
    """The persistent kernel's waves take their tiles from one queue per XCD (macro tiles dealt to queues by super-block, a workgroup
    starting on its XCD's queue and moving on when it is empty: csrc/aic_trace.hip order_tiles_kernel). Whatever the number of queues
    and the block size, and with or without the previous frame's cost record, the frame is the one the single dispenser gives --
    pixels, per-pixel step counts and totals -- and the oracle's."""
    opt = oracle.make_options(fog=1, transparency=1, lighting=2)
    size = (416, 232)  # not a multiple of the tile, the macro tile or any super-block
    base, ref = render_both(ctx, synth_space, opt, size, SYNTH_EYE, synth_quat(), tuning=abi.tuning(queues=1))
    assert_parity(base, ref)
    assert base["info"].tile_queues == 0  # (one queue = the single counter)
    w, h = size
    _, _, inv = oracle.camera_matrices(90.0, opt.view_distance, w / h, synth_quat(), SYNTH_EYE)
    for queues, shift in [(8, None), (8, 0), (8, 2), (8, 7), (3, 1), (5, None), (2, 12)]:
        tune = abi.tuning(queues=queues, super_shift=shift)
        frame = ctx.make_frame(w, h, world_inv=inv, tuning=tune)
        for attempt in range(2):  # the second frame of the same view is ordered by the first one's cost record
            got = ctx.render(frame, want_aux=True)
            assert (got["rgba8"] == base["rgba8"]).all(), (queues, shift, attempt)
            assert (got["aux"]["cubes_traced"] == base["aux"]["cubes_traced"]).all()
            assert got["info"].cubes_traced == base["info"].cubes_traced
            assert got["info"].tile_queues == queues and got["info"].variant == abi.VARIANT_RECORDING
        for variant in (abi.VARIANT_PLAIN, abi.VARIANT_EXCHANGING):  # the production variants
            plain = ctx.render(ctx.make_frame(w, h, world_inv=inv, tuning=tune | abi.tuning(variant=variant)))
            assert (plain["rgba8"] == base["rgba8"]).all() and plain["info"].cubes_traced == base["info"].cubes_traced
            assert plain["info"].variant == variant and plain["info"].tile_queues == queues


def test_frames_of_changing_shape_on_one_slot(ctx, synth_space):
    """A slot prepares its next frame behind the one that is done (counters and cost record cleared, the record turned into a tile order:
    csrc/aic_abi.cpp submit_frame), keyed by the frame's shape and view. Frames of different sizes, views and feedback settings taken in turn on the
    same slot must each be what a fresh context gives, sums included -- nothing of a neighbour's record, order or counters may leak."""
    opt = oracle.make_options(fog=1, transparency=1, lighting=2)
    ctx.upload_space(abi.LAYER_WORLD, synth_space)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    views = []
    for (w, h), eye in [((416, 232), SYNTH_EYE), ((96, 64), SYNTH_EYE), ((640, 360), (SYNTH_EYE[0] + 3.0, SYNTH_EYE[1], SYNTH_EYE[2] - 2.0)), ((416, 232), (SYNTH_EYE[0] - 2.0, SYNTH_EYE[1] + 1.0, SYNTH_EYE[2]))]:
        _, _, inv = oracle.camera_matrices(90.0, opt.view_distance, w / h, synth_quat(), eye)
        views.append((w, h, inv))
    fresh = abi.Context(0)
    try:
        fresh.upload_space(abi.LAYER_WORLD, synth_space)
        fresh.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        want = []
        for w, h, inv in views:
            r = fresh.render(fresh.make_frame(w, h, world_inv=inv), counters=True)
            want.append((r["rgba8"].copy(), int(r["info"].cubes_traced), int(r["info"].n_hits)))
    finally:
        fresh.close()
    rng = np.random.default_rng(5)
    for k in rng.integers(0, len(views), 40):
        w, h, inv = views[int(k)]
        frame = ctx.make_frame(w, h, world_inv=inv)
        if rng.integers(0, 3) == 0:
            frame.flags |= abi.FRAME_NO_FEEDBACK
        counters = bool(rng.integers(0, 2))
        r = ctx.render(frame, counters=counters)
        assert (r["rgba8"] == want[int(k)][0]).all(), int(k)
        assert int(r["info"].cubes_traced) == want[int(k)][1]
        if counters:
            assert int(r["info"].n_hits) == want[int(k)][2]


def test_step_cap_and_camera_inside_geometry(ctx):
    # a corridor of recursive blocks made only of invisible voxels: every voxel is a counted
    # step, so rays along the corridor run into the 1000-step cap (sr.rs:643)
    from all_is_cubes_amd import flat

    sp = flat.FlatSpace((0, 0, 0), (64, 3, 3))
    sp.set_sky_uniform((0.2, 0.5, 0.9))
    pal = np.stack([flat.evoxel((0, 0, 0, 0)), flat.evoxel((1.0, 0.5, 0.25, 1.0))])
    vox = np.zeros((32, 32, 32), np.uint16)
    vox[5, 7, 9] = 1  # one visible voxel so the block is not trivially empty
    sp.block_index[...] = sp.add_block(flat.voxel_block(32, vox, pal))
    opt = oracle.unaltered_colors(view_distance=500.0)
    eye = (-3.0, 1.4, 1.6)
    got, ref = render_both(ctx, sp, opt, (64, 48), eye, oracle.look_at_y_up(eye, (64.0, 1.5, 1.5)))
    assert_parity(got, ref)
    assert int(ref["aux"]["cubes_traced"].max()) >= 1001  # the cap was reached
    # camera inside solid geometry (ray origin within an opaque cube => Face7::Within hits)
    sp2 = scenes.synthetic_space(n=24, resolution=8, n_blocks=4, seed=9)
    eye = (12.3, 3.2, 12.7)
    got, ref = render_both(ctx, sp2, oracle.make_options(), (64, 48), eye, oracle.look_at_y_up(eye, (0.0, 4.0, 0.0)))
    assert_parity(got, ref)


def test_row_partition_matches_full_frame(ctx, synth_space):
    opt = oracle.make_options()
    w, h = 100, 70
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, synth_quat(), SYNTH_EYE)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, synth_space)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    full = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    strip, n = 16, 3
    total = 0
    for part in range(n):
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv, partition=(strip, n, part)))
        rows = [y for y in range(h) if (y // strip) % n == part]
        assert got["rgba8"].shape[0] == len(rows)
        assert (got["rgba8"] == full[rows]).all()
        total += got["info"].cubes_traced
    assert total == ctx.render(ctx.make_frame(w, h, world_inv=inv))["info"].cubes_traced


# --- incremental update protocol (updating.rs:295-332: updated snapshot == fresh snapshot) ---
def test_incremental_updates_equal_full_upload(ctx):
    sp = scenes.synthetic_space(n=20, resolution=8, n_blocks=6, seed=4, light="field")
    opt = oracle.make_options()
    w, h = 96, 64
    eye = (10.5, 18.5, 30.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (10, 6, 10)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    rng = np.random.default_rng(3)
    # SpaceChange::CubeBlock + CubeLight
    xyz = rng.integers(0, 20, (300, 3)).astype(np.int32)
    _, first = np.unique(xyz, axis=0, return_index=True)
    xyz = xyz[np.sort(first)]
    new_idx = rng.integers(0, len(sp.blocks), len(xyz)).astype(np.uint16)
    new_light = np.stack([rng.integers(100, 180, len(xyz))] * 3 + [np.full(len(xyz), 255)], axis=1).astype(np.uint8)
    ctx.update_cubes(abi.LAYER_WORLD, xyz, new_idx, new_light)
    sp.block_index[xyz[:, 0], xyz[:, 1], xyz[:, 2]] = new_idx
    sp.light[xyz[:, 0], xyz[:, 1], xyz[:, 2]] = new_light
    # SpaceChange::BlockEvaluation: replace one palette entry, append another
    repl = scenes.synthetic_blocks(8, 1, seed=77)[0]
    victim = len(sp.blocks) - 1
    ctx.replace_block(abi.LAYER_WORLD, victim, repl)
    sp.blocks[victim] = repl
    extra = scenes.synthetic_blocks(4, 1, seed=78)[0]
    ctx.replace_block(abi.LAYER_WORLD, len(sp.blocks), extra)
    new_index = sp.add_block(extra)
    ctx.update_cubes(abi.LAYER_WORLD, np.array([[10, 12, 10]], np.int32), np.array([new_index], np.uint16))
    sp.set((10, 12, 10), new_index)
    # whole light volume (config 5's per-frame light re-upload)
    sp.light[..., 0:3] = np.minimum(sp.light[..., 0:3].astype(int) + 3, 255).astype(np.uint8) * (sp.light[..., 3:4] == 255)
    ctx.update_light_volume(abi.LAYER_WORLD, sp.light)
    incremental = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    fresh = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
    assert (incremental["rgba8"] == fresh["rgba8"]).all()
    assert incremental["info"].cubes_traced == fresh["info"].cubes_traced
    ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(incremental, ref)


def test_errors_are_reported_not_swallowed(ctx):
    with pytest.raises(abi.AicError):
        ctx.update_cubes(abi.LAYER_UI + 5, np.zeros((1, 3), np.int32), np.zeros(1, np.uint16))
    bad = scenes.one_cube_space()
    bad.block_index[...] = 7  # index beyond the palette
    with pytest.raises((abi.AicError, ValueError)):
        ctx.upload_space(abi.LAYER_WORLD, bad)
    o = abi.make_options()
    o.fog = 9
    with pytest.raises(abi.AicError):
        ctx.set_options(abi.LAYER_WORLD, o)


def test_positive_sign_inputs_are_checked_and_negative_zero_is_canonical(ctx):
    """Colours, sky and exposure are PositiveSign<f32> in the reference (restricted_number.rs: not NaN, sign bit clear; -0.0 is stored as +0.0).
    The kernel's PositiveSign::mul relies on it (max(product, 0)), so the boundary rejects what the type cannot hold and stores -0.0 as +0.0."""
    sp = scenes.synthetic_space(n=12, resolution=4, n_blocks=4, seed=5)
    w, h = 48, 32
    eye = (6.5, 9.5, 20.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (6.0, 4.0, 6.0)), eye)
    opt = abi.make_options()
    ctx.set_options(abi.LAYER_WORLD, opt)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    plain = ctx.render(ctx.make_frame(w, h, world_inv=inv))
    bad = scenes.synthetic_space(n=12, resolution=4, n_blocks=4, seed=5)
    bad.sky[0, 1] = -0.25
    with pytest.raises(abi.AicError):
        ctx.upload_space(abi.LAYER_WORLD, bad)
    bad.sky[0, 1] = np.nan
    with pytest.raises(abi.AicError):
        ctx.upload_space(abi.LAYER_WORLD, bad)
    for ex in (-1.0, float("nan")):
        with pytest.raises(abi.AicError):
            ctx.render(ctx.make_frame(w, h, world_inv=inv, exposure=ex))
    # -0.0 where the scene has +0.0: same frame, and equal to the oracle's
    nz = scenes.synthetic_space(n=12, resolution=4, n_blocks=4, seed=5)
    nz.sky = np.where(nz.sky == 0, np.float32(-0.0), nz.sky)
    for b in nz.blocks:
        pal = getattr(b, "palette", None)
        if pal is not None and len(pal):
            pal[...] = np.where(pal == 0, np.float32(-0.0), pal)
    ctx.upload_space(abi.LAYER_WORLD, nz)
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
    assert (got["rgba8"] == plain["rgba8"]).all()
    ref = oracle.render(oracle.Space(sp), oracle.make_options(), oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(got, ref)


# --- the C++ host mirror (HeadlessRenderer surface) on the GPU --------------------------------
def test_hip_rt_renderer_update_and_draw(ctx, synth_space):
    """HipRtRenderer mirrors RtRenderer: update() snapshots (full, then incremental through
    SpaceChange notifications, updating.rs:107-219), draw() never touches the scene."""
    import all_is_cubes_amd as A
    from all_is_cubes_amd import _host as H

    w, h = 120, 80
    cams = H.StandardCameras()
    o = H.GraphicsOptions()
    cams.graphics_options = o
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    space = A.space_from_flat(synth_space)
    cams.world_space = space
    cams.world_view_transform = H.look_at_y_up(SYNTH_EYE, (14.0, 8.0, 12.0))
    r = H.HipRtRenderer(cams)
    assert r.update() is True
    img = r.draw("")
    assert (img.width, img.height) == (w, h) and img.data.shape == (h, w, 4)
    assert img.flaws & H.Flaws.NO_BLOOM == H.Flaws.NO_BLOOM  # default bloom_intensity != 0 (renderer.rs:293-297)
    opt = oracle.make_options()
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(SYNTH_EYE, (14.0, 8.0, 12.0)), SYNTH_EYE)
    ref = oracle.render(oracle.Space(synth_space), opt, oracle.make_camera(inv, w, h))
    assert np.abs(img.data.astype(int) - ref["rgba8"].astype(int)).max() <= RGBA_TOL
    assert img.info.cubes_traced == int(ref["info"]["cubes_traced"])
    assert "cubes/pixel" in img.info.status_text()
    # nothing changed => update reports False and the image is identical
    assert r.update() is False
    assert (r.draw("").data == img.data).all()
    # incremental: edit cubes + light through the Space API, compare with the oracle on the edited scene
    a_idx = 0
    for (x, y, z) in [(14, 12, 12), (13, 11, 12), (14, 13, 13)]:
        space.set(x, y, z, a_idx)
        synth_space.set((x, y, z), a_idx)
        space.set_light(x, y + 1, z, H.PackedLight(170, 160, 150, 255))
        synth_space.light[x, y + 1, z] = (170, 160, 150, 255)
    assert r.update(H.Cursor()) is True
    img2 = r.draw("hello")
    ref2 = oracle.render(oracle.Space(synth_space), opt, oracle.make_camera(inv, w, h))
    from tests.test_oracle_goldens import with_info_text  # the default options draw the info text over the frame (renderer.rs:205-217)
    assert np.abs(img2.data.astype(int) - with_info_text(ref2["rgba8"], "hello").astype(int)).max() <= RGBA_TOL
    assert (img2.data[5 + 3:5 + 13, 5:5 + 35] == 255).all(axis=-1).any()  # ... in white
    assert img2.info.cubes_traced == int(ref2["info"]["cubes_traced"])
    assert img2.flaws & H.Flaws.NO_CURSOR == H.Flaws.NO_CURSOR
    # size_policy + options change (any options change => re-sent, updating.rs:68-73)
    o.lighting_display = H.LightingOption(H.LightingKind.Bounce, 4)
    cams.graphics_options = o
    r.update()
    img3 = r.draw("")
    # Bounce is traced as such (round 4), not substituted by Linear (the UNSUPPORTED bit left in `flaws` is NO_CURSOR's)
    o3 = oracle.make_options(lighting=5, bounce_samples=4)
    ref3 = oracle.render(oracle.Space(synth_space), o3, oracle.make_camera(inv, w, h))
    assert np.abs(img3.data.astype(int) - ref3["rgba8"].astype(int)).max() <= RGBA_TOL
    assert img3.info.cubes_traced == int(ref3["info"]["cubes_traced"])


@pytest.mark.parametrize("transparency,samples,fog", [(0, 1, 1), (0, 3, 2), (1, 1, 1), (1, 4, 3), (2, 2, 0)])
def test_bounce_lighting_matches_the_oracle(ctx, synth_space, transparency, samples, fog):
    """LightingOption::Bounce { samples } proper (surface.rs:119-166): a fully opaque surface is lit by `samples` secondary rays --
    whole trace_ray_impl calls under the same options, Flat-lit, sky included -- in directions drawn from the primary ray's
    SmallRng (sr.rs:165-178) through rand_distr::UnitSphere. Device == oracle per pixel: step counts INCLUDING the secondary
    rays' (RaytraceInfo + secondary_info), first hits, t bits, counters, RGBA8. Both restate rand 0.10.1 / rand_distr 0.6.0 from
    their published algorithms: PARITY UNPINNED against the reference (it holds no golden for Bounce, cases/src/lib.rs:45-50)."""
    opt = oracle.make_options(fog=fog, transparency=transparency, threshold=0.6, lighting=5, bounce_samples=samples)
    got, ref = render_both(ctx, synth_space, opt, (128, 80), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)
    assert not (got["info"].flaws & abi.FLAW_UNSUPPORTED)
    # the production variant of the Bounce kernel draws the same frame and counts the same steps
    fast = ctx.render(ctx.make_frame(128, 80, world_inv=oracle.camera_matrices(90.0, opt.view_distance, 128 / 80, synth_quat(), SYNTH_EYE)[2]))
    assert (fast["rgba8"] == got["rgba8"]).all() and fast["info"].cubes_traced == got["info"].cubes_traced
    # ... and it is not the Flat frame: the bounce rays did something
    flat_opt = oracle.make_options(fog=fog, transparency=transparency, threshold=0.6, lighting=1)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(flat_opt))
    flat_frame = ctx.render(ctx.make_frame(128, 80, world_inv=oracle.camera_matrices(90.0, opt.view_distance, 128 / 80, synth_quat(), SYNTH_EYE)[2]))
    assert fast["info"].cubes_traced > flat_frame["info"].cubes_traced
    assert (fast["rgba8"] != flat_frame["rgba8"]).any()


def test_bounce_with_debug_pixel_cost_and_the_step_cap(ctx, synth_space):
    """Secondary rays run trace_ray_impl's whole tail: `finish` applies debug_pixel_cost to THEIR accumulators too (sr.rs:658-693),
    and the cost shown for the pixel counts primary steps only (primary_cubes_traced), while cubes_traced adds the secondary ones."""
    opt = oracle.make_options(lighting=5, bounce_samples=2, debug_pixel_cost=True)
    got, ref = render_both(ctx, synth_space, opt, (96, 64), SYNTH_EYE, synth_quat())
    assert_parity(got, ref)


def test_multi_part_gather_on_one_gpu(ctx, synth_space):
    """The N>1 data path on a single device: render each partition to device memory, stack as the
    gather would, de-interleave with the device kernel and with the torch reference."""
    import torch

    import all_is_cubes_amd as A
    from all_is_cubes_amd import _host as H
    from all_is_cubes_amd import distributed as D

    w, h, strip, n = 200, 90, 16, 4
    cams = H.StandardCameras()
    cams.graphics_options = H.GraphicsOptions()
    cams.viewport = H.Viewport.with_scale(1.0, w, h)
    cams.world_space = A.space_from_flat(synth_space)
    cams.world_view_transform = H.look_at_y_up(SYNTH_EYE, (14.0, 8.0, 12.0))
    r = H.HipRtRenderer(cams)
    r.update()
    full = torch.from_numpy(r.draw("").data.copy())
    max_rows = D.max_partition_rows(h, strip, n)
    gathered = torch.zeros((n, max_rows, w, 4), dtype=torch.uint8, device="cuda")
    for p in range(n):
        rows = r.partition_rows(strip, n, p)
        assert rows == len(D.partition_rows(h, strip, n, p))
        buf = torch.empty((rows, w, 4), dtype=torch.uint8, device="cuda")
        r.draw_rows_to_device(buf.data_ptr(), strip, n, p)
        gathered[p, :rows] = buf
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    r.assemble_strips(gathered.data_ptr(), out.data_ptr(), strip, n)
    torch.cuda.synchronize()
    assert (out.cpu() == full).all()
    assert (D.assemble_strips_torch(gathered, h, w, strip).cpu() == full).all()


# --- sparse blocks (long runs of invisible voxels), partial stored volumes, 16-bit palettes ----
def _sparse_block(res, seed, vlo=(0, 0, 0), vsize=None, n_pal=5):
    rng = np.random.default_rng(seed)
    vsize = vsize or (res, res, res)
    pal = np.zeros((n_pal, 8), np.float32)
    pal[0] = flat.evoxel((0, 0, 0, 0))                  # invisible
    pal[1] = flat.evoxel((0, 0, 0, 0))                  # a second, distinct-index invisible entry
    for i in range(2, n_pal):
        a = 1.0 if i % 2 else 0.4
        pal[i] = flat.evoxel((*rng.uniform(0.1, 0.9, 3), a), emission=(0.0, 0.05 * (i % 3), 0.0))
    vox = rng.integers(0, 2, vsize).astype(np.uint16)   # mostly the two invisible entries...
    pts = rng.integers(0, np.array(vsize), (max(3, res * res // 6), 3))
    vox[pts[:, 0], pts[:, 1], pts[:, 2]] = rng.integers(2, n_pal, len(pts))  # ...with a few visible voxels
    return flat.voxel_block(res, vox, pal, vlo=vlo)


def _sparse_space(extra_blocks=()):
    sp = flat.FlatSpace((-3, 0, -3), (6, 5, 6))
    sp.set_sky_uniform((0.6, 0.7, 0.9))
    blocks = [_sparse_block(32, 1), _sparse_block(16, 2, vlo=(3, 0, 5), vsize=(9, 16, 7)), _sparse_block(8, 3), *extra_blocks]
    ids = [sp.add_block(b) for b in blocks]
    rng = np.random.default_rng(9)
    for x in range(-3, 3):
        for y in range(0, 5):
            for z in range(-3, 3):
                if rng.random() < 0.6:
                    sp.set((x, y, z), ids[int(rng.integers(0, len(ids)))])
    return sp


@pytest.mark.parametrize("transparency", [0, 1])
def test_sparse_blocks_match_oracle(ctx, transparency):
    sp = _sparse_space()
    opt = oracle.make_options(transparency=transparency)
    w, h = 128, 96
    eye = (0.3, 2.6, 7.5)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0, 2.2, 0)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
    ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(got, ref)
    assert int(ref["info"]["n_inner"]) > 20 * w * h  # rays really cross long empty runs


def test_block_with_16bit_palette(ctx):
    sp = _sparse_space()
    opt = oracle.make_options()
    w, h = 96, 64
    eye = (0.3, 2.6, 7.5)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0, 2.2, 0)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    # a block whose palette needs more than 13 bits arrives as an incremental update
    rng = np.random.default_rng(21)
    n_pal = 9000
    pal = np.zeros((n_pal, 8), np.float32)
    pal[1:, 0:3] = rng.uniform(0.05, 0.95, (n_pal - 1, 3))
    pal[1:, 3] = 1.0
    vox = np.zeros((32, 32, 32), np.uint16)
    sel = rng.random((32, 32, 32)) < 0.3
    vox[sel] = rng.integers(1, n_pal, int(sel.sum()))
    vox[0, 0, 0] = n_pal - 1
    big = flat.voxel_block(32, vox, pal)
    ctx.replace_block(abi.LAYER_WORLD, 1, big)
    sp.blocks[1] = big
    inc = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
    ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(inc, ref)
    ctx.upload_space(abi.LAYER_WORLD, sp)  # and as a full snapshot
    assert_parity(ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True), ref)


def test_replace_block_changes_class_of_placed_cubes(ctx):
    """SpaceChange::BlockEvaluation that turns an atom into a recursive block (and into an invisible
    one): every cube already holding that index must be re-classified (updating.rs:139-153)."""
    sp = flat.FlatSpace((0, 0, 0), (5, 4, 5))
    sp.set_sky_uniform((0.5, 0.6, 0.8))
    a = sp.add_block(flat.atom((0.8, 0.2, 0.2, 1.0)))
    b = sp.add_block(flat.atom((0.2, 0.8, 0.2, 1.0)))
    rng = np.random.default_rng(5)
    for x in range(5):
        for z in range(5):
            sp.set((x, 0, z), a if (x + z) % 2 else b)
            if rng.random() < 0.4:
                sp.set((x, 1 + int(rng.integers(0, 3)), z), a)
    opt = oracle.make_options()
    w, h = 96, 72
    eye = (2.5, 3.2, 7.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (2.5, 1.0, 2.5)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    for replacement in (_sparse_block(8, 11), flat.atom((0.0, 0.0, 0.0, 0.0)), flat.atom((0.1, 0.1, 0.9, 0.5)), scenes.synthetic_blocks(4, 1, seed=3)[0]):
        ctx.replace_block(abi.LAYER_WORLD, a, replacement)
        sp.blocks[a] = replacement
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
        ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
        assert_parity(got, ref)


# --- the reference's own ASCII golden frames (raytracer/text.rs:195-341), from the device ---------
def _gpu_print_space(ctx, sp, direction=(1.0, 1.0, 1.0)):
    """PrintSpace (text.rs:147-182) with the device as the tracer: CharacterBuf shows the first hit
    block's character, ' ' for a ray that entered the space and hit nothing, '.' for one that never
    entered it -- all three are in the per-pixel aux record (first hit, step count)."""
    eye = oracle.eye_for_look_at(sp.lo, sp.hi, direction)
    center = (np.array(sp.lo, float) + np.array(sp.hi, float)) / 2.0
    _, _, inv = oracle.camera_matrices(90.0, 200.0, 1.0, oracle.look_at_y_up(eye, center), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options())
    aux = ctx.render(ctx.make_frame(80, 40, world_inv=inv), want_aux=True)["aux"]
    lines = []
    for row in aux:
        lines.append("".join(sp.blocks[int(p["block_index"])].name[0] if p["hit"] == 1 else (" " if p["cubes_traced"] > 0 else ".") for p in row))
    return "\n".join(lines) + "\n"


def test_reference_ascii_frames_from_device(ctx):
    golden = Path(__file__).parent / "golden"
    assert _gpu_print_space(ctx, scenes.print_space_test_space()) == (golden / "ascii_print_space.txt").read_text()
    assert _gpu_print_space(ctx, scenes.partial_voxels_space()) == (golden / "ascii_partial_voxels.txt").read_text()


# --- BASELINE config 5: orbiting camera, light volume re-uploaded every frame; in miniature (4 frames) and at its full
#     1920x1080 (3 frames, every pixel against the oracle) ---------------------------------------------------------------
@pytest.mark.parametrize("w,h,frames", [(96, 54, 4), (1920, 1080, 3)])
def test_orbit_with_light_reupload_matches_oracle(ctx, w, h, frames):
    sp = scenes.atrium_like_space()
    opt = oracle.make_options()
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    base = sp.light.copy()
    for k in range(frames):
        a = 2.0 * np.pi * k / 4.0
        sp.light[..., 0:3] = np.clip(base[..., 0:3].astype(np.int32) + int(round(8 * np.sin(a))), 0, 255).astype(np.uint8) * (base[..., 0:3] > 0)
        ctx.update_light_volume(abi.LAYER_WORLD, sp.light)
        eye = (0.5 + 7.0 * np.sin(a), 9.91, 7.0 * np.cos(a))
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0.5, 8.0, 0.0)), eye)
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
        ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True, threads=min(32, os.cpu_count() or 4))
        assert_parity(got, ref)


# --- streaming pair: two frames in flight equal two synchronous frames ---------------------------
def test_two_frames_in_flight_equal_synchronous_frames(ctx):
    import torch

    sp = scenes.synthetic_space(n=20, resolution=8, n_blocks=6, seed=4, light="field")
    w, h = 160, 96
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options())
    frames = []
    for k in range(5):
        eye = (10.5 + 6.0 * np.sin(k), 18.5, 30.0 - 2.0 * k)
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (10, 6, 10)), eye)
        frames.append(ctx.make_frame(w, h, world_inv=inv))
    sync = [ctx.render(f)["rgba8"].copy() for f in frames]
    sync_steps = [ctx.render(f)["info"].cubes_traced for f in frames]
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    got, steps, in_flight = [], [], []

    def complete():
        k, slot = in_flight.pop(0)
        info = ctx.render_wait(slot)
        got.append((k, bufs[slot].cpu().numpy().copy()))
        steps.append(info.cubes_traced)

    for k, f in enumerate(frames):
        if len(in_flight) == 2:
            complete()
        ctx.render_submit(f, bufs[k % 2].data_ptr(), k % 2)
        in_flight.append((k, k % 2))
    with pytest.raises(abi.AicError):  # that slot is taken
        ctx.render_submit(frames[0], bufs[0].data_ptr(), in_flight[0][1])
    while in_flight:
        complete()
    assert [k for k, _ in got] == list(range(5))
    for (k, img), ref in zip(got, sync):
        assert (img == ref).all(), f"frame {k}"
    assert steps == sync_steps
    # a scene update while a frame is in flight waits for it instead of racing it
    ctx.render_submit(frames[0], bufs[0].data_ptr(), 0)
    ctx.update_cubes(abi.LAYER_WORLD, np.array([[10, 12, 10]], np.int32), np.array([1], np.uint16))
    ctx.render_wait(0)
    assert (bufs[0].cpu().numpy() == sync[0]).all()


@pytest.mark.parametrize("in_flight_n", [3, 4, 8])
def test_streamed_full_size_frames_on_a_part_of_the_chip_equal_synchronous_frames(ctx, in_flight_n):
    """A streamed frame with several tiles per resident wave that is submitted while others are in flight is launched on a part of the resident
    grid (aic_abi.cpp submit_frames: a third with three others queued, a quarter from four on). Same frames, byte for byte, same step totals --
    whatever the grid: ten 1920x1080 frames of the C2 scene, a different camera each."""
    import torch

    sp = scenes.atrium_like_space()
    w, h = 1920, 1080
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options())
    frames = []
    for k in range(10):
        a = 2.0 * np.pi * k / 60.0
        eye = (0.5 + 7.0 * np.sin(a), 9.91, 7.0 * np.cos(a))
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0.5, 8.0, 0.0)), eye)
        frames.append(ctx.make_frame(w, h, world_inv=inv))
    sync, sync_steps = [], []
    for f in frames:
        r = ctx.render(f)
        sync.append(r["rgba8"].copy()); sync_steps.append(r["info"].cubes_traced)
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(in_flight_n)]
    queue, got = [], {}

    def complete():
        k, slot = queue.pop(0)
        info = ctx.render_wait(slot)
        got[k] = (bufs[slot].cpu().numpy().copy(), info.cubes_traced)

    for k, f in enumerate(frames):
        if len(queue) == in_flight_n:
            complete()
        slot = k % in_flight_n
        ctx.render_submit(f, bufs[slot].data_ptr(), slot)
        queue.append((k, slot))
    while queue:
        complete()
    for k in range(len(frames)):
        assert (got[k][0] == sync[k]).all(), f"frame {k}"
        assert got[k][1] == sync_steps[k], f"frame {k}: step total"


def test_light_reupload_does_not_disturb_a_frame_in_flight(ctx):
    """aic_update_light_volume is double-buffered: a frame submitted before it keeps the light it was
    submitted with, the next frame sees the new volume."""
    import torch

    sp = scenes.atrium_like_space()
    w, h = 192, 108
    eye = (0.5, 9.91, 10.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0.5, 8.0, -20.0)), eye)
    fr = ctx.make_frame(w, h, world_inv=inv)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options())
    dark = sp.light.copy()
    dark[..., 0:3] = np.clip(dark[..., 0:3].astype(np.int32) - 30, 0, 255).astype(np.uint8) * (sp.light[..., 0:3] > 0)
    before = ctx.render(fr)["rgba8"].copy()
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for rep in range(3):  # alternate volumes with frames in flight across the swap
        ctx.render_submit(fr, bufs[0].data_ptr(), 0)           # reads the bright volume
        ctx.update_light_volume(abi.LAYER_WORLD, dark)
        ctx.render_submit(fr, bufs[1].data_ptr(), 1)           # reads the dark one
        ctx.update_light_volume(abi.LAYER_WORLD, sp.light)
        ctx.render_submit(fr, bufs[2].data_ptr(), 2)           # bright again (buffer of frame 0: must wait for it)
        for s in range(3):
            ctx.render_wait(s)
        assert (bufs[0].cpu().numpy() == before).all()
        assert (bufs[2].cpu().numpy() == before).all()
        assert (bufs[1].cpu().numpy() != before).any()
    ctx.update_light_volume(abi.LAYER_WORLD, dark)
    ref = ctx.render(fr)["rgba8"]
    assert (bufs[1].cpu().numpy() == ref).all()


# --- BASELINE's full sizes: sampled rows against the oracle + size-independent properties -------------
@pytest.mark.parametrize("workload", ["atrium", "s256"])
def test_full_size_workload_rows_and_properties(ctx, workload):
    import bench

    sp, (w, h), eye, target, vd, _ = bench.build_workload(workload)
    opt = oracle.make_options(view_distance=vd)
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    fr = ctx.make_frame(w, h, world_inv=inv)
    first = ctx.render(fr, want_aux=True)   # index tile order (no cost record yet)
    again = ctx.render(fr, want_aux=True)   # costliest-first tile order from the first frame's record
    # idempotence / schedule independence: the picture and every per-pixel record do not depend on tile order
    assert (first["rgba8"] == again["rgba8"]).all()
    assert first["info"].cubes_traced == again["info"].cubes_traced
    fast = ctx.render(fr)  # the production kernel variant (no per-pixel records, 4 waves per SIMD)
    assert (fast["rgba8"] == again["rgba8"]).all() and fast["info"].cubes_traced == again["info"].cubes_traced
    # The PRODUCTION variant's per-pixel step counts: with debug_pixel_cost the pixel is
    # rgb(0.02 n, 0.002 n, ..) (accum.rs:228-234), which the linear float output hands over unrounded, so n comes back exactly
    # from the green channel -- no aux-recording kernel involved. Compared with the oracle's counts row by row below.
    opt_cost = oracle.make_options(view_distance=vd, debug_pixel_cost=True)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt_cost))
    cost = ctx.render(ctx.make_frame(w, h, world_inv=inv, flags=abi.FRAME_OUT_LINEAR))
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    prod_counts = np.rint(cost["rgba8"][..., 1].astype(np.float64) / float(np.float32(0.002))).astype(np.int64)
    assert cost["info"].cubes_traced == again["info"].cubes_traced and int(prod_counts.sum()) == again["info"].cubes_traced
    for k in ("cubes_traced", "hit", "cube", "voxel", "face", "block_index"):
        assert (first["aux"][k] == again["aux"][k]).all()
    # against the oracle, whole rows: steps, first hits, f64 t, RGBA8 -- EVERY row of the 1080p frame (config 2; the frame's
    # step total too), 64 rows spread over the 4K frame (config 3: a whole oracle frame there costs tens of seconds)
    osp, cam = oracle.Space(sp), oracle.make_camera(inv, w, h)
    spans = [(0, h)] if workload == "atrium" else [(y, y + 1) for y in sorted({int(round(k * (h - 1) / 63.0)) for k in range(64)})]
    steps_checked = 0
    for y0, y1 in spans:
        ref = oracle.render(osp, opt, cam, rows=(y0, y1), want_aux=True, threads=min(32, os.cpu_count() or 4))
        ga, ra = again["aux"][y0:y1], ref["aux"][y0:y1]
        assert (ga["cubes_traced"] == ra["cubes_traced"]).all(), f"rows {y0}..{y1}: step counts"
        assert (prod_counts[y0:y1] == ra["cubes_traced"]).all(), f"rows {y0}..{y1}: step counts of the production variant"
        for k in ("hit", "cube", "voxel", "resolution", "face", "block_index"):
            assert (ga[k] == ra[k]).all(), f"rows {y0}..{y1}: {k}"
        hit = ra["hit"] == 1
        assert (ga["t_distance"][hit].view(np.uint64) == ra["t_distance"][hit].view(np.uint64)).all(), f"rows {y0}..{y1}: t bits"
        assert np.abs(again["rgba8"][y0:y1].astype(np.int16) - ref["rgba8"][y0:y1].astype(np.int16)).max() <= RGBA_TOL
        steps_checked += int(ra["cubes_traced"].astype(np.int64).sum())
    if workload == "atrium":
        assert steps_checked == again["info"].cubes_traced
    # partition: eight ranks' strips add up to the frame, pixels and step count alike
    total = 0
    for part in range(8):
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv, partition=(16, 8, part)))
        rows = [y for y in range(h) if (y // 16) % 8 == part]
        assert (got["rgba8"] == again["rgba8"][rows]).all()
        total += got["info"].cubes_traced
    assert total == again["info"].cubes_traced


# Other production variants at BASELINE's full sizes: the test above runs <Volumetric, interpolated light>; these run
# <Surface, Flat> on C2 and <Threshold, None> on C3 -- the production kernels (no per-pixel records; lanes exchanged between waves since
# round 5), checked per pixel through the same trick: debug_pixel_cost + the linear float output give every pixel's step count exactly.
# (<VOL, LMODE> of the kernel: transparency 1 = Volumetric; lighting 0 None, 1 Flat, 2-4 interpolated. With the default <Volumetric, Linear> above these are all six
#  combinations the two scenes can reach -- the other six production kernels are the BIG ones, for block tables past 16 384 entries)
@pytest.mark.parametrize("workload,transparency,lighting", [("atrium", 0, 1), ("s256", 2, 0), ("atrium", 1, 0), ("s256", 1, 1), ("atrium", 2, 4)])
def test_full_size_other_production_variants(ctx, workload, transparency, lighting):
    import bench

    sp, (w, h), eye, target, vd, _ = bench.build_workload(workload)
    opt = oracle.make_options(view_distance=vd, transparency=transparency, lighting=lighting)
    opt_cost = oracle.make_options(view_distance=vd, transparency=transparency, lighting=lighting, debug_pixel_cost=True)
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    fr = ctx.make_frame(w, h, world_inv=inv)
    cold = ctx.render(fr)   # production variant, index tile order
    fast = ctx.render(fr)   # ... costliest-first order from the first frame's record
    assert (cold["rgba8"] == fast["rgba8"]).all() and cold["info"].cubes_traced == fast["info"].cubes_traced
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt_cost))
    cost = ctx.render(ctx.make_frame(w, h, world_inv=inv, flags=abi.FRAME_OUT_LINEAR))
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    prod_counts = np.rint(cost["rgba8"][..., 1].astype(np.float64) / float(np.float32(0.002))).astype(np.int64)
    assert cost["info"].cubes_traced == fast["info"].cubes_traced and int(prod_counts.sum()) == fast["info"].cubes_traced
    osp, cam = oracle.Space(sp), oracle.make_camera(inv, w, h)
    for y in sorted({int(round(k * (h - 1) / 31.0)) for k in range(32)}):
        ref = oracle.render(osp, opt, cam, rows=(y, y + 1), want_aux=True, threads=min(32, os.cpu_count() or 4))
        assert (prod_counts[y] == ref["aux"][y]["cubes_traced"]).all(), f"row {y}: step counts of the production variant"
        assert np.abs(fast["rgba8"][y].astype(np.int16) - ref["rgba8"][y].astype(np.int16)).max() <= RGBA_TOL, f"row {y}: RGBA8"


def test_full_size_bounce_frame_rows(ctx):
    """One C2-size frame with LightingOption::Bounce { samples: 2 } (device == oracle on 16 sampled rows; parity with the reference itself stays
    unpinned: it holds no golden for Bounce, cases/src/lib.rs:45-50). The Bounce variants keep their rays to the lane (no exchange)."""
    import bench

    sp, (w, h), eye, target, vd, _ = bench.build_workload("atrium")
    opt = oracle.make_options(view_distance=vd, lighting=5, bounce_samples=2)
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv))
    osp, cam = oracle.Space(sp), oracle.make_camera(inv, w, h)
    total = 0
    for y in sorted({int(round(k * (h - 1) / 15.0)) for k in range(16)}):
        ref = oracle.render(osp, opt, cam, rows=(y, y + 1), want_aux=True, threads=min(32, os.cpu_count() or 4))
        assert np.abs(got["rgba8"][y].astype(np.int16) - ref["rgba8"][y].astype(np.int16)).max() <= RGBA_TOL, f"row {y}: RGBA8"
        total += int(ref["aux"][y]["cubes_traced"].astype(np.int64).sum())
    assert total > 0


def test_device_side_handoff_orders_a_foreign_stream_behind_the_trace():
    """aic_stream_wait_frame: the stream an exchange step is issued from waits ON THE DEVICE for a submitted
    frame; the host only enqueues. Two contexts on the one GPU play two ranks: each streams its strips of six frames through four
    render slots, and after every submit a foreign (torch) stream is made to wait for the slot and copies the strips away --
    the stand-in for the gather, issued at once, with no host wait anywhere before the end. Without the device-side wait the
    copies run ahead of the traces (the buffers were zeroed) and the assembled frames come out wrong."""
    import torch

    import bench

    sp, (w, h), eye, target, vd, _ = bench.build_workload("atrium")
    opt = oracle.make_options(view_distance=vd)
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    dev = torch.device("cuda", 0)
    n_ranks, n_frames, depth = 2, 6, 4
    with abi.Context(0) as whole_ctx:
        whole_ctx.upload_space(abi.LAYER_WORLD, sp)
        whole_ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        want = whole_ctx.render(whole_ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    ranks = [abi.Context(0) for _ in range(n_ranks)]
    try:
        side = [torch.cuda.Stream(device=dev) for _ in range(n_ranks)]
        rows = [[y for y in range(h) if (y // 16) % n_ranks == r] for r in range(n_ranks)]
        src = [[torch.zeros((len(rows[r]), w, 4), dtype=torch.uint8, device=dev) for _ in range(depth)] for r in range(n_ranks)]
        dst = [[torch.zeros((len(rows[r]), w, 4), dtype=torch.uint8, device=dev) for _ in range(n_frames)] for r in range(n_ranks)]
        torch.cuda.synchronize()
        for r, c in enumerate(ranks):
            c.upload_space(abi.LAYER_WORLD, sp)
            c.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        for k in range(n_frames):
            for r, c in enumerate(ranks):
                slot = k % depth
                if k >= depth:
                    c.render_wait(slot)  # the slot's previous frame: long finished, its copy was issued four frames ago
                    side[r].synchronize()
                    src[r][slot].zero_()
                    torch.cuda.current_stream(dev).synchronize()
                c.render_submit(c.make_frame(w, h, world_inv=inv, partition=(16, n_ranks, r)), src[r][slot].data_ptr(), slot)
                c.stream_wait_frame(slot, side[r].cuda_stream)          # device-side: `side` waits for the trace
                with torch.cuda.stream(side[r]):
                    dst[r][k].copy_(src[r][slot], non_blocking=True)    # "the gather", issued at once
        for s_ in side:
            s_.synchronize()
        for r, c in enumerate(ranks):
            for slot in range(depth):
                c.render_wait(slot)
        for k in range(n_frames):
            frame = np.zeros((h, w, 4), np.uint8)
            for r in range(n_ranks):
                frame[rows[r]] = dst[r][k].cpu().numpy()
            assert (frame == want).all(), f"frame {k}: the copies were not ordered behind the traces"
    finally:
        for c in ranks:
            c.close()


def test_opaque_shortcut_counts_on_a_blocks_last_voxel(ctx):
    """the production variants end a ray on an opaque surface without taking the reference's remaining counted steps -- they work them out from the
    steps the level has left and from whether the suspended cube grid can still step. The cases where that reasoning is thinnest, per pixel (debug_pixel_cost
    through the float output): the opaque voxel is the LAST one of its block along the ray (the voxel level ends with it) or of a partly stored volume, in a
    block at the space's far edge (the cube grid ends with it too) and in one with cubes behind it; seen head-on, diagonally, and from inside the block; for
    Surface, Volumetric and Threshold transparency; plain and lane-exchanging variants against the aux-recording variant and the oracle."""
    res = 4
    pal = np.zeros((3, 8), np.float32)
    pal[1] = (0.8, 0.3, 0.2, 1.0, 0, 0, 0, 0)   # opaque
    pal[2] = (0.2, 0.4, 0.9, 0.5, 0, 0, 0, 0)   # translucent
    blocks = []
    for axis in range(3):
        v = np.zeros((res, res, res), np.uint16)
        idx = [slice(None)] * 3
        idx[axis] = res - 1
        v[tuple(idx)] = 1                       # an opaque wall on the block's far face along `axis`
        idx[axis] = 1
        v[tuple(idx)] = 2                       # a translucent sheet in front of it
        blocks.append(flat.voxel_block(res, v, pal))
        blocks.append(flat.voxel_block(res, np.ascontiguousarray(v[:, :, :res - 1] if axis != 2 else v[:res - 1]), pal))  # part of the volume stored
    low = np.zeros((res, res, res), np.uint16)
    low[0, 0, 0] = 1                            # ... and on the near corner: the last voxel going the other way
    blocks.append(flat.voxel_block(res, low, pal))
    sp = flat.FlatSpace((-1, 0, -2), (3, 2, 3))
    sp.set_sky_uniform((0.4, 0.5, 0.9))
    sp.add_block(flat.air())
    ids = [sp.add_block(b) for b in blocks]
    rng = np.random.default_rng(11)
    for c in np.ndindex(3, 2, 3):               # every cube holds one of the blocks or air: far-edge cubes, interior cubes, neighbours behind and none
        sp.block_index[c] = 0 if rng.random() < 0.25 else ids[int(rng.integers(0, len(ids)))]
    sp.light[...] = (200, 180, 160, 255)
    w, h = 72, 56
    views = [((-4.0, 1.0, -0.5), (0.5, 1.0, -0.5)), ((0.5, 1.0, 5.0), (0.5, 1.0, -0.5)), ((0.5, 6.0, -0.5), (0.5, 0.0, -0.4)), ((4.5, 3.5, 3.0), (0.5, 1.0, -0.5)),
             ((0.6, 0.7, -0.4), (2.0, 0.9, -1.9)), ((3.5, 1.0, -0.5), (0.5, 1.0, -0.5))]
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    for transparency in (0, 1, 2):
        opt = oracle.make_options(transparency=transparency, threshold=0.4, lighting=1, debug_pixel_cost=True)
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        for eye, target in views:
            _, _, inv = oracle.camera_matrices(70.0, opt.view_distance, w / h, oracle.look_at_y_up(eye, target), eye)
            fr = ctx.make_frame(w, h, world_inv=inv)
            got = ctx.render(fr, want_aux=True)
            ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
            assert_parity(got, ref)
            assert_production_variants(ctx, fr, got)
            for variant in (abi.VARIANT_PLAIN, abi.VARIANT_EXCHANGING):  # the per-pixel counts of the plain and of the lane-exchanging production variant
                cost = ctx.render(ctx.make_frame(w, h, world_inv=inv, flags=abi.FRAME_OUT_LINEAR, tuning=abi.tuning(variant=variant)))
                assert cost["info"].variant == variant
                counts = np.rint(cost["rgba8"][..., 1].astype(np.float64) / float(np.float32(0.002))).astype(np.int64)
                assert (counts == ref["aux"]["cubes_traced"]).all(), (transparency, eye, variant)


def test_block_table_past_14_bits_uses_the_class_table(ctx):
    """More than 16384 blocks: cube-grid entries can no longer carry the class bits (aic_device.h), the
    kernel classifies through the LDS table; growing past the limit by replace_block re-tags the grid."""
    rng = np.random.default_rng(31)
    sp = flat.FlatSpace((0, 0, 0), (12, 10, 12))
    sp.set_sky_uniform((0.4, 0.5, 0.7))
    sp.add_block(flat.air())
    rec = [sp.add_block(b) for b in scenes.synthetic_blocks(8, 3, seed=5)]
    n_atoms = 16384 - len(sp.blocks)  # exactly at the limit: still tagged
    for i in range(n_atoms):
        c = rng.uniform(0.05, 0.95, 3)
        sp.add_block(flat.atom((float(c[0]), float(c[1]), float(c[2]), 1.0 if i % 5 else 0.5)))
    assert len(sp.blocks) == 16384
    grid = rng.integers(1, len(sp.blocks), sp.size).astype(np.uint16)
    grid[rng.random(sp.size) < 0.75] = 0
    grid[:, 0, :] = rng.integers(1, len(sp.blocks), (12, 12))
    grid[3, 3, 3], grid[8, 2, 5] = rec[0], rec[2]
    sp.block_index[...] = grid
    opt = oracle.make_options()
    w, h = 112, 80
    eye = (6.0, 8.5, 17.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (6, 2, 6)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    fr = ctx.make_frame(w, h, world_inv=inv)
    assert_parity(ctx.render(fr, want_aux=True), oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True))
    # one more block: index 16384 needs 15 bits
    extra = scenes.synthetic_blocks(4, 1, seed=9)[0]
    ctx.replace_block(abi.LAYER_WORLD, len(sp.blocks), extra)
    idx = sp.add_block(extra)
    ctx.update_cubes(abi.LAYER_WORLD, np.array([[5, 1, 9], [6, 4, 6]], np.int32), np.array([idx, idx], np.uint16))
    sp.set((5, 1, 9), idx)
    sp.set((6, 4, 6), idx)
    ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), want_aux=True)
    assert_parity(ctx.render(fr, want_aux=True), ref)
    ctx.upload_space(abi.LAYER_WORLD, sp)  # and as a fresh snapshot of 16385 blocks
    assert_parity(ctx.render(fr, want_aux=True), ref)
    # the BIG production kernels (no per-pixel records; plain and lane-exchanging), all six <VOL, LMODE> of them, against the aux-recording variant
    for transparency, lighting in ((1, 3), (1, 0), (1, 1), (0, 3), (2, 0), (0, 1)):
        vopt = oracle.make_options(transparency=transparency, lighting=lighting)
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(vopt))
        assert_production_variants(ctx, fr, ctx.render(fr, want_aux=True))
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    # Bounce on the untagged grid: the secondary rays classify cubes through the class table too
    bopt = oracle.make_options(lighting=5, bounce_samples=2)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(bopt))
    assert_parity(ctx.render(fr, want_aux=True), oracle.render(oracle.Space(sp), bopt, oracle.make_camera(inv, w, h), want_aux=True))


# --- to_text::<CharacterBuf> through the host mirror: pixel-centre rays (sr.rs:367-472) ---------------
def test_draw_text_matches_reference_frames_and_oracle(ctx):
    import all_is_cubes_amd as A
    from all_is_cubes_amd import _host as H

    golden = Path(__file__).parent / "golden"

    def text_of(sp, eye, quat):
        cams = H.StandardCameras()
        cams.graphics_options = H.GraphicsOptions()
        # PrintSpace: nominal 40x40 viewport, 80x40 framebuffer (text.rs:147-182)
        vp = H.Viewport()
        vp.nominal_width, vp.nominal_height, vp.framebuffer_width, vp.framebuffer_height = 40.0, 40.0, 80, 40
        cams.viewport = vp
        cams.world_space = A.space_from_flat(sp)
        vt = H.ViewTransform()
        vt.rotation = tuple(float(v) for v in quat)
        vt.translation = tuple(float(v) for v in eye)
        cams.world_view_transform = vt
        r = H.HipRtRenderer(cams)
        r.update()
        return r.draw_text("\n")

    for sp, name in ((scenes.print_space_test_space(), "ascii_print_space.txt"), (scenes.partial_voxels_space(), "ascii_partial_voxels.txt")):
        eye = oracle.eye_for_look_at(sp.lo, sp.hi, (1.0, 1.0, 1.0))
        center = (np.array(sp.lo, float) + np.array(sp.hi, float)) / 2.0
        assert text_of(sp, eye, oracle.look_at_y_up(eye, center)) == (golden / name).read_text()
    # a busier scene, against the oracle's CharacterBuf (same pixel-centre rays): every character equal
    sp = scenes.synthetic_space(n=16, resolution=8, n_blocks=6, seed=21)
    for i, b in enumerate(sp.blocks):
        b.name = " " if b.is_air else "ABCDEFGHIJKLMNOP"[i % 16]
    eye = (8.5, 14.5, 26.0)
    q = oracle.look_at_y_up(eye, (8, 5, 8))
    want = oracle.render_text(oracle.Space(sp), oracle.make_options(), camera_for(80, 40, eye, q, aspect=1.0))
    got = text_of(sp, eye, q)
    assert got == want
    assert len(set(got) - set(" .\n")) >= 3


# --- RtScene::trace_patch for batches of rectangles (renderer.rs:418-451; raytrace_to_texture.rs:603-633) ---
@pytest.mark.parametrize("aa", [0, 2])
def test_trace_patches_equal_the_image_path(ctx, aa):
    sp = scenes.synthetic_space(n=20, resolution=8, n_blocks=6, seed=4, light="field")
    w, h = 96, 70
    eye = (10.5, 18.5, 30.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (10, 6, 10)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(antialiasing=aa))
    fr = ctx.make_frame(w, h, world_inv=inv, backdrop=(0.2, 0.1, 0.3, 0.4))
    img = ctx.render(fr, want_aux=True)
    # the rectangle of every pixel, as trace_scene_to_image_impl builds it (renderer.rs:537-550)
    ex = np.arange(w + 1, dtype=np.float64) / np.float64(w) * 2.0 - 1.0
    ey = -(np.arange(h + 1, dtype=np.float64) / np.float64(h) * 2.0 - 1.0)
    X, Y = np.meshgrid(np.arange(w), np.arange(h))
    rects = np.stack([ex[X], ey[Y], ex[X + 1], ey[Y + 1]], -1).reshape(-1, 4)
    # in a scrambled order and in two batches, like incremental refinement would ask for them
    order = np.random.default_rng(2).permutation(len(rects))
    got_rgba = np.zeros((len(rects), 4), np.uint8)
    got_aux = np.zeros(len(rects), img["aux"].dtype)
    steps = 0
    for part in (order[: len(order) // 3], order[len(order) // 3:]):
        r = ctx.trace_patches(fr, rects[part], want_aux=True)
        got_rgba[part], got_aux[part] = r["rgba8"], r["aux"]
        steps += r["info"].cubes_traced
    assert (got_rgba.reshape(h, w, 4) == img["rgba8"]).all()
    for k in ("hit", "cube", "voxel", "face", "block_index", "cubes_traced"):
        assert (got_aux[k].reshape((h, w) + got_aux[k].shape[1:]) == img["aux"][k]).all(), k
    assert steps == img["info"].cubes_traced
    # a coarser pass (2x2-pixel rectangles) equals the frame rendered at half size
    if w % 2 == 0 and h % 2 == 0:
        half = ctx.render(ctx.make_frame(w // 2, h // 2, world_inv=inv, backdrop=(0.2, 0.1, 0.3, 0.4)))
        X2, Y2 = np.meshgrid(np.arange(w // 2), np.arange(h // 2))
        ex2 = np.arange(w // 2 + 1, dtype=np.float64) / np.float64(w // 2) * 2.0 - 1.0
        ey2 = -(np.arange(h // 2 + 1, dtype=np.float64) / np.float64(h // 2) * 2.0 - 1.0)
        r2 = np.stack([ex2[X2], ey2[Y2], ex2[X2 + 1], ey2[Y2 + 1]], -1).reshape(-1, 4)
        assert (ctx.trace_patches(fr, r2)["rgba8"].reshape(h // 2, w // 2, 4) == half["rgba8"]).all()


# --- float outputs: the linear Rgba (bit-equal to the oracle's) and the ColorBuf itself --------------------
@pytest.mark.parametrize("aa", [0, 2])
def test_float_outputs(ctx, aa):
    sp = scenes.synthetic_space(n=20, resolution=8, n_blocks=6, seed=4, light="field")
    w, h = 112, 80
    eye = (10.5, 18.5, 30.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (10, 6, 10)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, abi.make_options(antialiasing=aa))
    ref = oracle.render(oracle.Space(sp), oracle.make_options(antialiasing=aa), oracle.make_camera(inv, w, h), want_linear=True)
    lin = ctx.render(ctx.make_frame(w, h, world_inv=inv, flags=abi.FRAME_OUT_LINEAR))["rgba8"]
    assert lin.dtype == np.float32 and lin.shape == (h, w, 4)
    assert (lin.view(np.uint32) == ref["linear"].view(np.uint32)).all(), "linear Rgba differs from the oracle's bit pattern"
    cb = ctx.render(ctx.make_frame(w, h, world_inv=inv, flags=abi.FRAME_OUT_COLORBUF))["rgba8"]
    # Rgba::from(ColorBuf) (raytracer_components.rs:141-163) applied on the host gives the linear image back
    t = cb[..., 3]
    alpha = np.float32(1.0) - t
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = cb[..., 0:3] / alpha[..., None]
    rgb = np.where(rgb > 0, rgb, np.float32(0.0))
    back = np.concatenate([rgb, alpha[..., None]], -1).astype(np.float32)
    back[t >= 1.0] = 0.0
    assert (back.view(np.uint32) == lin.view(np.uint32)).all()
    # and both agree with the encoded frame
    img = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    assert (img == ref["rgba8"]).all()


# --- degenerate inputs: empty spaces, one-cube spaces, blocks with empty stored volumes, R128 blocks ----
def test_degenerate_spaces_and_blocks(ctx):
    w, h = 64, 48
    eye = (0.5, 0.6, 3.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (0.5, 0.5, 0.5)), eye)
    opt = oracle.make_options()
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    cases = []
    # a Space with no cubes at all (SpaceRaytracer::new_empty-like): every ray misses
    cases.append(flat.FlatSpace((0, 0, 0), (0, 0, 0)))
    flatpl = flat.FlatSpace((3, -2, 1), (4, 0, 4))  # zero-thickness bounds
    cases.append(flatpl)
    # one cube holding a block whose stored voxel volume is empty (everything reads as AIR)
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.set_sky_uniform((0.3, 0.4, 0.5))
    pal = np.stack([flat.evoxel((0, 0, 0, 0)), flat.evoxel((1, 0, 0, 1))])
    sp.set((0, 0, 0), sp.add_block(flat.voxel_block(8, np.zeros((0, 0, 0), np.uint16), pal, vlo=(2, 2, 2))))
    cases.append(sp)
    # one cube at the largest resolution, a thin shell of voxels
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.set_sky_uniform((0.3, 0.4, 0.5))
    vox = np.zeros((128, 128, 128), np.uint16)
    vox[5:120, 64, 5:120] = 1
    vox[64, 5:120, 5:120] = 1
    sp.set((0, 0, 0), sp.add_block(flat.voxel_block(128, vox, pal)))
    cases.append(sp)
    for s in cases:
        for b in s.blocks:
            b.name = b.name or "#"
        ctx.upload_space(abi.LAYER_WORLD, s)
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv), want_aux=True)
        ref = oracle.render(oracle.Space(s), opt, oracle.make_camera(inv, w, h), want_aux=True)
        assert_parity(got, ref)
        fast = ctx.render(ctx.make_frame(w, h, world_inv=inv))
        assert (fast["rgba8"] == got["rgba8"]).all()


def test_replace_blocks_in_place_batched_and_compacted(ctx):
    """An animated block re-evaluated every tick must not grow device memory: same-size replacements are
    written in place, a batch goes under one synchronisation, blocks that outgrow their ranges are appended and the pools
    re-packed on the device -- and every state renders like a fresh upload of the same scene."""
    sp = scenes.synthetic_space(n=16, resolution=16, n_blocks=6, seed=5)
    opt = oracle.make_options()
    eye = (8.5, 14.5, 30.0)
    q = oracle.look_at_y_up(eye, (8.0, 6.0, 8.0))
    _, _, inv = oracle.camera_matrices(90.0, 200.0, 96 / 64, q, eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    recs = [i for i, b in enumerate(sp.blocks) if b.resolution > 1]
    variants = scenes.synthetic_blocks(16, 12, seed=77)

    def check():
        got = ctx.render(ctx.make_frame(96, 64, world_inv=inv), want_aux=True)
        ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, 96, 64), want_aux=True, threads=4)
        assert_parity(got, ref)

    for tick in range(40):  # same-size replacements, three blocks per tick in one call
        items = []
        for j, i in enumerate(recs[:3]):
            nb = variants[(tick + j) % len(variants)]
            sp.blocks[i] = nb
            items.append((i, nb))
        ctx.replace_blocks(abi.LAYER_WORLD, items)
    check()
    # outgrow: R32 replacements of R16 blocks (8x the voxels) force appends; repeated, they force a compaction
    big = scenes.synthetic_blocks(32, 4, seed=9)
    for tick in range(24):
        i = recs[tick % 2]
        nb = big[tick % 4] if tick % 3 else variants[tick % len(variants)]
        sp.blocks[i] = nb
        ctx.replace_block(abi.LAYER_WORLD, i, nb)
    check()
    ctx.compact(abi.LAYER_WORLD)  # the re-pack itself (also run automatically past a garbage threshold)
    check()
    ctx.replace_block(abi.LAYER_WORLD, recs[0], variants[0])  # and the table is still consistent afterwards
    sp.blocks[recs[0]] = variants[0]
    check()


def test_replace_block_voxel_atom_voxel_keeps_its_range(ctx):
    """a voxel block replaced by an atom (no voxels: its table entry holds offsets 0) and then by a voxel
    block of the old size must be written into the block's own reserved range, not over the cube grid at pool offset 0 --
    and a compaction in the atom state must move the reservation, not a copy of the cube grid."""
    sp = scenes.synthetic_space(n=12, resolution=8, n_blocks=5, seed=11)
    opt = oracle.make_options()
    eye = (6.5, 11.5, 22.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, 96 / 64, oracle.look_at_y_up(eye, (6.0, 4.0, 6.0)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    recs = [i for i, b in enumerate(sp.blocks) if b.resolution > 1]
    variants = scenes.synthetic_blocks(8, 6, seed=21)

    def check():
        got = ctx.render(ctx.make_frame(96, 64, world_inv=inv), want_aux=True)
        ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, 96, 64), want_aux=True, threads=4)
        assert_parity(got, ref)

    for k, i in enumerate(recs[:3]):
        for nb in (flat.atom((0.9, 0.1, 0.1, 1.0)), variants[k], flat.atom((0.0, 0.0, 0.0, 0.0)), variants[k + 1]):
            sp.blocks[i] = nb
            ctx.replace_block(abi.LAYER_WORLD, i, nb)
            check()
    sp.blocks[recs[0]] = flat.atom((0.1, 0.9, 0.1, 1.0))  # compaction while one reservation belongs to an atom ...
    ctx.replace_block(abi.LAYER_WORLD, recs[0], sp.blocks[recs[0]])
    ctx.compact(abi.LAYER_WORLD)
    check()
    sp.blocks[recs[0]] = variants[5]  # ... and the reservation is still usable afterwards
    ctx.replace_block(abi.LAYER_WORLD, recs[0], sp.blocks[recs[0]])
    check()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_multi_device_context_equals_single(ctx, n):
    """aic_create_multi: the single-process multi-device path behind the C ABI (what a Rust HipRtRenderer would hold).
    With every context on device 0 the strips still go through the whole partition -> peer copy -> assemble path, and
    the frame must equal the single-context frame byte for byte, step counts included; scene updates are replicated."""
    sp = scenes.synthetic_space(n=24, resolution=8, n_blocks=8, seed=11)
    opt = to_abi_options(oracle.make_options())
    eye = (12.5, 20.5, 40.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, 200 / 117, oracle.look_at_y_up(eye, (12.0, 8.0, 12.0)), eye)
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, opt)
    with abi.MultiContext([0] * n) as m:
        assert m.device_count == n
        m.upload_space(abi.LAYER_WORLD, sp)
        m.set_options(abi.LAYER_WORLD, opt)
        for size in [(200, 117), (64, 16), (33, 5)]:   # more strips than devices, exactly one strip, fewer rows than a strip
            w, h = size
            _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (12.0, 8.0, 12.0)), eye)
            one = ctx.render(ctx.make_frame(w, h, world_inv=inv))
            many = m.render(abi.Context.make_frame(w, h, world_inv=inv))
            assert (one["rgba8"] == many["rgba8"]).all()
            assert one["info"].cubes_traced == many["info"].cubes_traced
        # a replicated scene update
        xyz = np.array([[12, 10, 12], [11, 10, 12]], np.int32)
        bi = np.array([1, 1], np.uint16)
        ctx.update_cubes(abi.LAYER_WORLD, xyz, block_index=bi)
        m.update_cubes(abi.LAYER_WORLD, xyz, block_index=bi)
        w, h = 96, 64
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (12.0, 8.0, 12.0)), eye)
        assert (ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"] == m.render(abi.Context.make_frame(w, h, world_inv=inv))["rgba8"]).all()


def test_multi_device_streamed_submit_and_wait_equal_single(ctx):
    """aic_multi_render_submit / aic_multi_render_wait (ABI 3): frames queued on several slots of a three-device context (device ids repeated: one GPU)
    -- different cameras and backdrops, host and device targets, slots reused, waits out of order -- come out as the single-context frames, byte for
    byte and count for count; a busy slot is refused; aic_multi_render still works between them."""
    import torch

    sp = scenes.synthetic_space(n=24, resolution=8, n_blocks=8, seed=11)
    opt = to_abi_options(oracle.make_options())
    ctx.clear_space(abi.LAYER_UI)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    ctx.set_options(abi.LAYER_WORLD, opt)
    w, h = 200, 117
    frames = []
    for k in range(7):
        eye = (12.5 + 0.7 * k, 20.5 - 0.5 * k, 40.0 - k)
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (12.0, 8.0, 12.0)), eye)
        frames.append(abi.Context.make_frame(w, h, world_inv=inv, backdrop=(0.1 * k, 0.2, 0.3, 0.5) if k % 2 else (0, 0, 0, 0)))
    want = [ctx.render(f) for f in frames]
    with abi.MultiContext([0, 0, 0]) as m:
        m.upload_space(abi.LAYER_WORLD, sp)
        m.set_options(abi.LAYER_WORLD, opt)
        dev = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
        # three frames in flight on slots 0..2: host target, device target, host target
        o0 = m.render_submit(frames[0], 0)
        m.render_submit(frames[1], 1, dev[1].data_ptr())
        o2 = m.render_submit(frames[2], 2)
        with pytest.raises(abi.AicError):
            m.render_submit(frames[3], 1)  # busy
        i2, i0, i1 = m.render_wait(2), m.render_wait(0), m.render_wait(1)
        torch.cuda.synchronize()
        for got, info, k in ((o0, i0, 0), (dev[1].cpu().numpy(), i1, 1), (o2, i2, 2)):
            assert (got == want[k]["rgba8"]).all(), k
            assert info.cubes_traced == want[k]["info"].cubes_traced and info.rows_rendered == h
        # the synchronous call between streamed ones, then the slots again
        assert (m.render(frames[3])["rgba8"] == want[3]["rgba8"]).all()
        outs = {}
        for k in (4, 5, 6):
            slot = k % 2
            if slot in outs:
                kk, arr = outs.pop(slot)
                m.render_wait(slot)
                assert (arr == want[kk]["rgba8"]).all(), kk
            outs[slot] = (k, m.render_submit(frames[k], slot))
        for slot, (kk, arr) in outs.items():
            info = m.render_wait(slot)
            assert (arr == want[kk]["rgba8"]).all() and info.cubes_traced == want[kk]["info"].cubes_traced
        assert m.render_wait(5).cubes_traced == 0  # a slot with nothing in flight
        m.render_submit(frames[0], 3)  # never collected: destroying the context waits for the devices and copies nothing out


@pytest.mark.parametrize("k", [2, 4, 8])
def test_frames_of_one_launch_equal_the_frames_alone(ctx, synth_space, k):
    """aic_render_submit_batch (ABI 3): k frames traced by ONE launch -- every persistent workgroup bound to one of them (DevSub), each with its own
    cameras, backdrop, output, tile queues, cost record and counters -- are the frames submitted alone: bytes and per-frame step totals, for both
    production variants, with a UI layer (the pre-pass is a launch of its own over the same k frames), with antialiasing (the exchanging variant's
    sums in global memory, a region per workgroup), under a strip partition, and again when the batch is repeated (tile orders from the k cost records)."""
    import torch

    ui = scenes.synthetic_space(n=8, resolution=4, n_blocks=4, seed=5)
    w, h = 416, 232
    cams = []
    for j in range(k):
        eye = (SYNTH_EYE[0] + 1.3 * j, SYNTH_EYE[1] - 0.9 * j, SYNTH_EYE[2] - 1.1 * j)
        _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, oracle.look_at_y_up(eye, (14.0, 8.0, 14.0)), eye)
        _, _, ui_inv = oracle.camera_matrices(90.0, 200.0, w / h, (0, 0, 0, 1), (4.0 + 0.2 * j, 4.0, 12.0))
        cams.append((inv, ui_inv, (0.2, 0.1 * j, 0.4, 0.6) if j % 3 == 1 else (0, 0, 0, 0)))
    for antialiasing, with_ui, partition in ((0, False, None), (0, True, None), (2, False, None), (0, False, (16, 4, 1)), (2, True, (16, 2, 1))):
        opt = oracle.make_options(fog=1, transparency=1, lighting=3, antialiasing=antialiasing)
        ctx.upload_space(abi.LAYER_WORLD, synth_space)
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        if with_ui:
            ctx.upload_space(abi.LAYER_UI, ui)
            ctx.set_options(abi.LAYER_UI, to_abi_options(opt))
        else:
            ctx.clear_space(abi.LAYER_UI)
        rows = ctx.partition_rows(h, partition) if partition else h
        for variant in (abi.VARIANT_PLAIN, abi.VARIANT_EXCHANGING):
            frames = [ctx.make_frame(w, h, world_inv=c[0], ui_inv=c[1], backdrop=c[2], partition=partition, tuning=abi.tuning(variant=variant)) for c in cams]
            want = [ctx.render(f) for f in frames]
            bufs = [torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda") for _ in range(k)]
            for attempt in range(2):  # the second batch takes its tiles in the order of the first one's k cost records
                for b in bufs:
                    b.zero_()
                torch.cuda.synchronize()
                ctx.render_submit_batch(frames, [b.data_ptr() for b in bufs], 1)
                infos = ctx.render_wait_batch(1, k)
                torch.cuda.synchronize()
                for j in range(k):
                    assert (bufs[j].cpu().numpy() == want[j]["rgba8"]).all(), (antialiasing, with_ui, partition, variant, attempt, j)
                    assert infos[j].cubes_traced == want[j]["info"].cubes_traced and infos[j].variant == variant
            # aic_render_wait on a batch reports the sums
            ctx.render_submit_batch(frames, [b.data_ptr() for b in bufs], 2)
            assert ctx.render_wait(2).cubes_traced == sum(wj["info"].cubes_traced for wj in want)
    with pytest.raises(abi.AicError):
        ctx.render_submit_batch(frames[:1] + [ctx.make_frame(w + 8, h, world_inv=cams[0][0])], [bufs[0].data_ptr(), bufs[1].data_ptr()], 1)  # shapes differ
    with pytest.raises(abi.AicError):
        ctx.render_submit_batch(frames[:1] * 3, [b.data_ptr() for b in bufs[:3]], 1)  # three frames


@pytest.mark.parametrize("resolution", [8, 32])
def test_render_orthographic(ctx, resolution):
    """aic_render_orthographic = raytracer::ortho::render_orthographic (ortho.rs:30-88): five axis-aligned views, traced
    with UNALTERED_COLORS; against the oracle's restatement: same image size, every byte equal (colours need no powf
    here: opaque or exactly half-transparent voxels one voxel thick), step sums equal."""
    sp = scenes.synthetic_space(n=6, resolution=8, n_blocks=8, seed=21)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    got = ctx.render_orthographic(abi.LAYER_WORLD, resolution)
    ref = oracle.render_orthographic(oracle.Space(sp), resolution)
    assert got["rgba8"].shape == ref["rgba8"].shape
    d = np.abs(got["rgba8"].astype(int) - ref["rgba8"].astype(int))
    assert d.max() <= RGBA_TOL, np.bincount(d.max(axis=-1).ravel())
    assert got["info"].cubes_traced == ref["cubes_traced"]
    assert (got["rgba8"][0, 0] == 0).all()  # outside the views: transparent
    # a non-cubic space with an offset origin, through the UI layer slot
    sp2 = scenes.synthetic_space(n=5, resolution=4, n_blocks=4, seed=3)
    sp2.lo = (-7, 3, 11)
    ctx.upload_space(abi.LAYER_UI, sp2)
    got = ctx.render_orthographic(abi.LAYER_UI, 16)
    ref = oracle.render_orthographic(oracle.Space(sp2), 16)
    assert np.abs(got["rgba8"].astype(int) - ref["rgba8"].astype(int)).max() <= RGBA_TOL
    assert got["info"].cubes_traced == ref["cubes_traced"]
    ctx.clear_space(abi.LAYER_UI)


@pytest.mark.parametrize("per_launch", [1, 4])
def test_exchange_step_over_the_nccl_backend_on_one_gpu(per_launch):
    """(per_launch = 4: the rank's shares of four consecutive frames traced by one launch, aic_render_submit_batch; seven steps so that the region ends on a partial batch.)
    bench.py --gather-at-one: the N > 1 exchange step -- strip ring, collective, retire event, de-interleave -- over torch.distributed's
    nccl backend (RCCL) with a one-rank group, device-side hand-off; the assembled frame must equal the single-rank frame (the bench
    exits non-zero otherwise) and the line must say which hand-off ran. The only execution of the RCCL leg a one-GPU box allows."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gather-at-one", "--workload", "small", "--steps", "7" if per_launch > 1 else "6", "--warmup", "2",
                          "--min-seconds", "0", "--no-extras", "--no-cpu-baseline", "--frames-per-launch", str(per_launch)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["assembled_frame_equals_single_rank_frame"] is True and line["config"]["frames_per_launch"] == per_launch
    assert line["config"]["handoff"].startswith("device")
    assert out.stdout.strip().splitlines()[-1].startswith("{"), "the bench line must be the last line of the output"


@pytest.mark.parametrize("per_launch", [1, 2])
def test_two_ranks_launched_as_the_driver_launches_them(per_launch):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`, the driver's own command for N > 1, with the two ranks sharing the one
    MI355X (AIC_BENCH_ONE_GPU=1: RCCL refuses two ranks on one device, so the strips are gathered through host memory over gloo). Interleaved strips,
    the exchange pipeline, verification against a single-rank frame and the one JSON line from rank 0 are exactly what runs on a node."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, AIC_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29549",
           str(root / "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "6", "--warmup", "2", "--min-seconds", "0", "--no-extras", "--no-cpu-baseline",
           "--frames-per-launch", str(per_launch)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(root))
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["assembled_frame_equals_single_rank_frame"] is True
    assert "2 GPU" in line["config"]["partition"] or "2 GPU(s)" in line["config"]["partition"]


def test_eight_ranks_launched_as_the_driver_launches_them():
    """The driver's N = 8 command line -- `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...` -- run once before it meets a node (VERDICT
    r04 next 5): eight processes share the one MI355X (AIC_BENCH_ONE_GPU=1, strips gathered through host memory over gloo), the N >= 8 defaults -- a rank's shares
    of eight consecutive frames per launch, four launches (32 frames) in flight --, the assembled frame equal to the single-rank frame."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, AIC_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29551",
           str(root / "bench.py"), "--gpus", "8", "--workload", "small", "--steps", "20", "--warmup", "2", "--min-seconds", "0", "--no-extras", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=str(root))
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["assembled_frame_equals_single_rank_frame"] is True
    assert line["config"]["frames_in_flight"] == 32 and line["config"]["frames_per_launch"] == 8

