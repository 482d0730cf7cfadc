"""GPU tests (-m gpu) of the lighting scenes: the lit spaces of the reference's lighting image tests
(test-renderers/cases/src/lib.rs:501-512, 976-983, 1107-1135), rendered by the HIP path through the C ABI, against
(a) the oracle on the same lit space -- bit-exact hits / steps, RGBA8 +-1 -- and (b) the reference's golden PNGs at the
thresholds the cases state (tests/test_oracle_light.py explains the fog-* bound)."""
import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi
from tests import scenes
from tests.test_gpu_parity import assert_parity, to_abi_options
from tests.test_oracle_goldens import COMMON_VIEWPORT, histogram_ok
from tests.test_oracle_light import EXACT, FOG_BOUND, LIGHTING, image_diff, lit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


def render_both(ctx, space, opt, size, eye, look, fov=90.0, exposure=1.0):
    w, h = size
    q = oracle.look_at_y_up(eye, tuple(e + l for e, l in zip(eye, look)))
    _, _, inv = oracle.camera_matrices(fov, opt.view_distance, w / h, q, eye)
    ctx.upload_space(abi.LAYER_WORLD, space)
    ctx.clear_space(abi.LAYER_UI)
    ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
    got = ctx.render(ctx.make_frame(w, h, world_inv=inv, exposure=exposure), want_aux=True)
    opt.exposure = exposure
    ref = oracle.render(oracle.Space(space), opt, oracle.make_camera(inv, w, h), want_aux=True, threads=4)
    return got, ref


@pytest.mark.parametrize("option", list(LIGHTING))
def test_light_spread(ctx, golden_dir, option):
    got, ref = render_both(ctx, lit(scenes.light_spread_space), oracle.unaltered_colors(lighting=LIGHTING[option]), COMMON_VIEWPORT,
                           (0.0, 0.0, 8.0), (0.0, 0.0, -1.0), fov=45.0)
    assert_parity(got, ref)
    name = f"light_spread-{option}-all"
    assert image_diff(golden_dir, name, got["rgba8"]).max() <= 7
    if name in EXACT:
        assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / f"png_{name}.npy").astype(int)).max() <= 1


@pytest.mark.parametrize("option", list(LIGHTING))
def test_light_on_slab(ctx, golden_dir, option):
    got, ref = render_both(ctx, lit(scenes.light_on_slab_space), oracle.unaltered_colors(lighting=LIGHTING[option]), COMMON_VIEWPORT,
                           (0.5, -6.0, 6.0), (0.0, 1.0, -1.0), fov=45.0)
    assert_parity(got, ref)
    name = f"light_on_slab-{option}-all"
    assert image_diff(golden_dir, name, got["rgba8"]).max() <= 7
    if name in EXACT:
        assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / f"png_{name}.npy").astype(int)).max() <= 1


@pytest.mark.parametrize("name,fog", [("fog-None-ray", 0), ("fog-Abrupt-all", 1), ("fog-Compromise-all", 2), ("fog-Physical-all", 3)])
def test_fog(ctx, golden_dir, name, fog):
    got, ref = render_both(ctx, lit(scenes.fog_test_space), oracle.unaltered_colors(lighting=3, fog=fog, view_distance=50.0), COMMON_VIEWPORT,
                           (0.0, 10.0, 0.0), (0.4, 0.0, -1.0))
    assert_parity(got, ref)
    assert histogram_ok(image_diff(golden_dir, name, got["rgba8"]), FOG_BOUND)


@pytest.mark.parametrize(
    "name,tmo,maximum_intensity,exposure",
    [("tone_map-Clamp-1.0-0.5-all", 0, 1.0, 0.5), ("tone_map-Clamp-1.0-2.0-all", 0, 1.0, 2.0), ("tone_map-Reinhard-0.5-0.5-all", 1, 0.5, 0.5),
     ("tone_map-Reinhard-1.0-0.5-all", 1, 1.0, 0.5), ("tone_map-Reinhard-1.0-2.0-all", 1, 1.0, 2.0)],
)
def test_tone_map(ctx, golden_dir, name, tmo, maximum_intensity, exposure):
    sp = lit(scenes.tone_mapping_space)
    lo, size = np.array(sp.lo, float), np.array(sp.size, float)
    eye = tuple(lo + size / 2.0 + np.array([0.0, 0.0, 65.0]))
    opt = oracle.unaltered_colors(lighting=1, tone_mapping=tmo, maximum_intensity=maximum_intensity)
    got, ref = render_both(ctx, sp, opt, (256, 320), eye, (0.0, 0.0, -1.0), fov=45.0, exposure=exposure)
    assert_parity(got, ref)
    assert histogram_ok(image_diff(golden_dir, name, got["rgba8"]), [(10, 100), (3, 500), (1, 1 << 60)])
    assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / f"png_{name}.npy").astype(int)).max() <= 1


# cases/src/lib.rs:1054-1105 template("light-bench"): the scene of the reference's raytracer benchmark
def test_template_light_bench(ctx, golden_dir):
    sp = scenes.light_bench_space()
    direction = (0.0, 0.5, 1.0)
    eye = tuple(oracle.eye_for_look_at(sp.lo, sp.hi, direction))
    got, ref = render_both(ctx, sp, oracle.unaltered_colors(), COMMON_VIEWPORT, eye, tuple(-d for d in direction))
    assert_parity(got, ref)
    assert np.abs(got["rgba8"].astype(int) - np.load(golden_dir / "png_template-light-bench-all.npy").astype(int)).max() <= 1
