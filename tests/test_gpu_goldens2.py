"""GPU tests (-m gpu): the second batch of the reference's golden images (tests/test_oracle_goldens2.py) through the HIP
path and the C ABI -- bit-exact against the oracle on the same scene, and against the golden PNGs at the cases' thresholds.
The furnace spaces are lit ON THE DEVICE (aic_evaluate_light from the queue that the block placements leave)."""
import copy

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat
from tests import scenes
from tests.test_gpu_light import render_both
from tests.test_gpu_parity import assert_parity, to_abi_options
from tests.test_oracle_goldens import COMMON_VIEWPORT, diff_to, histogram_ok
from tests.test_oracle_goldens2 import FURNACE, furnace_lit, furnace_options
from tests.test_oracle_light import image_diff, lit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


def test_debug_pixel_cost(ctx, golden_dir):
    got, ref = render_both(ctx, lit(scenes.fog_test_space), oracle.unaltered_colors(debug_pixel_cost=True), COMMON_VIEWPORT,
                           (0.0, 10.0, 0.0), (0.4, 0.0, -1.0))
    assert_parity(got, ref)
    assert diff_to(golden_dir, "debug_pixel_cost-ray", got["rgba8"]).max() == 0


@pytest.mark.parametrize("name,foggy,transparent", FURNACE)
def test_furnace_lit_on_the_device(ctx, golden_dir, name, foggy, transparent):
    sp, queue = scenes.furnace_space(transparent)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 30, fast=False, epsilon=0, batch=32, queue_order=16, queue=queue)
    light = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    want = furnace_lit(transparent)
    assert info.updates > 0 and (light == np.asarray(want.light).reshape(light.shape)).all()
    sp.light = light
    got, ref = render_both(ctx, sp, furnace_options(foggy), COMMON_VIEWPORT, (-3.0, 4.0, 4.0), (1.0, -1.0, -1.0), fov=45.0)
    assert_parity(got, ref)
    assert image_diff(golden_dir, name, got["rgba8"]).max() <= 1


def test_bloom_zero(ctx, golden_dir):
    got, ref = render_both(ctx, scenes.bloom_test_space(), oracle.unaltered_colors(lighting=3), (128, 256), (1.5, 3.0, 8.0), (0.0, 0.0, -1.0), fov=45.0)
    assert_parity(got, ref)
    assert image_diff(golden_dir, "bloom-0.0-all", got["rgba8"]).max() <= 12


def test_no_update(ctx, golden_dir):
    w, h = COMMON_VIEWPORT
    q = oracle.look_at_y_up((0.5, 0.5, 2.0), (0.5, 0.5, 1.0))
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, q, (0.5, 0.5, 2.0))
    ctx.clear_space(abi.LAYER_WORLD)
    ctx.clear_space(abi.LAYER_UI)
    before = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    assert diff_to(golden_dir, "no_update-all", before).max() == 0
    got, ref = render_both(ctx, scenes.one_cube_space(), oracle.unaltered_colors(), COMMON_VIEWPORT, (0.5, 0.5, 2.0), (0.0, 0.0, -1.0))
    assert_parity(got, ref)
    assert image_diff(golden_dir, "no_update-2-all", got["rgba8"]).max() <= 5


def test_follow_options_change(ctx, golden_dir):
    sp = scenes.follow_options_space()
    got, ref = render_both(ctx, sp, oracle.unaltered_colors(lighting=3), COMMON_VIEWPORT, (0.5, 0.5, 2.0), (0.0, 0.0, -1.0), fov=90.0)
    assert_parity(got, ref)
    assert image_diff(golden_dir, "follow_options_change-all", got["rgba8"]).max() <= 1
    got, ref = render_both(ctx, sp, oracle.unaltered_colors(lighting=3, transparency=2, threshold=0.1), COMMON_VIEWPORT, (0.5, 0.5, 2.0),
                           (0.0, 0.0, -1.0), fov=70.0, exposure=1.5)
    assert_parity(got, ref)
    assert image_diff(golden_dir, "follow_options_change-2-all", got["rgba8"]).max() <= 1


def test_template_cornell_box(ctx, golden_dir):
    box = 28.0
    got, ref = render_both(ctx, scenes.cornell_box_space(), oracle.unaltered_colors(), COMMON_VIEWPORT, (0.5 * box, 0.5 * box, 1.6 * box), (0.0, 0.0, -1.0))
    assert_parity(got, ref)
    d = diff_to(golden_dir, "template-cornell-box-all", got["rgba8"])
    assert histogram_ok(d, [(254, 20), (30, 50), (1, 1 << 60)])
    assert d.max() == 0


def test_cornell_box_lit_on_the_device(ctx):
    """The template's own light physics (Rays, maximum_distance 56) and its fast_evaluate_light, then a bounded number of
    updates: device and oracle agree texel for texel on a space of 27 000 cubes with an emitter."""
    sp = scenes.cornell_box_space()
    ref = copy.deepcopy(sp)
    n_ref = oracle.evaluate_light(ref, maximum_distance=56, fast=True, epsilon=1, batch=32, max_updates=2000, hb_width=16)
    ctx.upload_space(abi.LAYER_WORLD, sp)
    info = ctx.evaluate_light(abi.LAYER_WORLD, 56, fast=True, epsilon=1, batch=32, queue_order=16, max_updates=2000)
    got = ctx.read_light_volume(abi.LAYER_WORLD, sp.size)
    assert info.updates == n_ref
    assert (got == np.asarray(ref.light).reshape(got.shape)).all()


@pytest.mark.parametrize("name,aa", [("antialias-None-all", 0), ("antialias-Always-ray", 2)])
def test_antialias(ctx, golden_dir, name, aa):
    """The 4-sample antialiasing path (renderer.rs:424-451) against the raytracer's own golden, on every pixel that does not
    show the one block of the scene that needs the reference's text engine (tests/test_oracle_goldens2.py)."""
    from tests.test_oracle_goldens2 import antialias_mask, masked_antialias_diff
    sp, stand_in = scenes.antialias_test_space()
    got, ref = render_both(ctx, sp, oracle.unaltered_colors(antialiasing=aa), COMMON_VIEWPORT, (0.0, 0.0, 0.0), (0.4, -0.2, -1.0))
    assert_parity(got, ref)
    assert masked_antialias_diff(golden_dir, name, got["rgba8"], antialias_mask(sp, stand_in)).max() == 0


@pytest.mark.parametrize("face", ["NX", "NY", "NZ", "PX", "PY", "PZ"])
def test_sky(ctx, golden_dir, face):
    """The octant sky from six directions against the `sky-*` goldens, outside the pixels of the cube (whose block needs
    the reference's text engine)."""
    from tests.test_oracle_goldens import neighbourhood_diff
    sp, eye, look = scenes.sky_test_space(face)
    got, ref = render_both(ctx, sp, oracle.unaltered_colors(lighting=3), COMMON_VIEWPORT, eye, look)
    assert_parity(got, ref)
    cube = np.asarray(got["aux"]["hit"]) != 0
    d = neighbourhood_diff(got["rgba8"], np.load(golden_dir / f"png_sky-{face}-all.npy")).max(axis=-1)
    grown = cube.copy()
    grown[1:, :] |= cube[:-1, :]; grown[:-1, :] |= cube[1:, :]; grown[:, 1:] |= cube[:, :-1]; grown[:, :-1] |= cube[:, 1:]
    d[grown] = 0
    assert d.max() <= 4


@pytest.mark.parametrize("name,with_world", [("viewport_zero-all", True), ("viewport_zero-2-all", True), ("layers_none_but_text-all", False)])
def test_text_overlay_cases(ctx, golden_dir, name, with_world):
    """Frames whose goldens carry the "hello world" info text (renderer.rs:659-683): drawn over the device's frame by the host
    mirror's draw_info_text, then the whole image is compared."""
    from tests.test_oracle_goldens import with_info_text
    w, h = COMMON_VIEWPORT
    if with_world:
        ctx.upload_space(abi.LAYER_WORLD, scenes.one_cube_space())
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(oracle.unaltered_colors()))
    else:
        ctx.clear_space(abi.LAYER_WORLD)
    ctx.clear_space(abi.LAYER_UI)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, w / h, (0, 0, 0, 1), (0.5, 0.5, 2.0))
    # a zero-area viewport first, as the case does: an empty image, and the renderer is fine afterwards
    assert ctx.render(ctx.make_frame(0, 0, world_inv=inv))["rgba8"].shape == (0, 0, 4)
    img = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    assert diff_to(golden_dir, name, with_info_text(img)).max() <= 2


@pytest.mark.parametrize("scale,name", [(1.0, "info_text-1.0-all"), (1.5, "info_text-1.5-ray"), (2.0, "info_text-2.0-ray")])
def test_info_text_through_the_headless_renderer(golden_dir, scale, name):
    """cases/src/lib.rs:667-711 info_text, end to end through the host mirror of HeadlessRenderer: an empty space with an orange
    sky traced on the device, the info text drawn over the frame by HipRtRenderer::draw (renderer.rs:205-217, 659-683; the
    default GraphicsOptions::debug_info_text is on). The reference's own expected images, pixel for pixel."""
    import all_is_cubes_amd as A
    from all_is_cubes_amd import _host as H
    from tests.test_info_text import INFO_TEXT

    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.set_sky_uniform((1.0, 0.5, 0.0))
    sp.add_block(flat.air())
    cams = H.StandardCameras()
    cams.graphics_options = H.GraphicsOptions.unaltered_colors()
    cams.viewport = H.Viewport.with_scale(1.0, int(128 * scale), int(96 * scale))
    cams.world_space = A.space_from_flat(sp)
    cams.world_view_transform = H.look_at_y_up((0.5, 0.5, 2.0), (0.5, 0.5, 0.0))
    r = H.HipRtRenderer(cams)
    r.update()
    img = r.draw(INFO_TEXT)
    ref = np.load(golden_dir / f"png_{name}.npy")
    assert img.data.shape == ref.shape and (img.data == ref).all()
    assert not (img.flaws & H.Flaws.OTHER)
    plain = r.draw("")  # no text: the bare sky
    assert (plain.data == np.array([255, 188, 0, 255], np.uint8)).all()
