"""A second batch of the reference's golden images (test-renderers/expected/renderers/, cases/src/lib.rs) against the
oracle: debug_pixel_cost, the four "white furnace" cases (light updater from a built-then-mutated space + tone mapping +
Physical fog + volumetric transparency), bloom at intensity 0, no_update, follow_options_change (never-evaluated light,
Threshold transparency, exposure), and the cornell-box template. Thresholds are the ones the cases state."""
import copy

import numpy as np
import pytest

import oracle
from tests import scenes
from tests.test_oracle_goldens import COMMON_VIEWPORT, camera_for, diff_to, histogram_ok, neighbourhood_diff
from tests.test_oracle_light import image_diff, lit, spawn_camera


# cases/src/lib.rs:286-294 debug_pixel_cost on fog_test_universe: Threshold [(2, 500), (15, 100)]; a raytracer-only golden
def test_png_debug_pixel_cost(golden_dir):
    sp = lit(scenes.fog_test_space)
    cam = spawn_camera(COMMON_VIEWPORT, (0.0, 10.0, 0.0), (0.4, 0.0, -1.0))
    img = oracle.render(oracle.Space(sp), oracle.unaltered_colors(debug_pixel_cost=True), cam, threads=4)["rgba8"]
    d = diff_to(golden_dir, "debug_pixel_cost-ray", img)
    assert d.max() == 0, np.bincount(d.max(axis=-1).ravel())  # the step counts of every pixel, exactly


def furnace_lit(transparent, _cache={}):
    if transparent not in _cache:
        sp, queue = scenes.furnace_space(transparent)
        oracle.evaluate_light(sp, maximum_distance=30, fast=False, epsilon=0, batch=32, queue=queue, hb_width=16)  # m.evaluate_light(0, drop)
        _cache[transparent] = sp
    return _cache[transparent]


def furnace_options(foggy):
    # GraphicsOptions::default() with fov 45, bloom 0, view distance 10 and fog None / Physical (cases/src/lib.rs:649-660)
    return oracle.make_options(fog=3 if foggy else 0, view_distance=10.0)


FURNACE = [("furnace-Clear-Opaque-all", False, False), ("furnace-Clear-Transparent-all", False, True),
           ("furnace-Foggy-Opaque-all", True, False), ("furnace-Foggy-Transparent-all", True, True)]


# cases/src/lib.rs:620-664: threshold 1
@pytest.mark.parametrize("name,foggy,transparent", FURNACE)
def test_png_furnace(golden_dir, name, foggy, transparent):
    sp = furnace_lit(transparent)
    cam = spawn_camera(COMMON_VIEWPORT, (-3.0, 4.0, 4.0), (1.0, -1.0, -1.0), fov=45.0, view_distance=10.0)
    img = oracle.render(oracle.Space(sp), furnace_options(foggy), cam, threads=4)["rgba8"]
    d = image_diff(golden_dir, name, img)
    assert d.max() <= 1, np.bincount(d.max(axis=-1).ravel())


def test_furnace_light_is_the_sky_everywhere_it_was_computed():
    """The point of a white furnace: every texel the updater touched carries the sky's light."""
    sp = furnace_lit(False)
    lt = np.asarray(sp.light).reshape(-1, 4)
    visible = lt[lt[:, 3] == 255]
    sky = oracle.packed_light_from_rgb((0.75, 0.75, 0.75)) if hasattr(oracle, "packed_light_from_rgb") else None
    assert len(visible) > 0
    assert (visible[:, :3] == visible[0, :3]).all()
    if sky is not None:
        assert tuple(visible[0, :3]) == tuple(sky[:3])


# cases/src/lib.rs:186-202 bloom(0.0): light_test_options (UNALTERED_COLORS, Linear lighting, fov 45), 128x256; threshold 12
def test_png_bloom_zero(golden_dir):
    sp = scenes.bloom_test_space()
    cam = spawn_camera((128, 256), (1.5, 3.0, 8.0), (0.0, 0.0, -1.0), fov=45.0)
    img = oracle.render(oracle.Space(sp), oracle.unaltered_colors(lighting=3), cam, threads=4)["rgba8"]
    d = image_diff(golden_dir, "bloom-0.0-all", img)
    assert d.max() <= 12, np.bincount(d.max(axis=-1).ravel())


# cases/src/lib.rs:988-1005 no_update: draw() before any update() shows no world; after update(), the cube. Threshold 5
def test_png_no_update(golden_dir):
    cam = camera_for(*COMMON_VIEWPORT, (0.5, 0.5, 2.0))
    before = oracle.render(None, oracle.unaltered_colors(), cam)["rgba8"]
    assert diff_to(golden_dir, "no_update-all", before).max() == 0
    after = oracle.render(oracle.Space(scenes.one_cube_space()), oracle.unaltered_colors(), cam)["rgba8"]
    assert image_diff(golden_dir, "no_update-2-all", after).max() <= 5


# cases/src/lib.rs:560-603 follow_options_change: threshold 1 for both frames
def test_png_follow_options_change(golden_dir):
    sp = scenes.follow_options_space()
    o1 = oracle.unaltered_colors(lighting=3)
    img1 = oracle.render(oracle.Space(sp), o1, camera_for(*COMMON_VIEWPORT, (0.5, 0.5, 2.0), fov=90.0), threads=2)["rgba8"]
    d1 = image_diff(golden_dir, "follow_options_change-all", img1)
    assert d1.max() <= 1, np.bincount(d1.max(axis=-1).ravel())
    o2 = oracle.unaltered_colors(lighting=3, transparency=2, threshold=0.1)
    o2.exposure = 1.5
    img2 = oracle.render(oracle.Space(sp), o2, camera_for(*COMMON_VIEWPORT, (0.5, 0.5, 2.0), fov=70.0), threads=2)["rgba8"]
    d2 = image_diff(golden_dir, "follow_options_change-2-all", img2)
    assert d2.max() <= 1, np.bincount(d2.max(axis=-1).ravel())


# cases/src/lib.rs:1054-1105 template("cornell-box"): UNALTERED_COLORS; Threshold [(254, 20), (30, 50), (1, all)]
def test_png_template_cornell_box(golden_dir):
    sp = scenes.cornell_box_space()
    box = 28.0
    cam = spawn_camera(COMMON_VIEWPORT, (0.5 * box, 0.5 * box, 1.6 * box), (0.0, 0.0, -1.0))
    img = oracle.render(oracle.Space(sp), oracle.unaltered_colors(), cam, threads=4)["rgba8"]
    d = diff_to(golden_dir, "template-cornell-box-all", img)
    assert histogram_ok(d, [(254, 20), (30, 50), (1, 1 << 60)]), np.bincount(d.max(axis=-1).ravel())
    print("cornell-box difference histogram", np.bincount(d.max(axis=-1).ravel()))
