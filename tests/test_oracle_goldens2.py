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


def antialias_mask(sp, stand_in, size=COMMON_VIEWPORT):
    """Pixels that may show the stand-in block in an antialiasing sample: the four samples of a pixel sit at odd eighths
    of it (renderer.rs:427-433), which are pixel centres of the same view at four times the resolution; any of a pixel's
    4x4 sub-pixels counts (a superset of its four samples)."""
    w, h = size
    cam2 = spawn_camera((4 * w, 4 * h), (0.0, 0.0, 0.0), (0.4, -0.2, -1.0))
    aux = oracle.render(oracle.Space(sp), oracle.unaltered_colors(), cam2, threads=4, want_aux=True)["aux"]
    shows = (np.asarray(aux["hit"]) != 0) & (np.asarray(aux["block_index"]) == stand_in)
    cam1 = spawn_camera((w, h), (0.0, 0.0, 0.0), (0.4, -0.2, -1.0))  # ... and, without antialiasing, the pixel centre itself
    aux1 = oracle.render(oracle.Space(sp), oracle.unaltered_colors(), cam1, threads=4, want_aux=True)["aux"]
    centre = (np.asarray(aux1["hit"]) != 0) & (np.asarray(aux1["block_index"]) == stand_in)
    return shows.reshape(h, 4, w, 4).any(axis=(1, 3)) | centre


def masked_antialias_diff(golden_dir, name, img, mask):
    d = diff_to(golden_dir, name, img).max(axis=-1)
    d[mask] = 0
    return d


# cases/src/lib.rs:169-183 antialias: UNALTERED_COLORS + AntialiasingOption; Threshold [(5, 1000), (40, 1)].
# "antialias-Always-ray" is the raytracer's own golden: the 4-sample path (renderer.rs:424-451), pinned on every pixel that
# does not show the one block this repo cannot build.
@pytest.mark.parametrize("name,aa", [("antialias-None-all", 0), ("antialias-Always-ray", 2)])
def test_png_antialias(golden_dir, name, aa):
    sp, stand_in = scenes.antialias_test_space()
    cam = spawn_camera(COMMON_VIEWPORT, (0.0, 0.0, 0.0), (0.4, -0.2, -1.0))
    img = oracle.render(oracle.Space(sp), oracle.unaltered_colors(antialiasing=aa), cam, threads=4)["rgba8"]
    mask = antialias_mask(sp, stand_in)
    d = masked_antialias_diff(golden_dir, name, img, mask)
    print(name, "masked fraction", round(float(mask.mean()), 3), "difference histogram", np.bincount(d.ravel()))
    assert mask.mean() < 0.15
    assert d.max() == 0  # stricter than the case's Threshold [(5, 1000), (40, 1)]: every unmasked pixel equals the golden


# cases/src/lib.rs:1007-1051 sky(face): UNALTERED_COLORS + Linear lighting, threshold 4. Every pixel that does not show the
# cube (the labelled block this repo cannot build) is sky: the octant sky's sampling and the six view directions are pinned.
@pytest.mark.parametrize("face", ["NX", "NY", "NZ", "PX", "PY", "PZ"])
def test_png_sky(golden_dir, face):
    sp, eye, look = scenes.sky_test_space(face)
    cam = spawn_camera(COMMON_VIEWPORT, eye, look)
    out = oracle.render(oracle.Space(sp), oracle.unaltered_colors(lighting=3), cam, threads=2, want_aux=True)
    cube = np.asarray(out["aux"]["hit"]) != 0
    d = neighbourhood_diff(out["rgba8"], np.load(golden_dir / f"png_sky-{face}-all.npy")).max(axis=-1)
    grown = cube.copy()
    grown[1:, :] |= cube[:-1, :]; grown[:-1, :] |= cube[1:, :]; grown[:, 1:] |= cube[:, :-1]; grown[:, :-1] |= cube[:, 1:]
    d[grown] = 0
    print(face, "cube pixels", int(cube.sum()), "difference histogram", np.bincount(d.ravel()))
    assert 0.02 < cube.mean() < 0.5
    assert d.max() <= 4


# cases/src/lib.rs:1167-1212 viewport_zero (the frames after the viewport grows back from 0x0) and 934-946
# layers_none_but_text: with the "hello world" info text drawn over the frame (renderer.rs:659-683) as in the layers_* tests.
@pytest.mark.parametrize("name,with_world", [("viewport_zero-all", True), ("viewport_zero-2-all", True), ("layers_none_but_text-all", False)])
def test_png_text_overlay_cases(golden_dir, name, with_world):
    from tests.test_oracle_goldens import with_info_text
    cam = camera_for(*COMMON_VIEWPORT, (0.5, 0.5, 2.0))
    img = oracle.render(oracle.Space(scenes.one_cube_space()) if with_world else None, oracle.unaltered_colors(), cam)["rgba8"]
    assert diff_to(golden_dir, name, with_info_text(img)).max() <= 2  # COLOR_ROUNDING_MAX_DIFF
