"""CPU tests of the C++ host mirror (all_is_cubes_amd._host): it must reproduce the
reference's camera/viewport/options behaviour -- the vectors are those of
all-is-cubes-render/src/camera/tests.rs -- and agree with the oracle's independent
restatement bit for bit."""
import math

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import _host as H


# camera/tests.rs:76-108 (assert_eq! on exact corner values)
def test_view_frustum_exact():
    o = H.GraphicsOptions()
    o.view_distance = 10.0**2
    o.fov_y = 90.0
    c = H.Camera(o, H.Viewport.with_scale(1.0, 10, 5))
    x_near, y_near, z_near = 0.062499999999999986, 0.031249999999999993, -0.03125
    x_far, y_far, z_far = 200.00000000003973, 100.00000000001987, -100.0000000000199
    assert tuple(c.project_ndc3_into_world((-1, -1, 0))) == (-x_near, -y_near, z_near)
    assert tuple(c.project_ndc3_into_world((1, 1, 0))) == (x_near, y_near, z_near)
    assert tuple(c.project_ndc3_into_world((-1, -1, 1))) == (-x_far, -y_far, z_far)
    assert tuple(c.project_ndc3_into_world((1, 1, 1))) == (x_far, y_far, z_far)
    assert tuple(c.project_ndc3_into_world((-1, 1, 1))) == (-x_far, y_far, z_far)


# camera/tests.rs:26-47
def test_set_options_updates_matrices_and_view_position():
    c = H.Camera(H.GraphicsOptions(), H.Viewport.with_scale(1.0, 2, 2))
    before = c.projection_matrix()
    o = c.options()
    o.fov_y = 30.0
    c.set_options(o)
    assert not np.array_equal(before, c.projection_matrix())
    t = H.ViewTransform.identity()
    t.translation = (1.0, 2.0, 3.0)
    c.set_view_transform(t)
    assert tuple(c.view_position()) == (1.0, 2.0, 3.0)


# camera/tests.rs:15-24: a degenerate viewport must not panic (aspect falls back to 1)
def test_bad_viewport_does_not_raise():
    H.Camera(H.GraphicsOptions(), H.Viewport())
    assert H.Viewport().nominal_aspect_ratio() == 1.0


# camera/tests.rs:110-126
def test_post_process():
    o = H.GraphicsOptions()
    c = H.Camera(o, H.Viewport.with_scale(1.0, 2, 2))
    color = [np.float32(v) for v in (0.1, 0.2, 0.3, 0.4)]
    assert c.post_process_color(color) == [float(v) for v in color]
    o.exposure.fixed = 0.5
    c.set_options(o)
    assert c.post_process_color(color) == [float(np.float32(v) * np.float32(0.5)) for v in color[:3]] + [float(color[3])]


# camera/tests.rs:128-156
def test_exposure_automatic():
    o = H.GraphicsOptions()
    o.exposure.automatic = True
    o.lighting_display = H.LightingOption(H.LightingKind.Linear)
    c = H.Camera(o, H.Viewport.with_scale(1.0, 2, 2))
    c.set_measured_exposure(7.0)
    assert c.exposure() == 7.0
    o.lighting_display = H.LightingOption(H.LightingKind.None_)
    c = H.Camera(o, H.Viewport.with_scale(1.0, 2, 2))
    c.set_measured_exposure(7.0)
    assert c.exposure() == 1.0


# camera/tests.rs:158-162
def test_look_at_identity():
    t = H.look_at_y_up((0, 0, 0), (0, 0, -10))
    assert t.rotation == [0.0, 0.0, 0.0, 1.0] and t.translation == [0.0, 0.0, 0.0]


# camera/tests.rs:184-196
def test_viewport_is_empty():
    assert H.Viewport.with_scale(1.0, 0, 1).is_empty() and H.Viewport.with_scale(1.0, 1, 0).is_empty()
    assert not H.Viewport.with_scale(1.0, 100, 1).is_empty()


# camera/tests.rs:198-234
def test_project_ndc_into_world():
    c = H.Camera(H.GraphicsOptions(), H.Viewport.with_scale(1.0, 2, 2))
    near = c.near_plane_distance()
    o, d = c.project_ndc_into_world(0.0, 0.0)
    assert tuple(o) == (0.0, 0.0, -near) and np.allclose(d, (0, 0, -(200.0 - near)), atol=1e-6)
    t = H.ViewTransform()
    t.rotation = H.rotation_around_y(math.pi / 2)
    t.translation = (0.0, 100.0, 0.0)
    c.set_view_transform(t)
    o, d = c.project_ndc_into_world(0.0, 0.0)
    assert np.allclose(o, (-near, 100.0, 0.0), atol=1e-6) and np.allclose(d, (-(200.0 - near), 0, 0), atol=1e-6)
    o, d = c.project_ndc_into_world(float("nan"), 0.0)
    assert math.isnan(o[0]) and math.isnan(d[0])
    o, d = c.project_ndc_into_world(float("inf"), 0.0)
    assert math.isnan(o[0]) and math.isnan(d[0])


def test_graphics_options_defaults_and_repair():
    d = H.GraphicsOptions()  # graphics_options.rs:256-280
    assert d.fog == H.FogOption.Abrupt and d.lighting_display.kind == H.LightingKind.Linear
    assert d.transparency.kind == H.TransparencyKind.Volumetric and d.view_distance == 200.0 and d.fov_y == 90.0
    assert d.bloom_intensity == 0.125 and d.debug_info_text and math.isinf(d.maximum_intensity)
    u = H.GraphicsOptions.unaltered_colors()  # 168-190
    assert u.fog == H.FogOption.None_ and u.lighting_display.kind == H.LightingKind.None_ and u.bloom_intensity == 0.0
    d.fov_y, d.view_distance = 500.0, 1e9
    r = d.repair()  # 194-198
    assert r.fov_y == 189.0 and r.view_distance == 10000.0


def test_viewport_normalisation():  # viewport.rs:89-113
    v = H.Viewport.with_scale(2.0, 8, 4)
    assert v.nominal_width == 4.0 and v.nominal_aspect_ratio() == 2.0
    assert v.normalize_fb_x_edge(0) == -1.0 and v.normalize_fb_x_edge(8) == 1.0
    assert v.normalize_fb_y_edge(0) == 1.0 and v.normalize_fb_y_edge(4) == -1.0
    assert v.normalize_fb_x(0) == (0.5 / 8 * 2.0 - 1.0) and v.normalize_fb_y(3) == -((3.5) / 4 * 2.0 - 1.0)


def test_camera_matches_oracle_bit_for_bit():
    rng = np.random.default_rng(11)
    for _ in range(50):
        eye = rng.uniform(-50, 50, 3)
        target = rng.uniform(-50, 50, 3)
        fov = float(rng.uniform(20, 120))
        vd = float(rng.uniform(10, 1000))
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        o = H.GraphicsOptions()
        o.fov_y, o.view_distance = fov, vd
        c = H.Camera(o, H.Viewport.with_scale(1.0, w, h))
        c.look_at_y_up(tuple(eye), tuple(target))
        q = oracle.look_at_y_up(eye, target)
        assert list(q) == c.view_transform().rotation
        _, w2e, inv = oracle.camera_matrices(fov, vd, w / h, q, eye)
        assert (c.inverse_projection_view().view(np.uint64) == inv.view(np.uint64)).all()
        assert (c.view_matrix().view(np.uint64) == w2e.view(np.uint64)).all()


def test_sky_for_blocks_and_packed_light_match_oracle():
    from tests import scenes

    assert H.PackedLight.one().as_texel() == [144, 144, 144, 255]  # light/data.rs:74-80
    assert H.PackedLight.scalar_in(0.5) == 134 and H.PackedLight.scalar_in(0.0) == 0 and H.PackedLight.scalar_in(1e30) == 255
    sp = scenes.one_cube_space()
    rng = np.random.default_rng(5)
    s = H.Sky()
    s.set_uniform((0.5, 0.25, 1.5))
    sp.set_sky_uniform((0.5, 0.25, 1.5))
    assert (s.for_blocks() == oracle.block_sky(oracle.Space(sp))).all()
    oct_ = rng.uniform(0.0, 4.0, (8, 3)).astype(np.float32)
    s.set_octants(oct_)
    sp.set_sky_octants(oct_)
    assert (s.for_blocks() == oracle.block_sky(oracle.Space(sp))).all()


def test_space_validates_like_the_reference():
    sp = H.Space((0, 0, 0), (2, 2, 2))
    a = sp.add_block(H.Evoxels.air())
    sp.fill_all(a)
    with pytest.raises(IndexError):
        sp.set(0, 0, 0, 7)  # block index not in the palette
    with pytest.raises(IndexError):
        sp.set(5, 0, 0, a)  # cube out of bounds


def test_renderer_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(H.RenderError):
        H.HipRtRenderer(H.StandardCameras())
