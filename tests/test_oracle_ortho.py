"""Pins of the oracle's axis-aligned raycaster and orthographic renderer (oracle/aic_ortho.inc):
the reference's own AxisAlignedRaycaster tests (all-is-cubes-base/src/raycast/axis_aligned.rs:215-394), which compare it
step by step with Raycaster on the equivalent Ray, and the AaRay doc-tests (raycast/ray.rs:183-235)."""
import numpy as np
import pytest

import oracle
from all_is_cubes_amd import flat
from tests import scenes

NX, NY, NZ, PX, PY, PZ = 1, 2, 3, 4, 5, 6
I32_MAX, I32_MIN = 2**31 - 1, -(2**31)


def compare_aa_to_regular(origin, direction, bounds=None, include_exit=True, take=10, **kw):
    """axis_aligned.rs:225-249: every step of the two raycasters must be equal (cube, face, t_distance, t_max)."""
    aa, aa_ended, (ro, rd) = oracle.aa_raycast(origin, direction, bounds, include_exit, max_steps=take, **kw)
    arb, arb_ended = oracle.raycast(ro, rd, bounds, include_exit, max_steps=take)
    assert len(aa) == len(arb), (aa["cube"], arb["cube"])
    for a, b in zip(aa, arb):
        assert tuple(a["cube"]) == tuple(b["cube"]) and a["face"] == b["face"]
        assert a["t_distance"] == b["t_distance"]
        assert tuple(a["t_max"]) == tuple(b["t_max"])
    assert aa_ended == arb_ended
    return [tuple(int(v) for v in s["cube"]) for s in aa]


B3 = ((0, 0, 0), (3, 3, 3))
BIG = ((-10, -20, -30), (10, 20, 30))


def test_unbounded():  # unbounded_positive / unbounded_negative
    assert len(compare_aa_to_regular((1, 2, 3), PX)) == 10
    assert len(compare_aa_to_regular((1, 2, 3), NX)) == 10


def test_start_in_bounds():
    assert compare_aa_to_regular((1, 1, 1), PX, B3) == [(1, 1, 1), (2, 1, 1), (3, 1, 1)]


def test_start_out_of_bounds():
    assert compare_aa_to_regular((-1, 1, 1), PX, B3) == [(0, 1, 1), (1, 1, 1), (2, 1, 1), (3, 1, 1)]
    assert compare_aa_to_regular((4, 1, 1), NX, B3) == [(2, 1, 1), (1, 1, 1), (0, 1, 1), (-1, 1, 1)]


def test_ray_misses_bounds():
    assert compare_aa_to_regular((1000, 0, -1000), NX, BIG) == []
    assert compare_aa_to_regular((-17, -1, 52), PX, BIG) == []


def test_exiting_integer_limits():
    assert compare_aa_to_regular((I32_MAX - 3, 10, 20), PX) == [(I32_MAX - 3, 10, 20), (I32_MAX - 2, 10, 20), (I32_MAX - 1, 10, 20)]
    assert compare_aa_to_regular((I32_MIN + 2, 10, 20), NX) == [(I32_MIN + 2, 10, 20), (I32_MIN + 1, 10, 20), (I32_MIN, 10, 20)]


def test_every_direction_and_exit_flag():
    rng = np.random.default_rng(3)
    for _ in range(200):
        origin = tuple(int(v) for v in rng.integers(-6, 9, 3))
        direction = int(rng.integers(1, 7))
        sub = rng.integers(0, 32, 3).astype(np.float32) / np.float32(32.0) + np.float32(1 / 64)
        compare_aa_to_regular(origin, direction, B3, bool(rng.integers(0, 2)), take=12, sub_origin=sub)


# ray.rs:183-193, 219-235
def test_aaray_doc_examples():
    _, _, (o, d) = oracle.aa_raycast((1, 2, 3), PX, max_steps=1)
    assert tuple(o) == (1.5, 2.5, 3.5) and tuple(d) == (1.0, 0.0, 0.0)


def test_zoom_in_visits_the_same_voxels_as_the_general_sub_ray():
    """RaycasterIsh::recursive_raycast for AaRay = zoom_in (raycast_traits.rs:78-91): inside a block the axis-aligned and
    the general raycaster visit the same voxels through the same faces (their t origins differ; only differences of t
    enter a colour)."""
    rng = np.random.default_rng(5)
    res = 16
    bounds = ((0, 0, 0), (res, res, res))
    for _ in range(100):
        cube = tuple(int(v) for v in rng.integers(-3, 4, 3))
        direction = int(rng.integers(1, 7))
        axis = (direction - 1) % 3
        origin = list(cube)
        back = int(rng.integers(0, 6))  # the ray starts inside the cube or behind it (zoom_in is only used on cubes it traverses)
        origin[axis] += -back if direction >= PX else back
        sub = (rng.integers(0, res, 3).astype(np.float32) + np.float32(0.5)) / np.float32(res)
        if back:
            # as the orthographic cameras cast them: from a cube face (an origin that lies partway through a cube BEHIND the
            # block makes zoom_in add that fraction to the entry voxel, ray.rs:279-293 -- a reference quirk no caller meets)
            sub[axis] = 0.0
        aa, _, _ = oracle.aa_raycast(tuple(origin), direction, bounds, True, max_steps=40, sub_origin=sub, zoom=(cube, res))
        # the general path: Ray::from(aa_ray) -> sub_ray.origin = (origin - cube) * resolution (raycast.rs:458-476)
        _, _, (ro, rd) = oracle.aa_raycast(tuple(origin), direction, sub_origin=sub, max_steps=1)
        sub_o = (ro - np.array(cube, float)) * res
        arb, _ = oracle.raycast(sub_o, rd, bounds, True, max_steps=40)
        assert [tuple(s["cube"]) for s in aa] == [tuple(s["cube"]) for s in arb]
        assert [int(s["face"]) for s in aa][1:] == [int(s["face"]) for s in arb][1:]
        assert np.array_equal(np.diff(aa["t_distance"])[1:], np.diff(arb["t_distance"])[1:])


def test_multi_ortho_layout():
    """MultiOrthoCamera::new (ortho.rs:147-184): top / left / front / right / bottom views around the front view."""
    (w, h), rect, tr, di = oracle.ortho_views((0, 0, 0), (2, 3, 4), 8)
    top, left, front, right, bottom = rect
    assert tuple(top) == (4 * 8 + 1, 0, 2 * 8, 4 * 8) and tuple(left) == (0, 4 * 8 + 1, 4 * 8, 3 * 8)
    assert tuple(front) == (4 * 8 + 1, 4 * 8 + 1, 2 * 8, 3 * 8) and tuple(right) == (4 * 8 + 2 * 8 + 2, 4 * 8 + 1, 4 * 8, 3 * 8)
    assert tuple(bottom) == (4 * 8 + 1, 4 * 8 + 3 * 8 + 2, 2 * 8, 4 * 8)
    assert (w, h) == (4 * 8 + 2 * 8 + 2 + 4 * 8, 4 * 8 + 3 * 8 + 2 + 4 * 8)
    # view directions: looking AT the viewed face, i.e. along its inward normal
    # (length 1 / resolution: TryFrom<Ray> for AaRay keeps only the axis, ray.rs:338-348)
    assert [tuple(int(c) for c in np.sign(v)) for v in di] == [(0, -1, 0), (1, 0, 0), (0, 0, -1), (-1, 0, 0), (0, 1, 0)]


def test_render_orthographic_views_of_a_marked_cube():
    """One 2x2x2 space whose +Y faces are red, everything else green: the top view shows red, the others the sides."""
    sp = flat.FlatSpace((0, 0, 0), (2, 2, 2))
    green = sp.add_block(flat.atom((0.0, 1.0, 0.0, 1.0)))
    sp.block_index[...] = green
    red = sp.add_block(flat.atom((1.0, 0.0, 0.0, 1.0)))
    sp.block_index[:, 1, :] = red
    out = oracle.render_orthographic(oracle.Space(sp), 4)
    img = out["rgba8"]
    (w, h), rect, _, _ = oracle.ortho_views((0, 0, 0), (2, 2, 2), 4)
    assert img.shape == (h, w, 4)
    top, left, front, right, bottom = rect
    assert (img[top[1] : top[1] + top[3], top[0] : top[0] + top[2]] == (255, 0, 0, 255)).all()
    assert (img[bottom[1] : bottom[1] + bottom[3], bottom[0] : bottom[0] + bottom[2]] == (0, 255, 0, 255)).all()
    f = img[front[1] : front[1] + front[3], front[0] : front[0] + front[2]]
    assert (f[: front[3] // 2] == (255, 0, 0, 255)).all() and (f[front[3] // 2 :] == (0, 255, 0, 255)).all()   # image y runs down
    assert (img[0, 0] == (0, 0, 0, 0)).all()  # between the views: transparent
    assert out["cubes_traced"] > 0
