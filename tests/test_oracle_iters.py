"""Pins the oracle's SurfaceIter / DepthIter / DepthBuf / ColorBuf / apply_transmittance
against the literal tables in the reference's unit tests:
all-is-cubes-render/src/raytracer/surface.rs:541-855, accum.rs:386-496,
all-is-cubes/src/raytracer_components.rs:269-331."""
import numpy as np
import pytest

import oracle
from tests import scenes

ENTER_SURFACE, INVISIBLE, ENTER_BLOCK = 0, 1, 2
D_INVISIBLE, D_SPAN, D_ENTER_BLOCK = 10, 11, 12
NX, NY = 1, 2


def surf(step):
    return (
        tuple(float(v) for v in step["color"]),
        tuple(float(v) for v in step["emission"]),
        tuple(int(v) for v in step["cube"]),
        int(step["resolution"]),
        tuple(int(v) for v in step["voxel"]),
        float(step["t_distance"]),
        tuple(float(v) for v in step["intersection_point"]),
        int(step["normal"]),
    )


RED = (1.0, 0.0, 0.0, 1.0)
YEL = (1.0, 1.0, 0.0, 1.0)
Z3 = (0.0, 0.0, 0.0)


# surface.rs:541-679
def test_surface_and_depth_iter_basic():
    sp = oracle.Space(scenes.surface_iter_basic_space())
    origin, direction = (0.25, -0.5, 0.25), (0.0, 1.0, 0.0)
    steps = oracle.surface_iter(sp, origin, direction)
    kinds = [int(s["kind"]) for s in steps]
    assert kinds == [INVISIBLE, ENTER_SURFACE, ENTER_BLOCK, ENTER_SURFACE, ENTER_SURFACE, INVISIBLE, INVISIBLE, INVISIBLE, INVISIBLE]
    assert steps[0]["t_distance"] == 0.5
    assert surf(steps[1]) == (RED, Z3, (0, 1, 0), 1, (0, 0, 0), 1.5, (0.25, 1.0, 0.25), NY)
    assert steps[2]["t_distance"] == 2.5
    assert surf(steps[3]) == (YEL, Z3, (0, 2, 0), 4, (1, 0, 1), 2.5, (0.25, 2.0, 0.25), NY)
    assert surf(steps[4]) == (YEL, Z3, (0, 2, 0), 4, (1, 1, 1), 2.75, (0.25, 2.25, 0.25), NY)
    assert [float(s["t_distance"]) for s in steps[5:]] == [3.0, 3.25, 3.5, 3.5]

    d = oracle.depth_iter(sp, origin, direction)
    assert [int(s["kind"]) for s in d] == [
        D_INVISIBLE, D_INVISIBLE, D_SPAN, D_ENTER_BLOCK, D_INVISIBLE, D_SPAN, D_SPAN, D_INVISIBLE, D_INVISIBLE, D_INVISIBLE,
    ]
    assert surf(d[2]) == (RED, Z3, (0, 1, 0), 1, (0, 0, 0), 1.5, (0.25, 1.0, 0.25), NY) and d[2]["exit_t_distance"] == 2.5
    assert d[3]["t_distance"] == 2.5
    assert surf(d[5]) == (YEL, Z3, (0, 2, 0), 4, (1, 0, 1), 2.5, (0.25, 2.0, 0.25), NY) and d[5]["exit_t_distance"] == 2.75
    assert surf(d[6]) == (YEL, Z3, (0, 2, 0), 4, (1, 1, 1), 2.75, (0.25, 2.25, 0.25), NY) and d[6]["exit_t_distance"] == 3.0


# surface.rs:682-709
def test_surface_iter_exit_block_at_end_of_space():
    sp = oracle.Space(scenes.one_red_cube_space())
    steps = oracle.surface_iter(sp, (-0.5, 0.5, 0.5), (1.0, 0.0, 0.0))
    assert [int(s["kind"]) for s in steps] == [ENTER_SURFACE, INVISIBLE]
    assert surf(steps[0]) == (RED, Z3, (0, 0, 0), 1, (0, 0, 0), 0.5, (0.0, 0.5, 0.5), NX)
    assert steps[1]["t_distance"] == 1.5


# surface.rs:713-749
def test_ray_misses_voxels():
    sp = oracle.Space(scenes.slab_cube_space())
    origin, direction = (-0.5, 0.75, 0.25), (1.0, 0.0, 0.0)
    steps = oracle.surface_iter(sp, origin, direction)
    assert [(int(s["kind"]), float(s["t_distance"])) for s in steps] == [(ENTER_BLOCK, 0.5), (INVISIBLE, 1.5)]
    d = oracle.depth_iter(sp, origin, direction)
    assert [int(s["kind"]) for s in d] == [D_INVISIBLE, D_ENTER_BLOCK, D_INVISIBLE]
    assert d[1]["t_distance"] == 0.5


# surface.rs:755-835
def test_depth_iter_exiting_block_volume_before_cube():
    sp = oracle.Space(scenes.half_transparent_slab_space())
    origin, direction = (0.25, -0.5, 0.25), (0.0, 1.0, 0.0)
    col = (1.0, 1.0, 0.0, 0.5)
    steps = oracle.surface_iter(sp, origin, direction)
    assert [int(s["kind"]) for s in steps] == [INVISIBLE, ENTER_BLOCK, ENTER_SURFACE, INVISIBLE, INVISIBLE, INVISIBLE]
    assert steps[0]["t_distance"] == 0.5 and steps[1]["t_distance"] == 1.5
    assert surf(steps[2]) == (col, Z3, (0, 1, 0), 2, (0, 0, 0), 1.5, (0.25, 1.0, 0.25), NY)
    assert [float(s["t_distance"]) for s in steps[3:]] == [2.0, 2.5, 3.5]
    d = oracle.depth_iter(sp, origin, direction)
    assert [int(s["kind"]) for s in d] == [D_INVISIBLE, D_INVISIBLE, D_ENTER_BLOCK, D_INVISIBLE, D_SPAN, D_INVISIBLE, D_INVISIBLE]
    assert d[2]["t_distance"] == 1.5
    assert surf(d[4]) == (col, Z3, (0, 1, 0), 2, (0, 0, 0), 1.5, (0.25, 1.0, 0.25), NY) and d[4]["exit_t_distance"] == 2.0


# surface.rs:837-855
def test_interpolation_functions():
    assert oracle.smoothstep(0.0) == 0.0 and oracle.smoothstep(0.5) == 0.5 and oracle.smoothstep(1.0) == 1.0
    for x, e in [(0.0, 1 / 8), (0.24, 1 / 8), (0.26, 3 / 8), (0.49, 3 / 8), (0.51, 5 / 8), (0.74, 5 / 8), (0.76, 7 / 8), (0.99, 7 / 8), (1.0, 7 / 8)]:
        assert oracle.coarsestep(x) == e


# accum.rs:442-496
@pytest.mark.parametrize(
    "expected,origin,direction",
    [
        (0.0, (0.25, 0.25, 0.0), (0.0, 0.0, 1.0)),
        (0.25, (0.25, 0.25, -0.25), (0.0, 0.0, 1.0)),
        (0.25 / 4.0, (0.25, 0.25, -0.25), (0.0, 0.0, 4.0)),
        (5.25 - 0.5, (0.5, 5.25, 0.5), (0.0, -1.0, 0.0)),
        (float("inf"), (0.5, 0.75, -0.5), (0.0, 0.0, 1.0)),
    ],
)
def test_depth_buf(expected, origin, direction):
    sp = oracle.Space(scenes.slab_cube_space())
    _, _, depth = oracle.trace_ray(sp, oracle.make_options(), origin, direction, include_sky=False)
    assert depth == expected


# raytracer_components.rs:269-331
def test_apply_transmittance():
    color = (1.0, 0.5, 0.0, 0.5)
    out, coeff = oracle.apply_transmittance(color, 1.0)
    assert tuple(out) == color and coeff == 1.0
    out, coeff = oracle.apply_transmittance(color, -0.125)
    assert tuple(out) == (0, 0, 0, 0) and coeff == 0.0
    opaque = (1.0, 0.5, 0.0, 1.0)
    out, coeff = oracle.apply_transmittance(opaque, -0.125)
    assert tuple(out) == opaque and coeff == 1.0
    out, coeff = oracle.apply_transmittance(color, 0.0)
    assert tuple(out) == (0, 0, 0, 0) and coeff == 0.0
    out, coeff = oracle.apply_transmittance(opaque, 0.0)
    assert tuple(out) == opaque and coeff == 1.0


# raytracer_components.rs:279-304: n layers of thickness 1/n compose back to the colour
@pytest.mark.parametrize("count", [1, 2, 8])
def test_apply_transmittance_equivalence(count):
    color = np.array((1.0, 0.5, 0.0, 0.5), np.float32)
    mod, _ = oracle.apply_transmittance(color, np.float32(1.0) / np.float32(count))
    light = np.zeros(3, np.float32)
    t = np.float32(1.0)
    for _ in range(count):
        light = light + (mod[:3] * mod[3]) * t
        t = t * (np.float32(1.0) - mod[3])
    alpha = np.float32(1.0) - t
    actual = np.array([*(light / alpha), alpha])
    assert (actual - color).sum() < 0.00001


# light/data.rs:301-354 table vs the defining formula; 362-378 round trip
def test_packed_light_lut(golden_dir):
    ref = np.load(golden_dir / "packed_light_lut.npy")
    got = oracle.packed_light_lut()
    assert got.dtype == np.float32 and (got.view(np.uint32) == ref.view(np.uint32)).all()
    for i in range(255):
        assert oracle.packed_light_scalar_in(float(ref[i])) == i
