"""Randomised differential test: the HIP path through the C ABI against the CPU oracle on seeded
random scenes, cameras and options -- mixed block resolutions, partial voxel volumes, random light
(all statuses), octant skies, backdrops, eyes inside and outside the space, axis-parallel views,
narrow and wide fields of view. Same bar as tests/test_gpu_parity.py: per-pixel step counts, first
hit (cube, voxel, face, block), f64 distance bit for bit; RGBA8 within one level."""
import os

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat
from tests import scenes
from tests.test_gpu_parity import assert_parity, to_abi_options

pytestmark = pytest.mark.gpu


def _random_space(rng):
    n = int(rng.integers(4, 14))
    size = (n, int(rng.integers(3, 12)), int(rng.integers(4, 14)))
    lo = tuple(int(v) for v in rng.integers(-20, 20, 3))
    sp = flat.FlatSpace(lo, size)
    if rng.random() < 0.5:
        sp.set_sky_uniform(tuple(float(v) for v in rng.uniform(0, 1.2, 3)))
    else:
        sp.set_sky_octants(rng.uniform(0, 1.0, (8, 3)).astype(np.float32))
    ids = [sp.add_block(flat.air())]
    for _ in range(int(rng.integers(1, 4))):
        c = rng.uniform(0, 1, 3)
        ids.append(sp.add_block(flat.atom((float(c[0]), float(c[1]), float(c[2]), float(rng.choice([1.0, 1.0, 0.6, 0.25]))),
                                          emission=tuple(float(v) for v in rng.uniform(0, 0.3, 3) * (rng.random() < 0.3)))))
    for res in rng.choice([2, 4, 8, 16, 32], int(rng.integers(1, 4))):
        res = int(res)
        for b in scenes.synthetic_blocks(res, 2, seed=int(rng.integers(1, 1 << 30)), translucent=bool(rng.random() < 0.7)):
            if rng.random() < 0.4 and res >= 4:  # store only part of the volume (the rest reads as AIR)
                a = [int(rng.integers(0, res // 2)) for _ in range(3)]
                e = [int(rng.integers(a[k] + 1, res + 1)) for k in range(3)]
                b = flat.voxel_block(res, np.ascontiguousarray(b.voxels[a[0]:e[0], a[1]:e[1], a[2]:e[2]]), b.palette, vlo=a)
            ids.append(sp.add_block(b))
    fill = rng.uniform(0.1, 0.6)
    grid = np.where(rng.random(size) < fill, rng.integers(1, len(ids), size), 0)
    sp.block_index[...] = np.array(ids, np.uint16)[grid]
    sp.light[..., 0:3] = rng.integers(0, 256, size + (3,))
    sp.light[..., 3] = rng.choice([0, 1, 128, 255, 255, 255], size)
    return sp


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AIC_FUZZ_N", "24"))))
def test_random_scene_camera_options(seed):
    rng = np.random.default_rng(1000 + seed)
    sp = _random_space(rng)
    opt = oracle.make_options(
        fog=int(rng.integers(0, 4)), transparency=int(rng.integers(0, 3)), threshold=float(rng.uniform(0.1, 0.9)),
        lighting=int(rng.integers(0, 5)), antialiasing=int(rng.choice([0, 0, 2])), debug_pixel_cost=bool(rng.random() < 0.1),
        tone_mapping=int(rng.integers(0, 2)), maximum_intensity=float(rng.choice([np.inf, 1.0, 2.5])),
        view_distance=float(rng.choice([6.0, 30.0, 200.0])))
    if seed % 6 == 5:  # LightingOption::Bounce (drawn after the other options, so every other seed keeps its scene and options)
        opt.lighting, opt.bounce_samples = 5, 1 + seed % 3
    lo, hi = np.array(sp.lo, float), np.array(sp.hi, float)
    mode = seed % 4
    if mode == 0:    # outside, looking at the centre
        eye = (lo + hi) / 2 + rng.normal(0, 1, 3) * (hi - lo) * 1.2
        q = oracle.look_at_y_up(tuple(eye), tuple((lo + hi) / 2))
    elif mode == 1:  # inside the space, any orientation
        eye = rng.uniform(lo, hi)
        v = rng.normal(0, 1, 4)
        q = v / np.linalg.norm(v)
    elif mode == 2:  # exactly along an axis, on cube boundaries
        eye = np.floor(rng.uniform(lo, hi)) + np.array([0.0, 0.5, 0.0])
        q = [(0, 0, 0, 1), (0, 1, 0, 0), (0, np.sqrt(0.5), 0, np.sqrt(0.5)), (np.sqrt(0.5), 0, 0, np.sqrt(0.5))][int(rng.integers(0, 4))]
    else:            # far away with a narrow field of view
        eye = (lo + hi) / 2 + np.array([3.0, 2.0, 5.0]) * float(np.max(hi - lo)) * 2
        q = oracle.look_at_y_up(tuple(eye), tuple((lo + hi) / 2))
    fov = float(rng.choice([20.0, 60.0, 90.0, 120.0]))
    w, h = int(rng.integers(33, 97)), int(rng.integers(17, 65))
    _, _, inv = oracle.camera_matrices(fov, opt.view_distance, w / h, tuple(float(v) for v in q), tuple(float(v) for v in eye))
    backdrop = (0, 0, 0, 0) if rng.random() < 0.6 else tuple(float(v) for v in rng.uniform(0, 1, 4))
    with abi.Context(0) as ctx:
        ctx.upload_space(abi.LAYER_WORLD, sp)
        ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv, backdrop=backdrop), want_aux=True)
        # the production kernel variants (no per-pixel records; built for 4 waves per SIMD): the plain one an image this small gets, and the one that
        # exchanges lanes between the waves of a workgroup (round 5; asked for through aic_frame_desc.tuning) -- parked rays, spare columns,
        # claims lost to other waves and the drain at the frame's end all happen at this size too
        fast = ctx.render(ctx.make_frame(w, h, world_inv=inv, backdrop=backdrop))
        cost = None
        xt = abi.tuning(variant=abi.VARIANT_EXCHANGING)
        exchanged = ctx.render(ctx.make_frame(w, h, world_inv=inv, backdrop=backdrop, tuning=xt))
        assert fast["info"].variant == abi.VARIANT_PLAIN and exchanged["info"].variant == (abi.VARIANT_PLAIN if opt.lighting == 5 else abi.VARIANT_EXCHANGING)
        if opt.antialiasing == 0 and opt.lighting != 5:  # (a mean of four samples / secondary rays' steps do not come back from one channel)
            was = opt.debug_pixel_cost
            opt.debug_pixel_cost = 1
            ctx.set_options(abi.LAYER_WORLD, to_abi_options(opt))
            cost = ctx.render(ctx.make_frame(w, h, world_inv=inv, backdrop=backdrop, flags=abi.FRAME_OUT_LINEAR, tuning=xt))
            opt.debug_pixel_cost = was
    assert (fast["rgba8"] == got["rgba8"]).all() and fast["info"].cubes_traced == got["info"].cubes_traced
    assert (exchanged["rgba8"] == got["rgba8"]).all() and exchanged["info"].cubes_traced == got["info"].cubes_traced
    if cost is not None:
        # the production variants' PER-PIXEL step counts (the opaque shortcut works the reference's remaining counted steps out instead of taking
        # them): with debug_pixel_cost the pixel is rgb(0.02 n, 0.002 n, ..) and the linear float output hands it over unrounded
        counts = np.rint(cost["rgba8"][..., 1].astype(np.float64) / float(np.float32(0.002))).astype(np.int64)
        assert (counts == got["aux"]["cubes_traced"]).all(), "per-pixel step counts of the production (exchanging) variant"
    ref = oracle.render(oracle.Space(sp), opt, oracle.make_camera(inv, w, h), backdrop=backdrop, want_aux=True)
    assert_parity(got, ref)
