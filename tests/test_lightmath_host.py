"""The SHADE event's light interpolation, checked on the CPU: all_is_cubes_amd/csrc/aic_lightmath.h -- the source the HIP
kernel inlines -- is compiled for the host (tests/native/lightmath_shim.cpp) and compared with the oracle's
get_interpolated_light (sr.rs:248-359) bit for bit on random surfaces: inside the space, on its outermost layer (BlockSky
samples), at its corners (NO_RAYS), for every face and LightingOption. The GPU parity tests then only have to show that the
device executes the same source the same way."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle
from all_is_cubes_amd import flat

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "lightmath_shim.cpp"
HDR = ROOT / "all_is_cubes_amd" / "csrc" / "aic_lightmath.h"


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("lightmath") / "liblightmath_shim.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-I", str(HDR.parent), "-o", str(out), str(SRC)],
                   check=True)
    lib = C.CDLL(str(out))
    lib.shim_interpolated_light.restype = C.c_uint32
    lib.shim_light_outside.restype = C.c_uint32
    return lib


def _space(rng, lo, size):
    sp = flat.FlatSpace(tuple(lo), tuple(size))
    sp.set_sky_uniform((0.9, 0.7, 0.5))
    n = int(np.prod(size))
    light = np.zeros((n, 4), np.uint8)
    light[:, 0:3] = rng.integers(0, 256, (n, 3))
    light[:, 3] = rng.choice(np.array([0, 1, 128, 255], np.uint8), n, p=[0.1, 0.1, 0.3, 0.5])  # every LightStatus
    sp.light[...] = light.reshape(tuple(size) + (4,))
    return sp


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _sky_texels(osp):
    bs = oracle.block_sky(osp).astype(np.uint32)
    return np.ascontiguousarray(bs[:, 0] | (bs[:, 1] << 8) | (bs[:, 2] << 16) | (bs[:, 3] << 24))


@pytest.mark.parametrize("lo,size", [((0, 0, 0), (6, 5, 4)), ((-7, -3, 11), (3, 4, 5)), ((-2, -2, -2), (1, 1, 1))])
def test_interpolated_light_equals_oracle(shim, lo, size):
    rng = np.random.default_rng(hash((lo, size)) & 0xffff)
    fs = _space(rng, lo, size)
    osp = oracle.Space(fs)
    light = np.ascontiguousarray(fs.light.reshape(-1, 4).astype(np.uint32))
    texels = np.ascontiguousarray(light[:, 0] | (light[:, 1] << 8) | (light[:, 2] << 16) | (light[:, 3] << 24))
    sky = _sky_texels(osp)
    lut = np.ascontiguousarray(oracle.packed_light_lut().astype(np.float32))
    lo_a, size_a = np.array(lo, np.int32), np.array(size, np.int32)
    n_checked = n_outside = 0
    for trial in range(6000):
        face = int(rng.integers(0, 7))
        mode = int(rng.choice([2, 3, 4]))
        # a cube of the space or of the layer of cubes around it, and a surface point on (or, for Within, inside) it
        cube = np.array([rng.integers(lo[a] - 1, lo[a] + size[a] + 1) for a in range(3)], np.int32)
        frac = rng.random(3)
        kind = trial % 4
        if kind == 1:   # exactly on the grid lines of the half-cube lattice
            frac = rng.choice([0.0, 0.5, 1.0, 0.25], 3)
        elif kind == 2:  # hair off them
            frac = np.clip(rng.choice([0.0, 0.5, 1.0], 3) + rng.choice([-1e-9, 1e-9, 2.0 ** -52], 3), 0.0, 1.0)
        sp = cube.astype(np.float64) + frac
        if face:  # on the face plane of the cube (voxel surfaces sit anywhere inside it: kind 3 keeps the random height)
            axis = (face - 1) % 3
            if kind != 3:
                sp[axis] = cube[axis] + (1.0 if face >= 4 else 0.0)
        ref, n_ref = oracle.interpolated_light(osp, cube, sp, face, mode)
        got = np.zeros(3, np.float32)
        n_got = shim.shim_interpolated_light(_p(texels), _p(lo_a), _p(size_a), _p(sky), _p(lut), _p(cube), _p(sp), face, mode, _p(got))
        assert got.tobytes() == ref.tobytes(), (trial, face, mode, cube, sp, got, ref)
        assert n_got == n_ref, (trial, face, mode, cube, sp, n_got, n_ref)
        n_checked += 1
        n_outside += int(n_ref > 0 and (np.any(cube <= lo_a) or np.any(cube >= lo_a + size_a - 1)))
    assert n_checked == 6000 and n_outside > 500  # the BlockSky paths were exercised


def test_interpolated_light_at_the_i32_edge(shim):
    """A space that touches i32::MIN / i32::MAX: samples without a containing cube take BlockSky::mean (sr.rs:307-311)."""
    rng = np.random.default_rng(3)
    lut = np.ascontiguousarray(oracle.packed_light_lut().astype(np.float32))
    for lo in ((-2147483648, 0, 0), (0, 2147483644, 0), (0, 0, 2147483645)):
        size = tuple(2 if lo[a] else 3 for a in range(3))
        fs = _space(rng, lo, size)
        osp = oracle.Space(fs)
        light = np.ascontiguousarray(fs.light.reshape(-1, 4).astype(np.uint32))
        texels = np.ascontiguousarray(light[:, 0] | (light[:, 1] << 8) | (light[:, 2] << 16) | (light[:, 3] << 24))
        sky = _sky_texels(osp)
        lo_a, size_a = np.array(lo, np.int32), np.array(size, np.int32)
        for trial in range(600):
            face = int(rng.integers(0, 7))
            cube = np.array([rng.integers(lo[a], lo[a] + size[a]) for a in range(3)], np.int64).astype(np.int32)
            sp = cube.astype(np.float64) + rng.choice([0.0, 0.3, 0.5, 1.0], 3)
            ref, n_ref = oracle.interpolated_light(osp, cube, sp, face, 3)
            got = np.zeros(3, np.float32)
            n_got = shim.shim_interpolated_light(_p(texels), _p(lo_a), _p(size_a), _p(sky), _p(lut), _p(cube), _p(sp), face, 3, _p(got))
            assert got.tobytes() == ref.tobytes() and n_got == n_ref, (lo, trial, face, cube, sp, got, ref, n_got, n_ref)
