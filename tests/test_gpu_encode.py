"""The device's sRGB8 encoder (threshold table + search, aic_trace.hip srgb8_channel) must equal
the reference formula `Rgba::to_srgb8` (color.rs:669-676, 1038-1054) for every input, not just
within a tolerance: checked through the C ABI by rendering flat emissive colours."""
import numpy as np
import pytest

import oracle
from all_is_cubes_amd import abi, flat

pytestmark = pytest.mark.gpu


def test_srgb_encode_matches_reference_formula_densely():
    # A 64x64 image of a 64x64x1 wall of R1 emissive blocks (alpha 1, black, emission = value):
    # every pixel shows exactly one block's emission, so the encoder sees 4096 chosen values.
    rng = np.random.default_rng(7)
    vals = np.concatenate([
        np.linspace(0.0, 1.2, 1500), rng.uniform(0, 0.01, 500), rng.uniform(0.0, 1.0, 1500),
        np.array([0.0031308, 0.00313081, 0.0031307, 1.0, 0.99999994, 1.0000001, 5.0, 1e-8]),
    ]).astype(np.float32)
    # add values straddling encoder thresholds: for each level k, the float just below/above where enc changes
    probe = np.linspace(0, 1, 70000, dtype=np.float32)
    enc = np.array([oracle.to_srgb8((v, 0, 0, 1))[0] for v in probe[::70]])
    vals = np.concatenate([vals, probe[::70][np.nonzero(np.diff(enc))[0]]])[:4096]
    vals = np.pad(vals, (0, 4096 - len(vals)))
    n = 64
    sp = flat.FlatSpace((0, 0, 0), (n, n, 1))
    sp.set_sky_uniform((0.0, 0.0, 0.0))
    for i, v in enumerate(vals):
        idx = sp.add_block(flat.atom((0.0, 0.0, 0.0, 1.0), emission=(float(v), float(v) * 0.5, float(np.float32(1.0) - min(v, 1.0)))))
        sp.set((i % n, i // n, 0), idx)
    w = h = 4 * n
    eye = (n / 2, n / 2, n / 2 + 1.0)
    _, _, inv = oracle.camera_matrices(90.0, 200.0, 1.0, (0, 0, 0, 1), eye)
    with abi.Context(0) as ctx:
        ctx.upload_space(abi.LAYER_WORLD, sp)
        ctx.set_options(abi.LAYER_WORLD, abi.unaltered_colors())
        got = ctx.render(ctx.make_frame(w, h, world_inv=inv))["rgba8"]
    ref = oracle.render(oracle.Space(sp), oracle.unaltered_colors(), oracle.make_camera(inv, w, h))["rgba8"]
    assert (got == ref).all()
    assert len(np.unique(ref[..., 0])) > 200  # the ramp really exercised the encoder


def _host_powf(x, y):
    """f32::powf as the reference gets it: the C library's powf (Rust's std calls it)."""
    import ctypes
    import ctypes.util

    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return np.array([libm.powf(float(a), float(b)) for a, b in zip(x, y)], np.float32)


def test_device_powf_equals_libm_powf():
    """apply_transmittance's powf (raytracer_components.rs:215-258) on the device -- glibc's table
    algorithm restated (aic_trace.hip powf_table) -- against the host libm, bit for bit."""
    rng = np.random.default_rng(11)
    n = 400_000
    x = np.concatenate([
        rng.uniform(0.0, 1.0, n), 1.0 - 10.0 ** rng.uniform(-7.5, -0.3, n // 4), 10.0 ** rng.uniform(-38, 0, n // 4),
        np.array([0.5, 0.25, 0.9999999, 1.1754944e-38, 0.99999994, 0.70710677, 0.70710683, 0.7, 0.3]),
    ]).astype(np.float32)
    y = np.concatenate([
        10.0 ** rng.uniform(-6, 2.7, n), rng.uniform(0.0, 4.0, n // 4), 10.0 ** rng.uniform(-3, 3, n // 4),
        np.array([1.0, 2.0, 1e-30, 3.4e38, 1e-45, 0.5, 0.5, 150.0, 126.5]),
    ]).astype(np.float32)
    keep = (x > 0) & (x < 1) & (y > 0)
    x, y = x[keep], y[keep]
    with abi.Context(0) as ctx:
        got = ctx.probe_powf(x, y)
        # outside the table's domain the kernel keeps the f64 evaluation: x >= 1, subnormal x, y = inf
        xs = np.array([1.0, 1.5, 1e-40, 0.5, 2.0], np.float32)
        ys = np.array([3.0, 2.5, 2.0, np.inf, 0.5], np.float32)
        assert (ctx.probe_powf(xs, ys).view(np.uint32) == _host_powf(xs, ys).view(np.uint32)).all()
    want = _host_powf(x, y)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, f"{len(bad)} of {len(x)} differ, e.g. x={x[bad[:3]]} y={y[bad[:3]]} got={got[bad[:3]]} want={want[bad[:3]]}"
