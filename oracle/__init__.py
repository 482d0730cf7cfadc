"""ctypes binding of the CPU oracle (oracle/libaic_oracle.so).

TEST INFRASTRUCTURE ONLY. Importable from tests/, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg -- never from the product package `all_is_cubes_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libaic_oracle.so"


def build(force: bool = False) -> Path:
    srcs = (_HERE / "aic_oracle.cpp", _HERE / "aic_light.inc", _HERE / "aic_ortho.inc", _HERE / "aic_oracle.h")
    srcs = srcs + (_HERE / "Makefile",)
    stale = (not _LIB_PATH.exists()) or (not (_HERE / "libaic_oracle_o2.so").exists()) or any(p.stat().st_mtime > _LIB_PATH.stat().st_mtime for p in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return _LIB_PATH


class OrcBlock(C.Structure):
    _fields_ = [
        ("resolution", C.c_int32),
        ("vlo", C.c_int32 * 3),
        ("vsize", C.c_int32 * 3),
        ("vox_off", C.c_uint32),
        ("pal_off", C.c_uint32),
        ("pal_len", C.c_uint32),
        ("is_one", C.c_uint32),
        ("name_char", C.c_int32),
    ]


class OrcSpace(C.Structure):
    _fields_ = [
        ("lo", C.c_int32 * 3),
        ("size", C.c_int32 * 3),
        ("block_index", C.c_void_p),
        ("light", C.c_void_p),
        ("always_invisible", C.c_void_p),
        ("n_blocks", C.c_uint32),
        ("blocks", C.c_void_p),
        ("voxels", C.c_void_p),
        ("palette", C.c_void_p),
        ("sky_kind", C.c_int32),
        ("sky", (C.c_float * 3) * 8),
    ]


class OrcOptions(C.Structure):
    _fields_ = [
        ("fog", C.c_int32),
        ("transparency", C.c_int32),
        ("threshold", C.c_float),
        ("lighting", C.c_int32),
        ("bounce_samples", C.c_int32),
        ("antialiasing", C.c_int32),
        ("debug_pixel_cost", C.c_int32),
        ("tone_mapping", C.c_int32),
        ("maximum_intensity", C.c_float),
        ("exposure", C.c_float),
        ("view_distance", C.c_double),
    ]


class OrcCamera(C.Structure):
    _fields_ = [("inverse_projection_view", C.c_double * 16), ("width", C.c_uint32), ("height", C.c_uint32)]


RC_STEP_DTYPE = np.dtype(
    [("cube", "<i4", (3,)), ("face", "<i4"), ("t_distance", "<f8"), ("t_max", "<f8", (3,)), ("intersection_point", "<f8", (3,))],
    align=True,
)
TRACE_STEP_DTYPE = np.dtype(
    [
        ("kind", "<i4"),
        ("block_index", "<i4"),
        ("t_distance", "<f8"),
        ("exit_t_distance", "<f8"),
        ("color", "<f4", (4,)),
        ("emission", "<f4", (3,)),
        ("cube", "<i4", (3,)),
        ("resolution", "<i4"),
        ("voxel", "<i4", (3,)),
        ("intersection_point", "<f8", (3,)),
        ("normal", "<i4"),
    ],
    align=True,
)
PIXEL_AUX_DTYPE = np.dtype(
    [
        ("hit", "<i4"),
        ("cube", "<i4", (3,)),
        ("voxel", "<i4", (3,)),
        ("resolution", "<i4"),
        ("face", "<i4"),
        ("block_index", "<i4"),
        ("cubes_traced", "<u4"),
        ("t_distance", "<f8"),
    ],
    align=True,
)
INFO_DTYPE = np.dtype([("cubes_traced", "<u8"), ("n_outer", "<u8"), ("n_inner", "<u8"), ("n_hits", "<u8"), ("n_light", "<u8")])

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_scale_to_integer_step.restype = C.c_double
        _lib.orc_scale_to_integer_step.argtypes = [C.c_double, C.c_double]
        _lib.orc_trace_ray.restype = C.c_uint64
        _lib.orc_packed_light_scalar_out.restype = C.c_float
        _lib.orc_packed_light_scalar_out.argtypes = [C.c_uint8]
        _lib.orc_packed_light_scalar_in.restype = C.c_uint8
        _lib.orc_packed_light_scalar_in.argtypes = [C.c_float]
        _lib.orc_smoothstep.restype = C.c_double
        _lib.orc_smoothstep.argtypes = [C.c_double]
        _lib.orc_coarsestep.restype = C.c_double
        _lib.orc_coarsestep.argtypes = [C.c_double]
        assert C.sizeof(OrcBlock) == 48
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _d3(v) -> np.ndarray:
    return np.ascontiguousarray(v, dtype=np.float64).reshape(3)


def _i3(v) -> np.ndarray:
    return np.ascontiguousarray(v, dtype=np.int32).reshape(3)


class Space:
    """Holds a packed flat space alive and exposes it as `orc_space`."""

    def __init__(self, flat) -> None:
        p = flat.pack() if hasattr(flat, "pack") else flat
        self.packed = p
        blocks = p.blocks.copy()
        # orc_block.is_one takes bit0 of the shared `flags` field
        blocks["flags"] = blocks["flags"] & 1
        self._blocks = blocks
        s = OrcSpace()
        s.lo[:] = [int(v) for v in p.lo]
        s.size[:] = [int(v) for v in p.size]
        s.block_index = _p(p.block_index)
        s.light = _p(p.light)
        s.always_invisible = _p(p.always_invisible)
        s.n_blocks = len(blocks)
        s.blocks = _p(blocks) if len(blocks) else None
        s.voxels = _p(p.voxels) if p.voxels.size else None
        s.palette = _p(p.palette) if p.palette.size else None
        s.sky_kind = p.sky_kind
        for i in range(8):
            for j in range(3):
                s.sky[i][j] = float(p.sky[i, j])
        self.c = s


def make_options(
    fog: int = 1,
    transparency: int = 1,
    threshold: float = 0.5,
    lighting: int = 3,
    bounce_samples: int = 0,
    antialiasing: int = 0,
    debug_pixel_cost: bool = False,
    tone_mapping: int = 0,
    maximum_intensity: float = float("inf"),
    exposure: float = 1.0,
    view_distance: float = 200.0,
) -> OrcOptions:
    """Defaults = GraphicsOptions::default() (graphics_options.rs:256-280)."""
    o = OrcOptions()
    o.fog, o.transparency, o.threshold = fog, transparency, threshold
    o.lighting, o.bounce_samples, o.antialiasing = lighting, bounce_samples, antialiasing
    o.debug_pixel_cost, o.tone_mapping = int(debug_pixel_cost), tone_mapping
    o.maximum_intensity, o.exposure, o.view_distance = maximum_intensity, exposure, view_distance
    return o


def unaltered_colors(**kw) -> OrcOptions:
    """GraphicsOptions::UNALTERED_COLORS (graphics_options.rs:168-190)."""
    base = dict(fog=0, lighting=0, transparency=1)
    base.update(kw)
    return make_options(**base)


def make_camera(inverse_projection_view, width: int, height: int) -> OrcCamera:
    c = OrcCamera()
    m = np.ascontiguousarray(inverse_projection_view, dtype=np.float64).reshape(16)
    c.inverse_projection_view[:] = [float(v) for v in m]
    c.width, c.height = width, height
    return c


# -- thin call wrappers ------------------------------------------------------------------


def scale_to_integer_step(s: float, ds: float) -> float:
    return lib().orc_scale_to_integer_step(s, ds)


def raycast(origin, direction, bounds=None, include_exit=True, max_steps=64):
    """Returns (steps structured array, ended flag)."""
    out = np.zeros(max_steps, RC_STEP_DTYPE)
    ended = C.c_int32(0)
    o, d = _d3(origin), _d3(direction)
    if bounds is None:
        lo = hi = _i3([0, 0, 0])
        use = 0
    else:
        lo, hi = _i3(bounds[0]), _i3(bounds[1])
        use = 1
    n = lib().orc_raycast(C.c_void_p(_p(o)), C.c_void_p(_p(d)), use, C.c_void_p(_p(lo)), C.c_void_p(_p(hi)),
                          int(include_exit), max_steps, C.c_void_p(_p(out)), C.byref(ended))
    return out[:n], bool(ended.value)


def recursive_raycast(origin, direction, outer_index, resolution, bounds, max_steps=64):
    out = np.zeros(max_steps, RC_STEP_DTYPE)
    ended = C.c_int32(0)
    o, d = _d3(origin), _d3(direction)
    lo, hi = _i3(bounds[0]), _i3(bounds[1])
    sub = np.zeros(3)
    n = lib().orc_recursive_raycast(C.c_void_p(_p(o)), C.c_void_p(_p(d)), outer_index, resolution, C.c_void_p(_p(lo)),
                                    C.c_void_p(_p(hi)), C.c_void_p(_p(sub)), max_steps, C.c_void_p(_p(out)), C.byref(ended))
    assert n >= 0
    return out[:n], bool(ended.value), sub


def surface_iter(space: Space, origin, direction, max_steps=64):
    out = np.zeros(max_steps, TRACE_STEP_DTYPE)
    o, d = _d3(origin), _d3(direction)
    n = lib().orc_surface_iter(C.byref(space.c), C.c_void_p(_p(o)), C.c_void_p(_p(d)), max_steps, C.c_void_p(_p(out)))
    return out[:n]


def depth_iter(space: Space, origin, direction, max_steps=64):
    out = np.zeros(max_steps, TRACE_STEP_DTYPE)
    o, d = _d3(origin), _d3(direction)
    n = lib().orc_depth_iter(C.byref(space.c), C.c_void_p(_p(o)), C.c_void_p(_p(d)), max_steps, C.c_void_p(_p(out)))
    return out[:n]


def trace_ray(space: Space, options: OrcOptions, origin, direction, include_sky=True):
    """Returns (cubes_traced, ColorBuf [light rgb, transmittance], DepthBuf depth)."""
    o, d = _d3(origin), _d3(direction)
    lt = np.zeros(4, np.float32)
    depth = C.c_double(0)
    n = lib().orc_trace_ray(C.byref(space.c), C.byref(options), C.c_void_p(_p(o)), C.c_void_p(_p(d)), int(include_sky),
                            C.c_void_p(_p(lt)), C.byref(depth))
    return int(n), lt, depth.value


def lib_o2() -> C.CDLL:
    """The same oracle built at -O2 (rounds 1-5's flags): only bench.py's cpu_baseline leg loads it, to report both."""
    lib()  # (builds both)
    return C.CDLL(str(_HERE / "libaic_oracle_o2.so"))


def render(world: Optional[Space], world_opt: OrcOptions, world_cam: OrcCamera, ui: Optional[Space] = None,
           ui_opt: Optional[OrcOptions] = None, ui_cam: Optional[OrcCamera] = None, backdrop=(0, 0, 0, 0),
           rows=None, threads: int = 0, want_linear=False, want_aux=False, use_lib=None):
    """RtRenderer::draw_rgba equivalent. Returns dict(rgba8, linear, aux, info)."""
    w, h = world_cam.width, world_cam.height
    r0, r1 = (0, h) if rows is None else rows
    rgba8 = np.zeros((h, w, 4), np.uint8)
    linear = np.zeros((h, w, 4), np.float32) if want_linear else None
    aux = np.zeros((h, w), PIXEL_AUX_DTYPE) if want_aux else None
    info = np.zeros(1, INFO_DTYPE)
    bd = np.ascontiguousarray(backdrop, dtype=np.float32)
    if threads <= 0:
        threads = os.cpu_count() or 1
    rc = (use_lib if use_lib is not None else lib()).orc_render(
        C.byref(world.c) if world is not None else None, C.byref(world_opt), C.byref(world_cam),
        C.byref(ui.c) if ui is not None else None,
        C.byref(ui_opt) if ui_opt is not None else None,
        C.byref(ui_cam) if ui_cam is not None else None,
        C.c_void_p(_p(bd)), r0, r1, threads,
        C.c_void_p(_p(rgba8)),
        C.c_void_p(_p(linear)) if linear is not None else None,
        C.c_void_p(_p(aux)) if aux is not None else None,
        C.c_void_p(_p(info)),
    )
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return {"rgba8": rgba8, "linear": linear, "aux": aux, "info": info[0], "threads": threads}


def render_text(space: Space, options: OrcOptions, cam: OrcCamera) -> str:
    out = np.zeros((cam.height, cam.width), np.int32)
    lib().orc_render_text(C.byref(space.c), C.byref(options), C.byref(cam), C.c_void_p(_p(out)))
    lines = []
    for row in out:
        lines.append("".join("." if v == -2 else " " if v == -1 else chr(v) for v in row))
    return "\n".join(lines) + "\n"


def look_at_y_up(eye, target) -> np.ndarray:
    q = np.zeros(4)
    e, t = _d3(eye), _d3(target)
    lib().orc_look_at_y_up(C.c_void_p(_p(e)), C.c_void_p(_p(t)), C.c_void_p(_p(q)))
    return q


def eye_for_look_at(lo, hi, direction) -> np.ndarray:
    out = np.zeros(3)
    l, h, d = _i3(lo), _i3(hi), _d3(direction)
    lib().orc_eye_for_look_at(C.c_void_p(_p(l)), C.c_void_p(_p(h)), C.c_void_p(_p(d)), C.c_void_p(_p(out)))
    return out


def camera_matrices(fov_y: float, view_distance: float, aspect: float, quat_ijkr=(0, 0, 0, 1), translation=(0, 0, 0)):
    """Returns (projection, world_to_eye, inverse_projection_view) as (4,4) arrays in euclid
    m11..m44 order, or raises if not invertible."""
    p, w, inv = np.zeros(16), np.zeros(16), np.zeros(16)
    q = np.ascontiguousarray(quat_ijkr, dtype=np.float64)
    t = _d3(translation)
    ok = lib().orc_camera_matrices(C.c_double(fov_y), C.c_double(view_distance), C.c_double(aspect), C.c_void_p(_p(q)),
                                   C.c_void_p(_p(t)), C.c_void_p(_p(p)), C.c_void_p(_p(w)), C.c_void_p(_p(inv)))
    if not ok:
        raise ValueError("projection and view matrix was not invertible")
    return p.reshape(4, 4), w.reshape(4, 4), inv.reshape(4, 4)


def project_ndc_into_world(inv, x: float, y: float):
    m = np.ascontiguousarray(inv, dtype=np.float64).reshape(16)
    o, d = np.zeros(3), np.zeros(3)
    lib().orc_project_ndc_into_world(C.c_void_p(_p(m)), C.c_double(x), C.c_double(y), C.c_void_p(_p(o)), C.c_void_p(_p(d)))
    return o, d


def unproject(inv, ndc) -> np.ndarray:
    m = np.ascontiguousarray(inv, dtype=np.float64).reshape(16)
    n, o = _d3(ndc), np.zeros(3)
    lib().orc_unproject(C.c_void_p(_p(m)), C.c_void_p(_p(n)), C.c_void_p(_p(o)))
    return o


def apply_transmittance(color, thickness: float):
    c = np.ascontiguousarray(color, dtype=np.float32)
    out = np.zeros(4, np.float32)
    coeff = C.c_float(0)
    lib().orc_apply_transmittance(C.c_void_p(_p(c)), C.c_float(thickness), C.c_void_p(_p(out)), C.byref(coeff))
    return out, coeff.value


def to_srgb8(rgba) -> np.ndarray:
    c = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.zeros(4, np.uint8)
    lib().orc_to_srgb8(C.c_void_p(_p(c)), C.c_void_p(_p(out)))
    return out


def packed_light_lut() -> np.ndarray:
    return np.array([lib().orc_packed_light_scalar_out(v) for v in range(256)], dtype=np.float32)


def packed_light_scalar_in(v: float) -> int:
    return int(lib().orc_packed_light_scalar_in(C.c_float(v)))


def block_sky(space: Space) -> np.ndarray:
    out = np.zeros((7, 4), np.uint8)
    lib().orc_block_sky(C.byref(space.c), C.c_void_p(_p(out)))
    return out


def interpolated_light(space: Space, cube, surface_point, face: int, lighting: int = 3):
    """get_interpolated_light (sr.rs:248-359) of one surface -> (rgb f32[3], number of get_packed_light calls)."""
    out = np.zeros(3, np.float32)
    f = lib().orc_interpolated_light
    f.restype = C.c_uint32
    n = f(C.byref(space.c), C.c_void_p(_p(_i3(cube))), C.c_void_p(_p(_d3(surface_point))), C.c_int32(face), C.c_int32(lighting), C.c_void_p(_p(out)))
    return out, int(n)


def smoothstep(x: float) -> float:
    return lib().orc_smoothstep(x)


def coarsestep(x: float) -> float:
    return lib().orc_coarsestep(x)


# ---- the light updater (aic_light.inc) -------------------------------------------

PRIORITY_NEWLY_VISIBLE, PRIORITY_UNINIT, PRIORITY_ESTIMATED = 250, 210, 200  # space/light/queue.rs:30-43


def compute_derived(space: Space) -> dict:
    """block/eval/derived.rs compute_derived per block of the space's block table."""
    n = len(space._blocks)
    out = np.zeros((n, 32), np.float32)
    opaque = np.zeros((n, 6), np.uint8)
    lib().orc_compute_derived.restype = None
    lib().orc_compute_derived(C.byref(space.c), C.c_void_p(_p(out)), C.c_void_p(_p(opaque)))
    return {
        "color": out[:, 0:4].copy(),
        "face_colors": out[:, 4:28].reshape(n, 6, 4).copy(),
        "emission": out[:, 28:31].copy(),
        "visible": out[:, 31] != 0,
        "opaque": opaque != 0,
    }


def light_chart():
    """(weights [n,6] f32, children [n,6] u32) of the light propagation chart; node 0 is the root."""
    f = lib().orc_light_chart
    f.restype = C.c_uint32
    n = int(f(None, None))
    w = np.zeros((n, 6), np.float32)
    c = np.zeros((n, 6), np.uint32)
    f(C.c_void_p(_p(w)), C.c_void_p(_p(c)))
    return w, c


def compute_light(space: Space, cube, maximum_distance: int = 30):
    f = lib().orc_compute_light
    f.restype = C.c_uint64
    cb = _i3(cube)
    out = np.zeros(4, np.uint8)
    cost = int(f(C.byref(space.c), C.c_int32(maximum_distance), C.c_void_p(_p(cb)), C.c_void_p(_p(out))))
    return out, cost


def evaluate_light(flat_space, maximum_distance: int = 30, fast: bool = True, epsilon: int = 1, batch: int = 32,
                   queue=None, max_updates: int = 1 << 62, hb_width: int = 16, threads: int = 1):
    """`Mutation::fast_evaluate_light()` (if `fast`) + `evaluate_light(epsilon)` on a flat space with
    `LightPhysics::Rays { maximum_distance }`. Without `fast`, starts from `flat_space.light` and `queue`
    (a list of (cube, priority)); `queue=None` enqueues every Uninitialized cube. Writes the result into
    `flat_space.light` and returns the number of updates. `hb_width`: order of equal-priority updates -- 16 / 8 =
    the reference's hashbrown table order with that Group::WIDTH, 0 = first-in-first-out (aic_light.inc)."""
    sp = Space(flat_space)
    light = np.ascontiguousarray(flat_space.light, dtype=np.uint8).copy()
    lib().orc_set_light_threads(C.c_int32(threads))  # a batch's compute_light calls on that many threads (results unchanged)
    f = lib().orc_evaluate_light
    f.restype = C.c_uint64
    if queue is None:
        nq, qc, qp = -1, np.zeros(3, np.int32), np.zeros(1, np.int32)
    else:
        nq = len(queue)
        qc = np.ascontiguousarray([c for c, _ in queue], dtype=np.int32).reshape(-1, 3) if nq else np.zeros((1, 3), np.int32)
        qp = np.ascontiguousarray([p for _, p in queue], dtype=np.int32) if nq else np.zeros(1, np.int32)
    n = int(f(C.byref(sp.c), C.c_int32(maximum_distance), C.c_int32(1 if fast else 0), C.c_int32(epsilon), C.c_int32(batch),
              C.c_uint64(max_updates), C.c_void_p(_p(light)), C.c_int32(nq), C.c_void_p(_p(qc)), C.c_void_p(_p(qp)), C.c_int32(hb_width)))
    flat_space.light = light.reshape(np.asarray(flat_space.light).shape)
    return n


# ---- axis-aligned rays and the orthographic renderer (aic_ortho.inc) ----------

def aa_raycast(origin, direction: int, bounds=None, include_exit=True, max_steps=64, sub_origin=None, zoom=None):
    """`AaRay::new(origin, direction)[.zoom_in(cube, resolution)].cast()[.within(bounds, include_exit)]`: returns
    (steps, ended, equivalent Ray as (origin, direction)). `zoom` = (cube, resolution)."""
    out = np.zeros(max_steps, RC_STEP_DTYPE)
    ended = C.c_int32(0)
    o = _i3(origin)
    lo, hi = (_i3(bounds[0]), _i3(bounds[1])) if bounds is not None else (_i3((0, 0, 0)), _i3((0, 0, 0)))
    sub = None if sub_origin is None else np.ascontiguousarray(sub_origin, np.float32)
    zc = _i3(zoom[0]) if zoom else _i3((0, 0, 0))
    ray = np.zeros(6)
    f = lib().orc_aa_raycast
    f.restype = C.c_int32
    n = f(C.c_void_p(_p(o)), C.c_int32(direction), C.c_void_p(_p(sub)) if sub is not None else None, C.c_int32(zoom[1] if zoom else 0),
          C.c_void_p(_p(zc)), C.c_int32(1 if bounds is not None else 0), C.c_void_p(_p(lo)), C.c_void_p(_p(hi)), C.c_int32(1 if include_exit else 0),
          C.c_int32(max_steps), C.c_void_p(_p(out)), C.byref(ended), C.c_void_p(_p(ray)))
    return out[:n], bool(ended.value), (ray[:3].copy(), ray[3:].copy())


def ortho_views(lo, size, resolution: int = 32):
    """MultiOrthoCamera::new(resolution, bounds): (image (w, h), rect [5,4], transform [5,4,4], direction [5,3])."""
    l, s = _i3(lo), _i3(size)
    w, h = C.c_uint32(0), C.c_uint32(0)
    lib().orc_ortho_image_size(C.c_void_p(_p(l)), C.c_void_p(_p(s)), C.c_int32(resolution), C.byref(w), C.byref(h))
    rect = np.zeros((5, 4), np.uint32)
    tr = np.zeros((5, 16), np.float64)
    di = np.zeros((5, 3), np.float64)
    lib().orc_ortho_views(C.c_void_p(_p(l)), C.c_void_p(_p(s)), C.c_int32(resolution), C.c_void_p(_p(rect)), C.c_void_p(_p(tr)), C.c_void_p(_p(di)))
    return (int(w.value), int(h.value)), rect, tr.reshape(5, 4, 4), di


def render_orthographic(space: Space, resolution: int = 32):
    """raytracer::ortho::render_orthographic: returns dict(rgba8 [h,w,4], cubes_traced)."""
    (w, h), _, _, _ = ortho_views(space.packed.lo, space.packed.size, resolution)
    out = np.zeros((h, w, 4), np.uint8)
    info = np.zeros(1, INFO_DTYPE)
    lib().orc_render_orthographic(C.byref(space.c), C.c_int32(resolution), C.c_void_p(_p(out)), C.c_void_p(_p(info)))
    return {"rgba8": out, "cubes_traced": int(info[0]["cubes_traced"])}
