"""ctypes binding of the C ABI in include/aic_hip.h (libaic_hip.so).

This is plumbing for the Python-side tests and bench: it adds nothing to the boundary.
There is NO fallback: if the HIP library is missing or no MI355X is usable the calls raise.
"""
from __future__ import annotations

import ctypes as C
import math
from pathlib import Path
from typing import Optional

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libaic_hip.so"

AIC_OK = 0
ERR_NAMES = {1: "AIC_ERR_INVALID", 2: "AIC_ERR_NO_DEVICE", 3: "AIC_ERR_OOM", 4: "AIC_ERR_DEVICE", 5: "AIC_ERR_UNSUPPORTED"}
LAYER_WORLD, LAYER_UI = 0, 1
MAX_IN_FLIGHT = 32  # AIC_MAX_IN_FLIGHT
FLAW_UNSUPPORTED, FLAW_NO_BLOOM = 1, 2
FRAME_COUNTERS, FRAME_AUX, FRAME_PIXEL_CENTERS, FRAME_OUT_LINEAR, FRAME_OUT_COLORBUF, FRAME_NO_FEEDBACK = 1, 2, 4, 8, 16, 32
# aic_frame_desc.tuning / aic_frame_info.variant (include/aic_hip.h)
TUNE_QUEUES_SHIFT, TUNE_SUPER_SHIFT, TUNE_VARIANT_SHIFT = 0, 4, 9
VARIANT_AUTO, VARIANT_PLAIN, VARIANT_EXCHANGING, VARIANT_RECORDING = 0, 1, 2, 3
LIGHT_HOOK_SESSION, LIGHT_HOOK_POOL_SHIFT = 1, 8


def tuning(queues=None, super_shift=None, variant=None) -> int:
    """aic_frame_desc.tuning: the number of tile queues (1..8), the super-block edge in macro tiles (log2) and the production variant of the
    trace kernel (VARIANT_*); None leaves the library's choice. Any value gives the same image."""
    t = 0
    if queues is not None:
        t |= (int(queues) & 15) << TUNE_QUEUES_SHIFT
    if super_shift is not None:
        t |= ((int(super_shift) + 1) & 31) << TUNE_SUPER_SHIFT
    if variant is not None:
        t |= (int(variant) & 3) << TUNE_VARIANT_SHIFT
    return t

# every symbol include/aic_hip.h declares
ABI_SYMBOLS = [
    "aic_abi_version", "aic_create", "aic_destroy", "aic_last_error", "aic_device_name", "aic_upload_space",
    "aic_clear_space", "aic_update_cubes", "aic_update_light_volume", "aic_replace_block", "aic_replace_blocks", "aic_compact", "aic_set_options",
    "aic_render", "aic_render_submit", "aic_render_wait", "aic_render_submit_batch", "aic_render_wait_batch", "aic_trace_patches", "aic_partition_rows", "aic_assemble_strips", "aic_assemble_strips_async", "aic_assemble_strips_on", "aic_read_aux", "aic_synchronize", "aic_stream", "aic_wait_event", "aic_stream_wait_frame",
    "aic_probe_raycast", "aic_probe_light_lut", "aic_probe_powf",
    "aic_ortho_image_size", "aic_render_orthographic",
    "aic_evaluate_light", "aic_evaluate_light_submit", "aic_evaluate_light_wait", "aic_evaluate_light_poll", "aic_light_cubes_changed", "aic_read_light_volume", "aic_read_light_cubes", "aic_light_chart", "aic_probe_derived", "aic_probe_log2f",
    "aic_create_multi", "aic_destroy_multi", "aic_multi_device_count", "aic_multi_context", "aic_multi_last_error", "aic_multi_upload_space",
    "aic_multi_clear_space", "aic_multi_update_cubes", "aic_multi_update_light_volume", "aic_multi_evaluate_light", "aic_multi_light_cubes_changed", "aic_multi_replace_blocks", "aic_multi_set_options",
    "aic_multi_render", "aic_multi_render_submit", "aic_multi_render_wait",
]


class AicError(RuntimeError):
    """Maps to `RenderError` on the reference side (all-is-cubes-render/src/lib.rs:46-54)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code


class BlockDesc(C.Structure):
    _fields_ = [("resolution", C.c_int32), ("vlo", C.c_int32 * 3), ("vsize", C.c_int32 * 3), ("vox_off", C.c_uint32),
                ("pal_off", C.c_uint32), ("pal_len", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_int32)]


class LightParams(C.Structure):
    _fields_ = [("maximum_distance", C.c_int32), ("fast", C.c_int32), ("epsilon", C.c_int32), ("batch", C.c_int32),
                ("queue_order", C.c_int32), ("n_queue", C.c_int32), ("lanes_per_cube", C.c_int32), ("hooks", C.c_int32), ("queue_cubes", C.c_void_p), ("queue_priorities", C.c_void_p),
                ("max_updates", C.c_uint64)]


class LightInfo(C.Structure):
    _fields_ = [("updates", C.c_uint64), ("batches", C.c_uint64), ("cost", C.c_uint64), ("device_ms", C.c_double),
                ("total_ms", C.c_double), ("queue_left", C.c_uint32), ("pad", C.c_uint32), ("bundles_visited", C.c_uint64)]


def light_chart():
    """The light propagation chart the library generates (chart/generator.rs): (weights [n,6] f32, children [n,6] u32, depth).
    Host-only: needs no device."""
    lib = load()
    lib.aic_light_chart.restype = C.c_uint32
    lib.aic_light_chart.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    depth = C.c_uint32(0)
    n = lib.aic_light_chart(None, None, C.byref(depth))
    w = np.zeros((n, 6), np.float32)
    ch = np.zeros((n, 6), np.uint32)
    lib.aic_light_chart(w.ctypes.data, ch.ctypes.data, C.byref(depth))
    return w, ch, int(depth.value)


class SpaceDesc(C.Structure):
    _fields_ = [("lo", C.c_int32 * 3), ("size", C.c_int32 * 3), ("block_index", C.c_void_p), ("light", C.c_void_p),
                ("n_blocks", C.c_uint32), ("blocks", C.c_void_p), ("voxels", C.c_void_p), ("n_voxels", C.c_uint64),
                ("palette", C.c_void_p), ("n_palette", C.c_uint64), ("sky_kind", C.c_int32), ("sky", (C.c_float * 3) * 8),
                ("block_sky", (C.c_uint8 * 4) * 7)]


class Options(C.Structure):
    _fields_ = [("fog", C.c_int32), ("transparency", C.c_int32), ("threshold", C.c_float), ("lighting", C.c_int32),
                ("bounce_samples", C.c_int32), ("antialiasing", C.c_int32), ("debug_pixel_cost", C.c_int32),
                ("tone_mapping", C.c_int32), ("maximum_intensity", C.c_float), ("bloom_intensity", C.c_float),
                ("view_distance", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("inverse_projection_view", C.c_double * 16), ("exposure", C.c_float), ("reserved", C.c_int32)]


class Partition(C.Structure):
    _fields_ = [("strip_rows", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32), ("reserved", C.c_uint32)]


class FrameDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("world", Camera), ("ui", Camera), ("backdrop", C.c_float * 4),
                ("partition", Partition), ("flags", C.c_uint32), ("tuning", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [("cubes_traced", C.c_uint64), ("n_outer", C.c_uint64), ("n_inner", C.c_uint64), ("n_hits", C.c_uint64),
                ("n_light", C.c_uint64), ("kernel_ms", C.c_float), ("total_ms", C.c_float), ("rows_rendered", C.c_uint32),
                ("flaws", C.c_uint32), ("variant", C.c_uint32), ("tile_queues", C.c_uint32)]


PIXEL_AUX_DTYPE = np.dtype(
    [("hit", "<i4"), ("cube", "<i4", (3,)), ("voxel", "<i4", (3,)), ("resolution", "<i4"), ("face", "<i4"),
     ("block_index", "<i4"), ("cubes_traced", "<u4"), ("layer", "<u4"), ("t_distance", "<f8")],
    align=True,
)
RC_STEP_DTYPE = np.dtype([("cube", "<i4", (3,)), ("face", "<i4"), ("t_distance", "<f8"), ("intersection_point", "<f8", (3,))], align=True)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads libaic_hip.so; raises if it has not been built (run `python __graft_entry__.py`)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc, gfx950)")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libaic_hip.so is linked against /opt/rocm's. Loaded
        # after torch, this library binds to the copy already in the process; loaded BEFORE torch, the process ends up with two
        # runtimes and the second one to initialise finds no GPU ("No HIP GPUs are available" from torch.zeros(device="cuda") --
        # met by tests/test_gpu_light_update.py run on its own). A process that has torch gets it loaded first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(str(LIB_PATH))
        lib.aic_create.restype = C.c_void_p
        lib.aic_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
        lib.aic_destroy.argtypes = [C.c_void_p]
        lib.aic_destroy.restype = None
        lib.aic_last_error.restype = C.c_char_p
        lib.aic_last_error.argtypes = [C.c_void_p]
        lib.aic_stream.restype = C.c_void_p
        lib.aic_stream.argtypes = [C.c_void_p]
        lib.aic_partition_rows.restype = C.c_uint32
        lib.aic_partition_rows.argtypes = [C.c_uint32, C.POINTER(Partition)]
        lib.aic_upload_space.argtypes = [C.c_void_p, C.c_int, C.POINTER(SpaceDesc)]
        lib.aic_clear_space.argtypes = [C.c_void_p, C.c_int]
        lib.aic_compact.argtypes = [C.c_void_p, C.c_int]
        lib.aic_update_cubes.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.aic_update_light_volume.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_replace_block.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(BlockDesc), C.c_void_p, C.c_void_p]
        lib.aic_replace_blocks.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.aic_set_options.argtypes = [C.c_void_p, C.c_int, C.POINTER(Options)]
        lib.aic_render.argtypes = [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.c_int, C.POINTER(FrameInfo)]
        lib.aic_render_submit.argtypes = [C.c_void_p, C.POINTER(FrameDesc), C.c_void_p, C.c_uint32]
        lib.aic_render_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(FrameInfo)]
        lib.aic_render_submit_batch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(FrameDesc), C.POINTER(C.c_void_p), C.c_uint32]
        lib.aic_render_wait_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(FrameInfo)]
        lib.aic_trace_patches.argtypes = [C.c_void_p, C.POINTER(FrameDesc), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameInfo)]
        lib.aic_assemble_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.aic_assemble_strips_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.aic_read_aux.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        lib.aic_synchronize.argtypes = [C.c_void_p]
        lib.aic_wait_event.argtypes = [C.c_void_p, C.c_void_p]
        lib.aic_stream_wait_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.aic_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        lib.aic_probe_raycast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        lib.aic_probe_light_lut.argtypes = [C.c_void_p, C.c_void_p]
        lib.aic_probe_powf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.aic_evaluate_light.argtypes = [C.c_void_p, C.c_int, C.POINTER(LightParams), C.POINTER(LightInfo)]
        lib.aic_evaluate_light_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(LightParams)]
        lib.aic_evaluate_light_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(LightInfo)]
        lib.aic_evaluate_light_poll.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.aic_read_light_volume.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_read_light_cubes.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.aic_light_cubes_changed.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int]
        lib.aic_probe_derived.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.aic_probe_log2f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        assert C.sizeof(BlockDesc) == 48
        _lib = lib
    return _lib


def packed_light_scalar_in(value: float) -> int:
    """PackedLight::scalar_in (all-is-cubes/src/space/light/data.rs:214-218)."""
    v = np.float32(value)
    if not v > 0:
        return 0
    x = float(np.float32(np.float32(np.log2(v)) * np.float32(10.0)) + np.float32(144.0))
    x = float(np.float32(x))
    r = math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)  # f32::round: half away from zero
    return int(min(max(r, 0), 255))


def block_sky_texels(sky_kind: int, sky: np.ndarray) -> np.ndarray:
    """Sky::for_blocks (all-is-cubes/src/space/sky.rs:45-82) as 7 texels: nx ny nz px py pz mean."""
    sky = np.asarray(sky, np.float32).reshape(8, 3)
    out = np.zeros((7, 4), np.uint8)
    out[:, 3] = 255  # LightStatus::Visible

    def some(rgb):
        return [packed_light_scalar_in(float(c)) for c in rgb]

    if sky_kind == 0:
        out[:, 0:3] = some(sky[0])
        return out
    # images of +X,+Y,+Z under Face::rotation_from_nz (face.rs:395-404)
    bases = {
        0: ((0, 1, 0), (0, 0, 1), (1, 0, 0)),  # NX RYZX
        1: ((0, 0, 1), (1, 0, 0), (0, 1, 0)),  # NY RZXY
        2: ((1, 0, 0), (0, 1, 0), (0, 0, 1)),  # NZ RXYZ
        3: ((0, -1, 0), (0, 0, 1), (-1, 0, 0)),  # PX RyZx
        4: ((0, 0, 1), (-1, 0, 0), (0, -1, 0)),  # PY RZxy
        5: ((1, 0, 0), (0, -1, 0), (0, 0, -1)),  # PZ RXyz
    }
    pts = [(-1, -1, -1), (-1, 1, -1), (1, -1, -1), (1, 1, -1)]
    for f in range(6):
        bx, by, bz = (np.array(v) for v in bases[f])
        acc = np.zeros(3, np.float32)
        for p in pts:
            d = p[0] * bx + p[1] * by + p[2] * bz
            idx = ((1 if d[0] >= 0 else 0) << 2) + ((1 if d[1] >= 0 else 0) << 1) + (1 if d[2] >= 0 else 0)
            acc = (acc + sky[idx]).astype(np.float32)
        out[f, 0:3] = some(acc * np.float32(0.25))
    acc = np.zeros(3, np.float32)
    for k in range(8):
        acc = (acc + sky[k]).astype(np.float32)
    out[6, 0:3] = some(acc * np.float32(1.0 / 8.0))
    return out


def make_options(fog=1, transparency=1, threshold=0.5, lighting=3, bounce_samples=0, antialiasing=0, debug_pixel_cost=False,
                 tone_mapping=0, maximum_intensity=float("inf"), bloom_intensity=0.125, view_distance=200.0) -> Options:
    """Defaults = GraphicsOptions::default() (camera/graphics_options.rs:256-280)."""
    o = Options()
    o.fog, o.transparency, o.threshold, o.lighting = fog, transparency, threshold, lighting
    o.bounce_samples, o.antialiasing, o.debug_pixel_cost = bounce_samples, antialiasing, int(debug_pixel_cost)
    o.tone_mapping, o.maximum_intensity, o.bloom_intensity = tone_mapping, maximum_intensity, bloom_intensity
    o.view_distance = min(max(view_distance, 1.0), 10000.0)  # GraphicsOptions::repair
    return o


def unaltered_colors(**kw) -> Options:
    """GraphicsOptions::UNALTERED_COLORS (graphics_options.rs:168-190)."""
    base = dict(fog=0, lighting=0, transparency=1, bloom_intensity=0.0)
    base.update(kw)
    return make_options(**base)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data


class Context:
    """One device-resident renderer state = `RtRenderer` minus cameras."""

    def __init__(self, device_id: int = -1):
        self._lib = load()
        st = C.c_int(0)
        self._h = self._lib.aic_create(device_id, C.byref(st))
        if not self._h:
            raise AicError(st.value, "aic_create failed (no usable MI355X / HIP device?)")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.aic_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != AIC_OK:
            raise AicError(rc, self._lib.aic_last_error(self._h).decode())

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self._lib.aic_device_name(self._h, buf, 256))
        return buf.value.decode()

    @property
    def stream(self) -> int:
        return int(self._lib.aic_stream(self._h) or 0)

    # -- scene ---------------------------------------------------------------------------
    @staticmethod
    def _space_desc(flat_space):
        """(SpaceDesc, keep-alive) for aic_upload_space / aic_multi_upload_space."""
        p = flat_space.pack() if hasattr(flat_space, "pack") else flat_space
        d = SpaceDesc()
        d.lo[:] = [int(v) for v in p.lo]
        d.size[:] = [int(v) for v in p.size]
        d.block_index = _ptr(p.block_index)
        d.light = _ptr(p.light)
        d.n_blocks = len(p.blocks)
        d.blocks = _ptr(p.blocks)
        d.voxels = _ptr(p.voxels)
        d.n_voxels = p.voxels.size
        d.palette = _ptr(p.palette)
        d.n_palette = len(p.palette)
        d.sky_kind = p.sky_kind
        for i in range(8):
            for j in range(3):
                d.sky[i][j] = float(p.sky[i, j])
        bs = block_sky_texels(p.sky_kind, p.sky)
        for i in range(7):
            for j in range(4):
                d.block_sky[i][j] = int(bs[i, j])
        return d, p

    def upload_space(self, layer: int, flat_space) -> None:
        d, _keep = self._space_desc(flat_space)
        self._check(self._lib.aic_upload_space(self._h, layer, C.byref(d)))

    def clear_space(self, layer: int) -> None:
        self._check(self._lib.aic_clear_space(self._h, layer))

    def update_cubes(self, layer: int, xyz, block_index=None, light=None) -> None:
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        bi = None if block_index is None else np.ascontiguousarray(block_index, np.uint16)
        lt = None if light is None else np.ascontiguousarray(light, np.uint8).reshape(-1, 4)
        self._check(self._lib.aic_update_cubes(self._h, layer, len(xyz), _ptr(xyz), _ptr(bi), _ptr(lt)))

    def update_light_volume(self, layer: int, light: np.ndarray) -> None:
        lt = np.ascontiguousarray(light, np.uint8)
        self._check(self._lib.aic_update_light_volume(self._h, layer, _ptr(lt)))

    def replace_block(self, layer: int, index: int, block) -> None:
        from . import flat

        d = BlockDesc()
        d.resolution = block.resolution
        d.vlo[:] = list(block.vlo)
        d.vsize[:] = list(block.voxels.shape)
        d.pal_len = len(block.palette)
        d.flags = (flat.FLAG_ONE if block.is_one else 0) | (flat.FLAG_AIR if block.is_air else 0)
        vox = np.ascontiguousarray(block.voxels, np.uint16)
        pal = np.ascontiguousarray(block.palette, np.float32)
        self._check(self._lib.aic_replace_block(self._h, layer, index, C.byref(d), _ptr(vox), _ptr(pal)))

    def render_orthographic(self, layer: int = LAYER_WORLD, resolution: int = 32):
        """raytracer::ortho::render_orthographic of the layer's space: dict(rgba8 [h,w,4], info)."""
        f = self._lib.aic_render_orthographic
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
        w, h = C.c_uint32(0), C.c_uint32(0)
        self._check(f(self._h, layer, resolution, None, 0, C.byref(w), C.byref(h), None))
        out = np.zeros((h.value, w.value, 4), np.uint8)
        info = FrameInfo()
        self._check(f(self._h, layer, resolution, _ptr(out), 0, C.byref(w), C.byref(h), C.byref(info)))
        return {"rgba8": out, "info": info}

    def compact(self, layer: int) -> None:
        self._check(self._lib.aic_compact(self._h, C.c_int(layer)))

    def replace_blocks(self, layer: int, items) -> None:
        """`items`: list of (index, flat.BlockDef) -- a batch of BlockEvaluation changes under one synchronisation."""
        n = len(items)
        idx = np.ascontiguousarray([i for i, _ in items], np.uint32)
        descs = (BlockDesc * max(n, 1))()
        keep, vp, pp = [], (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))()
        for k, (_, block) in enumerate(items):
            d = descs[k]
            d.resolution = block.resolution
            d.vlo[:] = list(block.vlo)
            d.vsize[:] = list(block.voxels.shape)
            d.pal_len = len(block.palette)
            d.flags = (flat.FLAG_ONE if block.is_one else 0) | (flat.FLAG_AIR if block.is_air else 0)
            vox = np.ascontiguousarray(block.voxels, np.uint16)
            pal = np.ascontiguousarray(block.palette, np.float32)
            keep += [vox, pal]
            vp[k], pp[k] = vox.ctypes.data, pal.ctypes.data
        self._check(self._lib.aic_replace_blocks(self._h, layer, n, _ptr(idx), C.cast(descs, C.c_void_p), C.cast(vp, C.c_void_p), C.cast(pp, C.c_void_p)))

    def set_options(self, layer: int, options: Options) -> None:
        self._check(self._lib.aic_set_options(self._h, layer, C.byref(options)))

    # -- drawing ---------------------------------------------------------------------------
    @staticmethod
    def make_frame(width, height, world_inv=None, ui_inv=None, exposure=1.0, ui_exposure=1.0, backdrop=(0, 0, 0, 0),
                   partition=None, flags=0, tuning=0) -> FrameDesc:
        f = FrameDesc()
        f.width, f.height = int(width), int(height)
        ident = np.eye(4).reshape(16)
        w = ident if world_inv is None else np.ascontiguousarray(world_inv, np.float64).reshape(16)
        u = ident if ui_inv is None else np.ascontiguousarray(ui_inv, np.float64).reshape(16)
        f.world.inverse_projection_view[:] = [float(v) for v in w]
        f.ui.inverse_projection_view[:] = [float(v) for v in u]
        f.world.exposure, f.ui.exposure = exposure, ui_exposure
        f.backdrop[:] = [float(v) for v in backdrop]
        if partition is not None:
            f.partition.strip_rows, f.partition.n_parts, f.partition.part = (int(v) for v in partition)
        f.flags = flags
        f.tuning = int(tuning)
        return f

    def partition_rows(self, height: int, partition) -> int:
        p = Partition(int(partition[0]), int(partition[1]), int(partition[2]), 0)
        return int(self._lib.aic_partition_rows(height, C.byref(p)))

    def render(self, frame: FrameDesc, want_aux: bool = False, counters: bool = False):
        """Returns dict(rgba8 [rows,w,4], info FrameInfo, aux or None). `frame` is left as it was given (until round 6 want_aux / counters
        stayed set in it, and a later plain render of the same object quietly ran the recording variant again)."""
        keep_flags = frame.flags
        if want_aux:
            frame.flags |= FRAME_AUX
        if counters:
            frame.flags |= FRAME_COUNTERS
        rows = int(self._lib.aic_partition_rows(frame.height, C.byref(frame.partition)))
        floats = bool(frame.flags & (FRAME_OUT_LINEAR | FRAME_OUT_COLORBUF))  # 16-byte float pixels instead of RGBA8
        out = np.zeros((rows, frame.width, 4), np.float32 if floats else np.uint8)
        info = FrameInfo()
        try:
            self._check(self._lib.aic_render(self._h, C.byref(frame), _ptr(out), 0, C.byref(info)))
        finally:
            frame.flags = keep_flags
        aux = None
        if want_aux:
            aux = np.zeros((rows, frame.width), PIXEL_AUX_DTYPE)
            if aux.size:
                self._check(self._lib.aic_read_aux(self._h, _ptr(aux), aux.size))
        return {"rgba8": out, "info": info, "aux": aux}

    def render_to_device(self, frame: FrameDesc, device_ptr: int) -> FrameInfo:
        """Leaves the RGBA8 rows in HBM at `device_ptr` (e.g. a torch tensor's data_ptr())."""
        info = FrameInfo()
        self._check(self._lib.aic_render(self._h, C.byref(frame), C.c_void_p(device_ptr), 1, C.byref(info)))
        return info

    def trace_patches(self, frame: FrameDesc, rects, want_aux: bool = False):
        """RtScene::trace_patch for a batch of NDC rectangles [n,4] = (min.x, min.y, max.x, max.y)."""
        r = np.ascontiguousarray(rects, np.float64).reshape(-1, 4)
        out = np.zeros((len(r), 4), np.float32 if frame.flags & (FRAME_OUT_LINEAR | FRAME_OUT_COLORBUF) else np.uint8)
        aux = np.zeros(len(r), PIXEL_AUX_DTYPE) if want_aux else None
        info = FrameInfo()
        self._check(self._lib.aic_trace_patches(self._h, C.byref(frame), len(r), _ptr(r), _ptr(out), _ptr(aux), C.byref(info)))
        return {"rgba8": out, "aux": aux, "info": info}

    def render_submit(self, frame: FrameDesc, device_ptr: int, slot: int) -> None:
        """Queues a frame on `slot` (0..MAX_IN_FLIGHT-1); returns without waiting (aic_render_submit)."""
        self._check(self._lib.aic_render_submit(self._h, C.byref(frame), C.c_void_p(device_ptr), int(slot)))

    def render_submit_batch(self, frames, device_ptrs, slot: int) -> None:
        """aic_render_submit_batch: 1, 2, 4 or 8 frames (same size, partition, flags) traced by one launch, frame i into device_ptrs[i]."""
        arr = (FrameDesc * len(frames))(*frames)
        ptrs = (C.c_void_p * len(frames))(*[int(p) for p in device_ptrs])
        self._check(self._lib.aic_render_submit_batch(self._h, len(frames), arr, ptrs, slot))

    def render_wait_batch(self, slot: int, n_frames: int):
        infos = (FrameInfo * n_frames)()
        self._check(self._lib.aic_render_wait_batch(self._h, slot, n_frames, infos))
        return list(infos)

    def render_wait(self, slot: int) -> FrameInfo:
        info = FrameInfo()
        self._check(self._lib.aic_render_wait(self._h, int(slot), C.byref(info)))
        return info

    def assemble_strips(self, gathered_ptr: int, out_ptr: int, width: int, height: int, strip_rows: int, n_parts: int) -> None:
        self._check(self._lib.aic_assemble_strips(self._h, C.c_void_p(gathered_ptr), C.c_void_p(out_ptr), width, height, strip_rows, n_parts))

    def synchronize(self) -> None:
        self._check(self._lib.aic_synchronize(self._h))

    def stream_wait_frame(self, slot: int, hip_stream: int) -> None:
        """`hip_stream` (a raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) waits on the device for the frame
        submitted on `slot`; the host does not block (aic_stream_wait_frame)."""
        self._check(self._lib.aic_stream_wait_frame(self._h, int(slot), C.c_void_p(hip_stream)))

    def wait_event(self, hip_event: int) -> None:
        """Everything the context queues from now on waits for the caller's hipEvent_t (aic_wait_event)."""
        self._check(self._lib.aic_wait_event(self._h, C.c_void_p(hip_event)))

    # -- probes ----------------------------------------------------------------------------
    def probe_raycast(self, origin, direction, bounds=None, include_exit=True, max_steps=64):
        o = np.ascontiguousarray(origin, np.float64).reshape(3)
        d = np.ascontiguousarray(direction, np.float64).reshape(3)
        lo = np.zeros(3, np.int32) if bounds is None else np.ascontiguousarray(bounds[0], np.int32)
        hi = np.zeros(3, np.int32) if bounds is None else np.ascontiguousarray(bounds[1], np.int32)
        out = np.zeros(max_steps, RC_STEP_DTYPE)
        n = C.c_uint32(0)
        ended = C.c_int(0)
        self._check(self._lib.aic_probe_raycast(self._h, o.ctypes.data, d.ctypes.data, int(bounds is not None), lo.ctypes.data,
                                                hi.ctypes.data, int(include_exit), max_steps, out.ctypes.data, C.byref(n), C.byref(ended)))
        return out[: n.value], bool(ended.value)

    def probe_powf(self, x, y) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        out = np.zeros(x.shape, np.float32)
        self._check(self._lib.aic_probe_powf(self._h, _ptr(x), _ptr(y), x.size, _ptr(out)))
        return out

    def evaluate_light(self, layer: int, maximum_distance: int, fast: bool = True, epsilon: int = 1, batch: int = 32,
                       queue_order: int = 16, queue=None, max_updates: int = 0, lanes_per_cube: int = 0, session: bool = False,
                       dep_pool_chunks: int = 0) -> LightInfo:
        """`Mutation::fast_evaluate_light` (if `fast`) then `Mutation::evaluate_light(epsilon)` (space.rs:1496-1540) on
        the uploaded space, compute_light on the device. `queue`: None = every Uninitialized texel (when not `fast`), or a
        list of ((x, y, z), priority). The layer's light volume is updated in place."""
        hooks = (LIGHT_HOOK_SESSION if session else 0) | ((int(dep_pool_chunks) & 0xffff) << LIGHT_HOOK_POOL_SHIFT)  # (test hooks: aic_light_params.hooks)
        p = LightParams(maximum_distance, int(fast), epsilon, batch, queue_order, -1, lanes_per_cube, hooks, None, None, max_updates)
        keep = []
        if queue is not None:
            qc = np.ascontiguousarray([q[0] for q in queue], np.int32).reshape(-1, 3)
            qp = np.ascontiguousarray([q[1] for q in queue], np.int32)
            keep = [qc, qp]
            p.n_queue = len(qp)
            p.queue_cubes = qc.ctypes.data
            p.queue_priorities = qp.ctypes.data
        info = LightInfo()
        self._check(self._lib.aic_evaluate_light(self._h, layer, C.byref(p), C.byref(info)))
        del keep
        return info

    def evaluate_light_submit(self, layer: int, maximum_distance: int, fast: bool = True, epsilon: int = 1, batch: int = 32, queue_order: int = 16,
                              queue=None, max_updates: int = 0, lanes_per_cube: int = 0) -> None:
        """aic_evaluate_light_submit: the same update on the context's worker thread; `evaluate_light_wait` publishes and reports it."""
        p = LightParams(maximum_distance, int(fast), epsilon, batch, queue_order, -1, lanes_per_cube, 0, None, None, max_updates)
        keep = []
        if queue is not None:
            qc = np.ascontiguousarray([q[0] for q in queue], np.int32).reshape(-1, 3)
            qp = np.ascontiguousarray([q[1] for q in queue], np.int32)
            keep = [qc, qp]
            p.n_queue = len(qp)
            p.queue_cubes = qc.ctypes.data
            p.queue_priorities = qp.ctypes.data
        self._check(self._lib.aic_evaluate_light_submit(self._h, layer, C.byref(p)))
        del keep  # (the library copied the arrays)

    def evaluate_light_done(self, layer: int) -> bool:
        """aic_evaluate_light_poll: False while a submitted update is still running."""
        d = C.c_int(1)
        self._check(self._lib.aic_evaluate_light_poll(self._h, layer, C.byref(d)))
        return bool(d.value)

    def evaluate_light_wait(self, layer: int) -> LightInfo:
        info = LightInfo()
        self._check(self._lib.aic_evaluate_light_wait(self._h, layer, C.byref(info)))
        return info

    def light_cubes_changed(self, layer: int, xyz, queue_order: int = 16) -> None:
        """`modified_cube_needs_update` (updater.rs:135-173) for cubes just changed with `update_cubes`: queues them and their
        neighbours on the layer's light update queue (drained by `evaluate_light(fast=False, queue=[])`)."""
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        self._check(self._lib.aic_light_cubes_changed(self._h, layer, len(xyz), _ptr(xyz), queue_order))

    def read_light_volume(self, layer: int, shape) -> np.ndarray:
        out = np.zeros(tuple(shape) + (4,), np.uint8)
        self._check(self._lib.aic_read_light_volume(self._h, layer, out.ctypes.data))
        return out

    def read_light_cubes(self, layer: int, xyz) -> np.ndarray:
        """The texels of the listed cubes, [n][4] (`LightStorage::get`, light/data.rs, over a list)."""
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        out = np.zeros((len(xyz), 4), np.uint8)
        self._check(self._lib.aic_read_light_cubes(self._h, layer, len(xyz), _ptr(xyz), out.ctypes.data))
        return out

    def probe_derived(self, layer: int, n_blocks: int):
        out = np.zeros((n_blocks, 32), np.float32)
        opq = np.zeros((n_blocks, 6), np.uint8)
        self._check(self._lib.aic_probe_derived(self._h, layer, out.ctypes.data, opq.ctypes.data))
        return out, opq

    def probe_log2f(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(x.shape, np.float32)
        self._check(self._lib.aic_probe_log2f(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def probe_light_lut(self) -> np.ndarray:
        out = np.zeros(256, np.float32)
        self._check(self._lib.aic_probe_light_lut(self._h, out.ctypes.data))
        return out


class MultiContext:
    """`aic_multi`: one object over several devices (ids may repeat), the form a Rust `HipRtRenderer` would hold.
    Scene calls are replicated; `render` deals 8-row strips to the devices and assembles the frame on the first."""

    def __init__(self, device_ids):
        self._lib = load()
        lib = self._lib
        lib.aic_create_multi.restype = C.c_void_p
        lib.aic_create_multi.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        lib.aic_destroy_multi.argtypes = [C.c_void_p]
        lib.aic_multi_last_error.restype = C.c_char_p
        lib.aic_multi_last_error.argtypes = [C.c_void_p]
        lib.aic_multi_device_count.argtypes = [C.c_void_p]
        lib.aic_multi_upload_space.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_multi_clear_space.argtypes = [C.c_void_p, C.c_int]
        lib.aic_multi_set_options.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_multi_update_light_volume.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_multi_evaluate_light.argtypes = [C.c_void_p, C.c_int, C.POINTER(LightParams), C.POINTER(LightInfo)]
        lib.aic_multi_light_cubes_changed.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int]
        lib.aic_multi_update_cubes.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.aic_multi_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.aic_multi_render_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32]
        lib.aic_multi_render_wait.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        ids = np.ascontiguousarray(device_ids, np.int32)
        st = C.c_int(0)
        self._h = lib.aic_create_multi(len(ids), _ptr(ids), C.byref(st))
        if not self._h:
            raise AicError(st.value, "aic_create_multi failed (no usable MI355X?)")

    def close(self) -> None:
        if self._h:
            self._lib.aic_destroy_multi(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != AIC_OK:
            raise AicError(rc, self._lib.aic_multi_last_error(self._h).decode())

    @property
    def device_count(self) -> int:
        return int(self._lib.aic_multi_device_count(self._h))

    def upload_space(self, layer: int, flat_space) -> None:
        d, _keep = Context._space_desc(flat_space)
        self._check(self._lib.aic_multi_upload_space(self._h, layer, C.byref(d)))

    def clear_space(self, layer: int) -> None:
        self._check(self._lib.aic_multi_clear_space(self._h, layer))

    def set_options(self, layer: int, options: Options) -> None:
        self._check(self._lib.aic_multi_set_options(self._h, layer, C.byref(options)))

    def update_light_volume(self, layer: int, light) -> None:
        lt = np.ascontiguousarray(light, np.uint8)
        self._check(self._lib.aic_multi_update_light_volume(self._h, layer, _ptr(lt)))

    def evaluate_light(self, layer: int, maximum_distance: int, fast: bool = True, epsilon: int = 1, batch: int = 32, queue_order: int = 16) -> LightInfo:
        p = LightParams(maximum_distance, int(fast), epsilon, batch, queue_order, -1, 0, 0, None, None, 0)
        info = LightInfo()
        self._check(self._lib.aic_multi_evaluate_light(self._h, layer, C.byref(p), C.byref(info)))
        return info

    def light_cubes_changed(self, layer: int, xyz, queue_order: int = 16) -> None:
        """`Context.light_cubes_changed` on the device that runs the updater; the texels it writes reach the others at once."""
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        self._check(self._lib.aic_multi_light_cubes_changed(self._h, layer, len(xyz), _ptr(xyz), queue_order))

    def update_cubes(self, layer: int, xyz, block_index=None, light=None) -> None:
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        bi = None if block_index is None else np.ascontiguousarray(block_index, np.uint16)
        lt = None if light is None else np.ascontiguousarray(light, np.uint8).reshape(-1, 4)
        self._check(self._lib.aic_multi_update_cubes(self._h, layer, len(xyz), _ptr(xyz), _ptr(bi), _ptr(lt)))

    def render(self, frame: FrameDesc):
        out = np.zeros((frame.height, frame.width, 4), np.uint8)
        info = FrameInfo()
        self._check(self._lib.aic_multi_render(self._h, C.byref(frame), _ptr(out), 0, C.byref(info)))
        return {"rgba8": out, "info": info}

    def render_submit(self, frame: FrameDesc, slot: int, device_ptr: int = 0):
        """aic_multi_render_submit: the frame is queued on `slot` of every device; returns the host array the wait fills (or None: device_ptr)."""
        out = None if device_ptr else np.zeros((frame.height, frame.width, 4), np.uint8)
        self._check(self._lib.aic_multi_render_submit(self._h, C.byref(frame), C.c_void_p(device_ptr) if device_ptr else _ptr(out), 1 if device_ptr else 0, slot))
        return out

    def render_wait(self, slot: int) -> FrameInfo:
        info = FrameInfo()
        self._check(self._lib.aic_multi_render_wait(self._h, slot, C.byref(info)))
        return info
