"""Reader, writer and replayer of call dumps (`AIC_DUMP=<path>`).

With `AIC_DUMP` set, `libaic_hip.so` appends every argument it is given -- scene snapshots, cube /
light / block deltas, options, frame descriptors -- to a file, verbatim. That is how a scene the
reference produces (DemoCity, the real Atrium: generators that need the un-vendored noise crate and
the block-evaluation engine) can be captured on a machine where the Rust shim runs and replayed
here: on the device (`replay`) or, in the tests, through the CPU oracle.

File layout (little endian): b"AICDUMP1", then records {u32 tag, u32 layer_or_slot, u64 payload
bytes, payload}. Payloads are the C structs of include/aic_hip.h followed by the arrays they point
to, in declaration order (see `_parse`).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional

import numpy as np

from . import flat

MAGIC = b"AICDUMP1"
UPLOAD, CLEAR, CUBES, LIGHT, BLOCK, OPTIONS, FRAME = 1, 2, 3, 4, 5, 6, 7

_UPLOAD_HDR = np.dtype([("lo", "<i4", 3), ("size", "<i4", 3), ("n_blocks", "<u4"), ("sky_kind", "<i4"), ("n_voxels", "<u8"),
                        ("n_palette", "<u8"), ("sky", "<f4", (8, 3)), ("block_sky", "u1", (7, 4))], align=True)
_OPTIONS = np.dtype([("fog", "<i4"), ("transparency", "<i4"), ("threshold", "<f4"), ("lighting", "<i4"), ("bounce_samples", "<i4"),
                     ("antialiasing", "<i4"), ("debug_pixel_cost", "<i4"), ("tone_mapping", "<i4"), ("maximum_intensity", "<f4"),
                     ("bloom_intensity", "<f4"), ("view_distance", "<f8")], align=True)
_CAMERA = np.dtype([("inverse_projection_view", "<f8", 16), ("exposure", "<f4"), ("reserved", "<i4")], align=True)
_FRAME = np.dtype([("width", "<u4"), ("height", "<u4"), ("world", _CAMERA), ("ui", _CAMERA), ("backdrop", "<f4", 4),
                   ("partition", "<u4", 4), ("flags", "<u4"), ("tuning", "<u4")], align=True)


@dataclass
class Record:
    tag: int
    layer: int  # layer for scene calls, slot for frames
    data: Dict[str, object]


def _parse(tag: int, payload: memoryview) -> Dict[str, object]:
    buf = np.frombuffer(payload, np.uint8)
    off = 0

    def take(dtype, count):
        nonlocal off
        dt = np.dtype(dtype)
        n = dt.itemsize * int(count)
        out = buf[off:off + n].view(dt).copy()
        off += n
        return out

    if tag == UPLOAD:
        h = take(_UPLOAD_HDR, 1)[0]
        n = int(np.prod(h["size"].astype(np.int64)))
        return {"header": h, "block_index": take("<u2", n), "light": take("u1", n * 4).reshape(n, 4),
                "blocks": take(flat.BLOCK_DTYPE, h["n_blocks"]), "voxels": take("<u2", h["n_voxels"]),
                "palette": take("<f4", int(h["n_palette"]) * 8).reshape(-1, 8)}
    if tag == CLEAR:
        return {}
    if tag == CUBES:
        n, has_bi, has_light, _ = (int(v) for v in take("<u4", 4))
        return {"xyz": take("<i4", n * 3).reshape(n, 3), "block_index": take("<u2", n) if has_bi else None,
                "light": take("u1", n * 4).reshape(n, 4) if has_light else None}
    if tag == LIGHT:
        n = int(take("<u8", 1)[0])
        return {"light": take("u1", n * 4).reshape(n, 4)}
    if tag == BLOCK:
        index, nvox = (int(v) for v in take("<u8", 2))
        desc = take(flat.BLOCK_DTYPE, 1)[0]
        return {"index": index, "desc": desc, "voxels": take("<u2", nvox), "palette": take("<f4", int(desc["pal_len"]) * 8).reshape(-1, 8)}
    if tag == OPTIONS:
        return {"options": take(_OPTIONS, 1)[0]}
    if tag == FRAME:
        return {"frame": take(_FRAME, 1)[0]}
    raise ValueError(f"unknown dump record tag {tag}")


def read_dump(path) -> Iterator[Record]:
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != MAGIC:
        raise ValueError("not an AIC dump")
    off, mv = 8, memoryview(blob)
    while off < len(blob):
        if off + 16 > len(blob):
            raise ValueError("truncated dump record header")
        tag, layer, n = struct.unpack_from("<IIQ", blob, off)
        off += 16
        if off + n > len(blob):
            raise ValueError("truncated dump record payload")
        yield Record(tag, layer, _parse(tag, mv[off:off + n]))
        off += n


# ---- writing (the same format, from Python objects: fixtures and captures made without the library) ----
class DumpWriter:
    def __init__(self, path):
        self._f = open(path, "wb")
        self._f.write(MAGIC)

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _rec(self, tag, layer, *parts):
        body = b"".join(np.ascontiguousarray(p).tobytes() for p in parts if p is not None)
        self._f.write(struct.pack("<IIQ", tag, layer, len(body)))
        self._f.write(body)

    def upload_space(self, layer: int, space) -> None:
        from . import abi

        p = space.pack() if hasattr(space, "pack") else space
        h = np.zeros(1, _UPLOAD_HDR)
        h["lo"], h["size"], h["n_blocks"], h["sky_kind"] = p.lo, p.size, len(p.blocks), p.sky_kind
        h["n_voxels"], h["n_palette"], h["sky"] = p.voxels.size, len(p.palette), p.sky
        h["block_sky"] = abi.block_sky_texels(p.sky_kind, p.sky)
        self._rec(UPLOAD, layer, h, p.block_index.astype("<u2"), p.light.astype(np.uint8), p.blocks, p.voxels.astype("<u2"),
                  p.palette.astype("<f4"))

    def clear_space(self, layer: int) -> None:
        self._rec(CLEAR, layer)

    def update_cubes(self, layer: int, xyz, block_index=None, light=None) -> None:
        xyz = np.asarray(xyz, "<i4").reshape(-1, 3)
        self._rec(CUBES, layer, np.array([len(xyz), block_index is not None, light is not None, 0], "<u4"), xyz,
                  None if block_index is None else np.asarray(block_index, "<u2"), None if light is None else np.asarray(light, np.uint8))

    def update_light_volume(self, layer: int, light) -> None:
        light = np.asarray(light, np.uint8).reshape(-1, 4)
        self._rec(LIGHT, layer, np.array([len(light)], "<u8"), light)

    def replace_block(self, layer: int, index: int, block: flat.BlockDef) -> None:
        desc, vox, pal = flat.pack_block(block)
        self._rec(BLOCK, layer, np.array([index, vox.size], "<u8"), desc, vox.astype("<u2"), pal.astype("<f4"))

    def set_options(self, layer: int, options) -> None:
        self._rec(OPTIONS, layer, np.frombuffer(bytes(options), np.uint8))

    def frame(self, frame_desc, slot: int = 0) -> None:
        self._rec(FRAME, slot, np.frombuffer(bytes(frame_desc), np.uint8))


# ---- scene state reconstruction (what the layer holds after the recorded calls) ----
def _block_from_desc(desc, voxels: np.ndarray, palette: np.ndarray) -> flat.BlockDef:
    one = bool(int(desc["flags"]) & flat.FLAG_ONE)
    air = bool(int(desc["flags"]) & flat.FLAG_AIR)
    pal = np.ascontiguousarray(palette, np.float32).reshape(-1, 8)
    name = chr(int(desc["name_char"])) if 32 <= int(desc["name_char"]) < 127 else "#"
    if one:
        return flat.BlockDef(1, (0, 0, 0), np.zeros((1, 1, 1), np.uint16), pal[:1].copy(), is_one=True, is_air=air, name=name)
    vs = tuple(int(v) for v in desc["vsize"])
    return flat.BlockDef(int(desc["resolution"]), tuple(int(v) for v in desc["vlo"]), np.ascontiguousarray(voxels, np.uint16).reshape(vs),
                         pal.copy(), is_air=air, name=name)


def space_from_upload(data) -> flat.FlatSpace:
    h = data["header"]
    size = tuple(int(v) for v in h["size"])
    sp = flat.FlatSpace(tuple(int(v) for v in h["lo"]), size)
    sp.sky_kind = int(h["sky_kind"])
    sp.sky = np.array(h["sky"], np.float32).reshape(8, 3)
    sp.block_index[...] = data["block_index"].reshape(size)
    sp.light[...] = data["light"].reshape(size + (4,))
    for d in data["blocks"]:
        nv = 1 if int(d["flags"]) & flat.FLAG_ONE else int(np.prod([int(v) for v in d["vsize"]]))
        vox = data["voxels"][int(d["vox_off"]):int(d["vox_off"]) + nv]
        pal = data["palette"][int(d["pal_off"]):int(d["pal_off"]) + int(d["pal_len"])]
        sp.blocks.append(_block_from_desc(d, vox, pal))
    return sp


class SceneState:
    """Applies recorded scene calls to FlatSpaces, one per layer (updating.rs:107-172 semantics)."""

    def __init__(self):
        self.spaces: List[Optional[flat.FlatSpace]] = [None, None]
        self.options: List[Optional[np.void]] = [None, None]

    def apply(self, r: Record) -> None:
        d = r.data
        if r.tag == UPLOAD:
            self.spaces[r.layer] = space_from_upload(d)
        elif r.tag == CLEAR:
            self.spaces[r.layer] = None
        elif r.tag == CUBES:
            sp = self.spaces[r.layer]
            rel = d["xyz"] - np.array(sp.lo, np.int32)
            ok = ((rel >= 0) & (rel < np.array(sp.size))).all(axis=1)
            rel = rel[ok]
            if d["block_index"] is not None:
                sp.block_index[rel[:, 0], rel[:, 1], rel[:, 2]] = d["block_index"][ok]
            if d["light"] is not None:
                sp.light[rel[:, 0], rel[:, 1], rel[:, 2]] = d["light"][ok]
        elif r.tag == LIGHT:
            sp = self.spaces[r.layer]
            sp.light[...] = d["light"].reshape(sp.light.shape)
        elif r.tag == BLOCK:
            sp = self.spaces[r.layer]
            b = _block_from_desc(d["desc"], d["voxels"], d["palette"])
            if d["index"] == len(sp.blocks):
                sp.blocks.append(b)
            else:
                sp.blocks[d["index"]] = b
        elif r.tag == OPTIONS:
            self.options[r.layer] = d["options"]


# ---- replay on the device ----
def replay(path, ctx=None, device_id: int = -1) -> List[np.ndarray]:
    """Re-issues every recorded call through the C ABI; returns the RGBA8 image of every recorded frame."""
    from . import abi

    own = ctx is None
    ctx = ctx or abi.Context(device_id)
    frames = []
    try:
        for r in read_dump(path):
            d = r.data
            if r.tag == UPLOAD:
                ctx.upload_space(r.layer, space_from_upload(d))
            elif r.tag == CLEAR:
                ctx.clear_space(r.layer)
            elif r.tag == CUBES:
                ctx.update_cubes(r.layer, d["xyz"], d["block_index"], d["light"])
            elif r.tag == LIGHT:
                ctx.update_light_volume(r.layer, d["light"])
            elif r.tag == BLOCK:
                ctx.replace_block(r.layer, d["index"], _block_from_desc(d["desc"], d["voxels"], d["palette"]))
            elif r.tag == OPTIONS:
                ctx.set_options(r.layer, abi.Options.from_buffer_copy(d["options"].tobytes()))
            elif r.tag == FRAME:
                frames.append(ctx.render(abi.FrameDesc.from_buffer_copy(d["frame"].tobytes()))["rgba8"])
    finally:
        if own:
            ctx.close()
    return frames


if __name__ == "__main__":
    import sys

    imgs = replay(sys.argv[1])
    for i, im in enumerate(imgs):
        np.save(f"{sys.argv[1]}.frame{i}.npy", im)
    print(f"replayed {len(imgs)} frame(s)")
