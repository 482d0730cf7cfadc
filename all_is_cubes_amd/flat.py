"""Flat (device-uploadable) description of a `Space` snapshot.

This is the host-side data model of the drop-in boundary: exactly the data that
`SpaceRaytracer::new` snapshots out of a `Space`
(reference: all-is-cubes-render/src/raytracer/sr.rs:64-88, 543-587):

* per cube: `u16` block index (Z-major `Vol`, all-is-cubes-base/src/math/vol.rs:988-1023)
  and `PackedLight` as the 4-byte texel `r,g,b,status`
  (all-is-cubes/src/space/light/data.rs:160-170);
* per block-palette entry: the evaluated block's `Evoxels` -- `One(Evoxel)` or paletted
  `u16` indices + `Evoxel` palette, whose voxel volume may be smaller than `resolution^3`
  (all-is-cubes/src/block/eval/voxel_storage.rs:199-227);
* the `Sky` (all-is-cubes/src/space/sky.rs:16-21).

The arrays produced by :meth:`FlatSpace.pack` are handed verbatim to the C ABI
(`include/aic_hip.h`, `aic_upload_space`); tests hand the same arrays to the CPU oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# 48-byte block descriptor shared by `aic_block_desc` (include/aic_hip.h) and `orc_block`.
BLOCK_DTYPE = np.dtype(
    [
        ("resolution", "<i4"),
        ("vlo", "<i4", (3,)),
        ("vsize", "<i4", (3,)),
        ("vox_off", "<u4"),
        ("pal_off", "<u4"),
        ("pal_len", "<u4"),
        ("flags", "<u4"),  # bit0: Evoxels::One, bit1: block == AIR
        ("name_char", "<i4"),
    ],
    align=False,
)
assert BLOCK_DTYPE.itemsize == 48

FLAG_ONE = 1
FLAG_AIR = 2

# LightStatus discriminants (light/data.rs:36-47)
STATUS_UNINITIALIZED = 0
STATUS_NO_RAYS = 1
STATUS_OPAQUE = 128
STATUS_VISIBLE = 255

#: PackedLight::ONE texel (light/data.rs:74-80): value byte 144 decodes to 1.0
LIGHT_ONE = (144, 144, 144, STATUS_VISIBLE)

VALID_RESOLUTIONS = (1, 2, 4, 8, 16, 32, 64, 128)  # resolution.rs:18-31


@dataclass
class BlockDef:
    """One evaluated block's voxels (`Evoxels`)."""

    resolution: int
    vlo: Tuple[int, int, int]
    voxels: np.ndarray  # uint16, shape (sx, sy, sz): palette indices, Z fastest
    palette: np.ndarray  # float32, shape (n, 8): r,g,b,a, er,eg,eb, 0
    is_one: bool = False
    is_air: bool = False
    name: str = "#"

    def __post_init__(self) -> None:
        if self.resolution not in VALID_RESOLUTIONS:
            raise ValueError(f"invalid resolution {self.resolution}")
        self.voxels = np.ascontiguousarray(self.voxels, dtype=np.uint16)
        self.palette = np.ascontiguousarray(self.palette, dtype=np.float32).reshape(-1, 8)
        if self.voxels.ndim != 3:
            raise ValueError("voxels must be 3-D")
        if self.voxels.size and int(self.voxels.max()) >= len(self.palette):
            raise ValueError("voxel index exceeds palette")
        lo = np.asarray(self.vlo, dtype=np.int64)
        hi = lo + np.asarray(self.voxels.shape, dtype=np.int64)
        if (lo < 0).any() or (hi > self.resolution).any():
            raise ValueError("voxel bounds exceed GridAab::for_block(resolution)")


def evoxel(rgba: Sequence[float], emission: Sequence[float] = (0.0, 0.0, 0.0)) -> np.ndarray:
    return np.array([*rgba, *emission, 0.0], dtype=np.float32)


def atom(rgba: Sequence[float], emission: Sequence[float] = (0.0, 0.0, 0.0), name: str = "#") -> BlockDef:
    """An R1 block stored as `Evoxels::One` (what `Block::from(color)` evaluates to)."""
    return BlockDef(1, (0, 0, 0), np.zeros((1, 1, 1), np.uint16), evoxel(rgba, emission)[None, :], is_one=True, name=name)


def air() -> BlockDef:
    """`AIR`: Evoxel::AIR (voxel_storage.rs:68-73), flagged so the cube test can short-cut."""
    b = atom((0.0, 0.0, 0.0, 0.0), name=" ")
    b.is_air = True
    return b


@dataclass
class PackedSpace:
    lo: np.ndarray
    size: np.ndarray
    block_index: np.ndarray
    light: np.ndarray
    always_invisible: np.ndarray
    blocks: np.ndarray
    voxels: np.ndarray
    palette: np.ndarray
    sky_kind: int
    sky: np.ndarray


@dataclass
class FlatSpace:
    lo: Tuple[int, int, int]
    size: Tuple[int, int, int]
    blocks: List[BlockDef] = field(default_factory=list)
    sky_kind: int = 0
    sky: np.ndarray = field(default_factory=lambda: np.zeros((8, 3), np.float32))
    block_index: Optional[np.ndarray] = None
    light: Optional[np.ndarray] = None

    def __post_init__(self) -> None:
        sx, sy, sz = (int(v) for v in self.size)
        if self.block_index is None:
            self.block_index = np.zeros((sx, sy, sz), np.uint16)
        if self.light is None:
            # LightPhysics::None => every cube reads PackedLight::ONE (space.rs:1241-1245)
            self.light = np.empty((sx, sy, sz, 4), np.uint8)
            self.light[...] = LIGHT_ONE
        self.sky = np.ascontiguousarray(self.sky, dtype=np.float32).reshape(8, 3)

    # -- construction helpers ----------------------------------------------------------
    def add_block(self, block: BlockDef) -> int:
        self.blocks.append(block)
        if len(self.blocks) > 65536:
            raise ValueError("too many blocks for u16 BlockIndex")
        return len(self.blocks) - 1

    def set_sky_uniform(self, rgb: Sequence[float]) -> None:
        self.sky_kind = 0
        self.sky[:] = 0
        self.sky[0] = rgb

    def set_sky_octants(self, colors: np.ndarray) -> None:
        self.sky_kind = 1
        self.sky[:] = np.asarray(colors, np.float32).reshape(8, 3)

    def set(self, cube: Sequence[int], index: int) -> None:
        x, y, z = (int(c) - int(l) for c, l in zip(cube, self.lo))
        self.block_index[x, y, z] = index

    @property
    def hi(self) -> Tuple[int, int, int]:
        return tuple(int(l) + int(s) for l, s in zip(self.lo, self.size))

    # -- flattening ------------------------------------------------------------------
    def pack(self) -> PackedSpace:
        n = len(self.blocks)
        table = np.zeros(n, BLOCK_DTYPE)
        vox_parts, pal_parts = [], []
        vox_off = pal_off = 0
        for i, b in enumerate(self.blocks):
            table[i]["resolution"] = b.resolution
            table[i]["vlo"] = b.vlo
            table[i]["vsize"] = b.voxels.shape
            table[i]["vox_off"] = vox_off
            table[i]["pal_off"] = pal_off
            table[i]["pal_len"] = len(b.palette)
            table[i]["flags"] = (FLAG_ONE if b.is_one else 0) | (FLAG_AIR if b.is_air else 0)
            table[i]["name_char"] = ord(b.name[0]) if b.name else ord("#")
            vox_parts.append(b.voxels.reshape(-1))
            pal_parts.append(b.palette)
            vox_off += b.voxels.size
            pal_off += len(b.palette)
        voxels = np.concatenate(vox_parts) if vox_parts else np.zeros(0, np.uint16)
        palette = np.concatenate(pal_parts) if pal_parts else np.zeros((0, 8), np.float32)
        air_flags = np.array([b.is_air for b in self.blocks], dtype=np.uint8)
        bi = np.ascontiguousarray(self.block_index, dtype=np.uint16)
        if bi.size and n and int(bi.max()) >= n:
            raise ValueError("block index out of range")
        always_invisible = air_flags[bi] if n else np.zeros(bi.shape, np.uint8)
        return PackedSpace(
            lo=np.asarray(self.lo, np.int32),
            size=np.asarray(self.size, np.int32),
            block_index=bi,
            light=np.ascontiguousarray(self.light, dtype=np.uint8),
            always_invisible=np.ascontiguousarray(always_invisible, dtype=np.uint8),
            blocks=table,
            voxels=np.ascontiguousarray(voxels, dtype=np.uint16),
            palette=np.ascontiguousarray(palette, dtype=np.float32),
            sky_kind=int(self.sky_kind),
            sky=np.ascontiguousarray(self.sky, dtype=np.float32),
        )


def pack_block(b: BlockDef):
    """(descriptor, voxels, palette) of one block as `aic_replace_block` takes them (offsets unused)."""
    d = np.zeros(1, BLOCK_DTYPE)
    d["resolution"], d["vlo"], d["vsize"] = b.resolution, b.vlo, b.voxels.shape
    d["pal_len"] = len(b.palette)
    d["flags"] = (FLAG_ONE if b.is_one else 0) | (FLAG_AIR if b.is_air else 0)
    d["name_char"] = ord(b.name[0]) if b.name else ord("#")
    return d, np.ascontiguousarray(b.voxels, np.uint16).reshape(-1), np.ascontiguousarray(b.palette, np.float32)


def voxel_block(
    resolution: int,
    voxel_palette_index: np.ndarray,
    palette: np.ndarray,
    vlo: Sequence[int] = (0, 0, 0),
    name: str = "#",
) -> BlockDef:
    """A recursive block from a dense (sx,sy,sz) array of palette indices."""
    return BlockDef(resolution, tuple(int(v) for v in vlo), voxel_palette_index, palette, name=name)
