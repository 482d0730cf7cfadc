"""Scene restatements shared by the oracle tests and the GPU parity tests.

Each builder cites the reference test whose scene it restates. All return
`all_is_cubes_amd.flat.FlatSpace` objects (plain numpy data)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from all_is_cubes_amd import flat

GOLDEN = Path(__file__).resolve().parent / "golden"


def srgb_lut() -> np.ndarray:
    return np.load(GOLDEN / "srgb_decode_lut.npy")


def from_srgb8(rgb) -> tuple:
    """Rgb::from_srgb8 / Rgb01::from_srgb8 via the reference's decode table (color.rs:301-308)."""
    lut = srgb_lut()
    return tuple(float(lut[c]) for c in rgb)


# -- all-is-cubes-render/src/raytracer/surface.rs:541-570 ---------------------------------
def surface_iter_basic_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 3, 1))
    a = sp.add_block(flat.air())
    solid = sp.add_block(flat.atom((1.0, 0.0, 0.0, 1.0)))
    # slab_with_extra_space: R4; `cube.y >= 2 && cube.x != 0` is AIR, else the slab colour
    vox = np.zeros((4, 4, 4), np.uint16)
    for x in range(4):
        for y in range(4):
            for z in range(4):
                vox[x, y, z] = 0 if (y >= 2 and x != 0) else 1
    pal = np.stack([flat.evoxel((0, 0, 0, 0)), flat.evoxel((1.0, 1.0, 0.0, 1.0))])
    slab = sp.add_block(flat.voxel_block(4, vox, pal))
    sp.block_index[...] = a
    sp.set((0, 1, 0), solid)
    sp.set((0, 2, 0), slab)
    return sp


# surface.rs:682-689
def one_red_cube_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.block_index[...] = sp.add_block(flat.atom((1.0, 0.0, 0.0, 1.0)))
    return sp


# all-is-cubes/src/content.rs:165-211 make_slab(universe, numerator, denominator): voxel
# volume is [denominator, numerator, denominator] in a checkerboard of two opaque colours.
def slab_block(numerator: int, denominator: int, name: str = "S") -> flat.BlockDef:
    vox = np.zeros((denominator, numerator, denominator), np.uint16)
    for x in range(denominator):
        for y in range(numerator):
            for z in range(denominator):
                vox[x, y, z] = (x + y + z) % 2
    pal = np.stack([flat.evoxel((0.6, 0.4, 0.2, 1.0)), flat.evoxel((0.636, 0.424, 0.212, 1.0))])
    return flat.voxel_block(denominator, vox, pal, name=name)


# surface.rs:713-721, accum.rs:442-450
def slab_cube_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.block_index[...] = sp.add_block(slab_block(1, 2))
    return sp


# surface.rs:755-777
def half_transparent_slab_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 3, 1))
    a = sp.add_block(flat.air())
    # voxels_fn(R2): y > 0 is AIR, else colour (1,1,0,0.5); bounds auto-shrink to y in [0,1)
    vox = np.ones((2, 1, 2), np.uint16)
    pal = np.stack([flat.evoxel((0, 0, 0, 0)), flat.evoxel((1.0, 1.0, 0.0, 0.5))])
    slab = sp.add_block(flat.voxel_block(2, vox, pal))
    sp.block_index[...] = a
    sp.set((0, 1, 0), slab)
    return sp


# raytracer/text.rs:195-208 (make_some_blocks names "0","1","2"; colours are opaque)
def print_space_test_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (3, 1, 1))
    for i in range(3):
        sp.set((i, 0, 0), sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0), name=str(i))))
    return sp


# text.rs:262-289
def partial_voxels_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (2, 1, 1))
    b0 = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0), name="0"))
    vox = np.zeros((4, 2, 4), np.uint16)
    pal = flat.evoxel((1.0, 1.0, 1.0, 1.0))[None, :]
    p = sp.add_block(flat.voxel_block(4, vox, pal, name="P"))
    sp.set((0, 0, 0), b0)
    sp.set((1, 0, 0), p)
    return sp


# -- test-renderers/cases/src/lib.rs --------------------------------------------------------
# one_cube_space() 1239-1248: sky (0.5,0.5,0.5), filled with opaque green unless replaced
def one_cube_space(block: flat.BlockDef | None = None) -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.set_sky_uniform((0.5, 0.5, 0.5))
    sp.block_index[...] = sp.add_block(block if block is not None else flat.atom((0.0, 1.0, 0.0, 1.0)))
    return sp


# transparent_one 1138-1147
def transparent_one_space() -> flat.FlatSpace:
    return one_cube_space(flat.atom((1.0, 0.0, 0.0, 0.5)))


# emission 297-334
def emission_space() -> flat.FlatSpace:
    e_refl = flat.evoxel((*from_srgb8((200, 0, 0)), 1.0), from_srgb8((0, 200, 0)))
    e_only = flat.evoxel((0.0, 0.0, 0.0, 1.0), from_srgb8((0, 200, 0)))
    white = flat.evoxel((1.0, 1.0, 1.0, 1.0))
    pal = np.stack([white, e_refl, e_only])
    # Vol::from_y_flipped_array([[ "....", ".E..", "..e.", "...." ]]): rows are listed top (y=3) first,
    # columns are x; the same pattern for every z (p.z = 0)
    rows = ["....", ".E..", "..e.", "...."]
    vox = np.zeros((4, 4, 4), np.uint16)
    for row_i, row in enumerate(rows):
        y = 3 - row_i
        for x, ch in enumerate(row):
            vox[x, y, :] = {".": 0, "E": 1, "e": 2}[ch]
    return one_cube_space(flat.voxel_block(4, vox, pal))


# voxel_shape_test 371-418 with the atoms of emission_only (351-357) / emission_semi (360-367)
def voxel_shape_space(kind: str) -> flat.FlatSpace:
    if kind == "only":
        ev = flat.evoxel((0.0, 0.0, 0.0, 0.0), from_srgb8((0, 200, 0)))
    elif kind == "semi":
        ev = flat.evoxel((0.0, 0.0, 0.0, 1.0 - 2.0 ** -3), from_srgb8((0, 200, 0)))
    else:
        raise ValueError(kind)
    sp = flat.FlatSpace((-1, 0, 0), (4, 1, 1))
    sp.set_sky_uniform(from_srgb8((0, 0, 127)))
    a = sp.add_block(flat.air())
    atom_block = sp.add_block(flat.BlockDef(1, (0, 0, 0), np.zeros((1, 1, 1), np.uint16), ev[None, :], is_one=True))
    vox = np.zeros((2, 2, 2), np.uint16)
    for x in range(2):
        for y in range(2):
            for z in range(2):
                vox[x, y, z] = 1 if (x == 0 or y == 0 or z == 0) else 0
    pal = np.stack([flat.evoxel((0, 0, 0, 0)), ev])
    vb = sp.add_block(flat.voxel_block(2, vox, pal))
    sp.block_index[...] = a
    sp.set((-1, 0, 0), atom_block)
    sp.set((1, 0, 0), vb)
    return sp


# color_srgb_ramp 205-233
def color_srgb_ramp_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (32, 32, 1))
    a = sp.add_block(flat.air())
    sp.block_index[...] = a
    lut = srgb_lut()
    cache = {}

    def blk(rgb):
        if rgb not in cache:
            cache[rgb] = sp.add_block(flat.atom((float(lut[rgb[0]]), float(lut[rgb[1]]), float(lut[rgb[2]]), 1.0)))
        return cache[rgb]

    for i in range(256):
        px, py = (i % 16) * 2, (i // 16) * 2
        sp.set((px, py, 0), blk((i, i, i)))
        sp.set((px + 1, py, 0), blk((i, 0, 0)))
        sp.set((px + 1, py + 1, 0), blk((0, i, 0)))
        sp.set((px, py + 1, 0), blk((0, 0, i)))
    return sp


# ui_space 1260-1268
def ui_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((-3, -3, -4), (1, 1, 1))
    sp.set_sky_uniform((1.0, 1.0, 0.5))
    sp.block_index[...] = sp.add_block(flat.atom((0.0, 1.0, 0.0, 1.0)))
    return sp


# -- synthetic scenes for parity / bench (SURVEY.md 8d "S256"): product-side generators, re-exported ----------
from all_is_cubes_amd.workloads import (  # noqa: E402,F401
    atrium_like_space, light_bench_layout, light_bench_space, synthetic_blocks, synthetic_space,
)


# ---------------------------------------------------------------------------------------------
# Scenes of the reference's lighting image tests (test-renderers/cases/src/lib.rs:1354-1610). They carry no
# light: tests light them with the oracle's restatement of fast_evaluate_light + evaluate_light(1).

def _fill(sp: flat.FlatSpace, lo, size, index: int) -> None:
    x0, y0, z0 = (int(l) - int(b) for l, b in zip(lo, sp.lo))
    sp.block_index[x0 : x0 + size[0], y0 : y0 + size[1], z0 : z0 + size[2]] = index


def _unlit(sp: flat.FlatSpace) -> flat.FlatSpace:
    sp.light[...] = (0, 0, 0, flat.STATUS_UNINITIALIZED)
    return sp


DAY_SKY_COLOR = (243, 243, 255)  # palette.rs:63, the default Sky::Uniform (space/physics.rs DEFAULT)
ALMOST_BLACK = (0x3D, 0x3D, 0x3D)  # palette.rs:82
PLANK = (0xE8, 0xCC, 0x95)  # palette.rs:100


# fog_test_universe 1354-1407: eye (0,10,0) looking along (0.4, 0, -1)
def fog_test_space() -> flat.FlatSpace:
    z_length = 60
    sp = flat.FlatSpace((-30, 0, -z_length), (60, 20, z_length))
    sp.set_sky_uniform(from_srgb8(DAY_SKY_COLOR))
    sp.add_block(flat.air())
    floor = sp.add_block(flat.atom((0.0, 1.0, 0.5, 1.0)))
    wall = sp.add_block(flat.atom((1.0, 0.5, 0.5, 1.0)))
    pillar = sp.add_block(flat.atom((*from_srgb8(ALMOST_BLACK), 1.0)))
    lamp = sp.add_block(flat.atom((1.0, 0.05, 0.05, 1.0), (40.0, 0.05, 0.05)))
    _fill(sp, (-30, 0, -z_length), (60, 1, z_length), floor)   # bounds.abut(NY, -1)
    _fill(sp, (29, 0, -z_length), (1, 20, z_length), wall)     # bounds.abut(PX, -1)
    for z in range(-z_length, 0, 2):
        x = (z * 19) % 60 + (-30)
        _fill(sp, (x, 1, z), (1, 10, 1), pillar)
        sp.set((x, 8, z + 1), lamp)
    return _unlit(sp)


# light_spread_test_universe 1409-1444: eye (0,0,8) looking along -Z, fov 45
def light_spread_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((-10, -10, -1), (20, 20, 5))
    sp.set_sky_uniform(from_srgb8(DAY_SKY_COLOR))
    sp.add_block(flat.air())
    wall = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))
    pillar = sp.add_block(flat.atom((*from_srgb8(ALMOST_BLACK), 1.0)))
    source = sp.add_block(flat.atom((1.0, 0.05, 0.05, 1.0), (10.0, 5.0, 0.0)))
    _fill(sp, (-10, -10, -1), (20, 20, 1), wall)  # bounds.abut(NZ, -1)
    sp.set((-2, 2, 0), source)
    sp.set((-3, -1, 1), source)
    for i in range(-4, 5):
        sp.set((i, i, 0), pillar)
    return _unlit(sp)


def rotated_slab_block(height: int) -> flat.BlockDef:
    """content::make_slab(height, R16) (content.rs:176-211: checkerboard of PLANK and PLANK * 1.06) rotated by
    GridRotation::RXZy (rotation.rs to_basis: x -> +X, y -> +Z, z -> -Y; positive-octant transform, so
    new = (x, 15 - z, y))."""
    r = 16
    plank = np.array(from_srgb8(PLANK), np.float32)
    light = np.minimum(plank * np.float32(1.06), np.float32(1.0))  # Rgb01::saturating_scale (color.rs:512-518)
    pal = np.stack([flat.evoxel((*plank, 1.0)), flat.evoxel((*light, 1.0))])
    vox = np.zeros((r, r, height), np.uint16)  # rotated volume: x in [0,16), y in [0,16), z in [0,height)
    for x in range(r):
        for y in range(height):
            for z in range(r):
                vox[x, r - 1 - z, y] = (x + y + z) % 2
    return flat.voxel_block(r, vox, pal, name="S")


# light_on_slab_test_universe 1455-1500: eye (0.5,-6,6) looking along (0,1,-1), fov 45
def light_on_slab_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((-10, -10, -1), (20, 20, 5))
    sp.set_sky_uniform(from_srgb8(DAY_SKY_COLOR))
    sp.add_block(flat.air())
    wall = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))
    _fill(sp, (-10, -10, -1), (20, 20, 1), wall)
    for height in range(1, 17):
        position = height - 1
        cube = (-3 + (position % 4) * 2, -3 + (position // 4) * 2, 0)
        sp.set(cube, sp.add_block(rotated_slab_block(height)))
    return _unlit(sp)


TONE_MAP_LUMINANCE_RAMP = [1 / 64, 1 / 32, 1 / 16, 1 / 4, 1.0, 4.0, 16.0, 32.0, 64.0, 128.0]


# tone_mapping_test_universe 1503-1597: eye = bounds.center() + (0,0,65) looking along -Z, fov 45
def tone_mapping_space() -> flat.FlatSpace:
    low = 0.25
    colors = [(1, 0, 0), (1, low, 0), (1, 1, 0), (low, 1, 0), (0, 1, 0), (0, 1, low), (0, 1, 1), (0, low, 1), (0, 0, 1), (low, 0, 1),
              (1, 0, 1), (1, 0, low), (1, 1, 1)]
    xs, ys = 4, 4
    size = (len(TONE_MAP_LUMINANCE_RAMP) * xs + 1, len(colors) * ys + 1, 3)
    sp = flat.FlatSpace((-1, -1, -1), size)
    sp.set_sky_uniform((0.0, 0.0, 0.0))
    black = sp.add_block(flat.atom((*from_srgb8(ALMOST_BLACK), 1.0)))  # filled_with
    sp.block_index[...] = black
    wall = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))
    air = sp.add_block(flat.air())
    _fill(sp, (-1, -1, -1), (size[0], size[1], 1), wall)   # abut(NZ, -1)
    _fill(sp, (-1, -1, 1), (size[0], size[1], 1), air)     # abut(PZ, -1)
    for i, luminance in enumerate(TONE_MAP_LUMINANCE_RAMP):
        x = i * xs
        for j, color in enumerate(colors):
            y = j * ys
            em = tuple(np.float32(c) * np.float32(luminance) for c in color)
            source = sp.add_block(flat.atom((1.0, 1.0, 1.0, 1.0), em))
            _fill(sp, (x, y, 0), (xs - 1, ys - 1, 1), air)
            sp.set((x + 1, y, 0), source)
    return _unlit(sp)


# -- second batch of lighting / template cases ---------------------------------------------------------------------------

NEWLY_VISIBLE = 250  # space/light/queue.rs Priority::NEWLY_VISIBLE


def builder_light_after_sets(sp: flat.FlatSpace, sets, opaque_for_light):
    """The light state of a space that was built empty (all AIR: every texel NO_RAYS, empty queue -- LightPhysics::
    initialize_light, updater.rs:643-651) and then had `sets` = [(cube, block index)] applied with Mutation::set: what
    modified_cube_needs_update (updater.rs:135-173) does per set. `opaque_for_light(block index)` = all faces opaque and no
    emission. Writes sp.block_index / sp.light and returns the queue as [(cube, priority)] in insertion order."""
    sp.light[...] = (0, 0, 0, 1)  # PackedLight::NO_RAYS
    queue = []
    lo, size = sp.lo, sp.size

    def inb(c):
        return all(lo[a] <= c[a] < lo[a] + size[a] for a in range(3))

    def all_opaque(c):  # uc.get_evaluated(neighbor).opaque()[face]: only uniform blocks in these scenes
        return inb(c) and opaque_face[int(sp.block_index[tuple(c[a] - lo[a] for a in range(3))])]

    opaque_face = {i: bool(b.is_one and b.palette[0][3] >= 1.0) for i, b in enumerate(sp.blocks)}
    for cube, index in sets:
        sp.set(cube, index)
        rel = tuple(cube[a] - lo[a] for a in range(3))
        if opaque_for_light(index):
            sp.light[rel] = (0, 0, 0, 128)  # PackedLight::OPAQUE
            queue = [(c, p) for c, p in queue if c != tuple(cube)]
        else:
            queue.append((tuple(cube), NEWLY_VISIBLE))
        for f in range(6):
            nb = list(cube)
            nb[f % 3] += 1 if f >= 3 else -1
            if inb(nb) and not all_opaque(nb):
                queue.append((tuple(nb), NEWLY_VISIBLE))
    return queue


# furnace 620-664: a "white furnace": three white blocks (opaque or alpha 0.5) in a 3x3x3 space under a uniform 0.75 sky;
# m.evaluate_light(0) after the sets
def furnace_space(transparent: bool):
    sp = flat.FlatSpace((-1, -1, -1), (3, 3, 3))
    sp.set_sky_uniform((0.75, 0.75, 0.75))
    sp.add_block(flat.air())
    white = sp.add_block(flat.atom((1.0, 1.0, 1.0, 0.5 if transparent else 1.0)))
    queue = builder_light_after_sets(sp, [((-1, -1, 1), white), ((1, -1, 0), white), ((-1, 1, -1), white)],
                                     lambda i: i == white and not transparent)
    return sp, queue


# bloom_test_universe 1332-1351: one black cube emitting (0.5, 100, 0), LightPhysics::None, black sky; eye (1.5, 3, 8)
def bloom_test_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    sp.set_sky_uniform((0.0, 0.0, 0.0))
    sp.block_index[...] = sp.add_block(flat.atom((0.0, 0.0, 0.0, 1.0), (0.5, 100.0, 0.0)))
    return sp


# follow_options_change 560-603: green opaque cube and a blue alpha-0.5 cube, set into an empty space, light never evaluated
def follow_options_space() -> flat.FlatSpace:
    sp = flat.FlatSpace((-1, 0, 0), (3, 1, 1))
    sp.set_sky_uniform((0.5, 0.5, 0.5))
    sp.add_block(flat.air())
    green = sp.add_block(flat.atom((0.0, 1.0, 0.0, 1.0)))
    blue = sp.add_block(flat.atom((0.0, 0.0, 1.0, 0.5)))
    builder_light_after_sets(sp, [((0, 0, 0), green), ((1, 0, 0), blue)], lambda i: i == green)
    return sp


# all-is-cubes-content/src/template.rs:395-460 cornell_box at the default size 30: box_size 28
def cornell_box_space() -> flat.FlatSpace:
    box = 28
    sp = flat.FlatSpace((-1, -1, -1), (box + 2, box + 2, box + 2))
    sp.set_sky_uniform((0.0, 0.0, 0.0))
    sp.add_block(flat.air())
    white = sp.add_block(flat.atom((1.0, 1.0, 1.0, 1.0)))
    red = sp.add_block(flat.atom((0.57, 0.025, 0.025, 1.0)))
    green = sp.add_block(flat.atom((0.025, 0.236, 0.025, 1.0)))
    e = float(np.float32(1.07) * np.sqrt(np.float32(box)))
    light = sp.add_block(flat.atom((1.0, 1.0, 1.0, 1.0), (e, e, e), name="Light"))

    def scaled(lo, hi):  # GridAab::multiply(box).divide(55): lower bounds floor, upper bounds ceil (grid_aab.rs)
        lo2 = [(v * box) // 55 for v in lo]
        hi2 = [(v * box + 54) // 55 for v in hi]
        return lo2, [h - l for l, h in zip(lo2, hi2)]

    _fill(sp, (0, -1, 0), (box, 1, box), white)      # floor
    _fill(sp, (0, box, 0), (box, 1, box), white)     # ceiling
    llo, lsz = scaled((21, 55, 23), (34, 55, 33))    # light in the ceiling: .abut(PY, 1)
    _fill(sp, (llo[0], llo[1] + lsz[1], llo[2]), (lsz[0], 1, lsz[2]), light)
    _fill(sp, (0, 0, -1), (box, box, 1), white)      # back wall
    _fill(sp, (box, 0, 0), (1, box, box), green)     # right wall
    _fill(sp, (-1, 0, 0), (1, box, box), red)        # left wall
    for lo, size in (((29, 0, 36), (16, 16, 15)), ((10, 0, 13), (18, 33, 15))):
        blo, bsz = scaled(lo, tuple(l + s for l, s in zip(lo, size)))
        _fill(sp, tuple(blo), tuple(bsz), white)
    return sp


# antialias_test_universe (cases/src/lib.rs:1271-1329): a floor of R2 checker voxel blocks and a wall of solid blocks, no
# light. One floor block in nine is `make_some_voxel_blocks`' labelled block (an R16 block with a composited text glyph: needs
# the text engine and its font); it is a full opaque cube, so a plain opaque stand-in occludes exactly as it does and only the
# pixels that show it differ. Returns (space, index of the stand-in block).
def antialias_test_space():
    lo, size = (-5, -2, -60), (10, 10, 60)
    sp = flat.FlatSpace(lo, size)
    sp.set_sky_uniform(from_srgb8(DAY_SKY_COLOR))
    sp.add_block(flat.air())
    neutral = sp.add_block(flat.atom((1.0, 1.0, 1.0, 1.0)))
    large = sp.add_block(flat.atom((1.0, 0.0, 0.0, 1.0)))
    g = np.arange(2)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    pal = np.stack([flat.evoxel((0.5, 0.0, 1.0, 1.0)), flat.evoxel((1.0, 1.0, 1.0, 1.0))])
    vox = np.where((X + Y + Z) % 2 == 0, 0, 1).astype(np.uint16)
    voxel_block_1 = sp.add_block(flat.voxel_block(2, vox, pal))
    stand_in = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))
    for x in range(lo[0], lo[0] + size[0]):
        for z in range(lo[2], lo[2] + size[2]):
            sp.set((x, lo[1], z), stand_in if (x % 3 == 0 and z % 3 == 2) else voxel_block_1)   # floor: abut(NY, -1)
    x = lo[0] + size[0] - 1                                                                       # wall: abut(PX, -1)
    for y in range(lo[1], lo[1] + size[1]):
        for z in range(lo[2], lo[2] + size[2]):
            sp.set((x, y, z), large if (x + y + z) % 2 == 0 else neutral)
    return sp, stand_in


# sky (cases/src/lib.rs:1007-1051): one cube under an axis-coloured octant sky, seen from the side opposite `face`. The cube
# is `make_some_voxel_blocks`' labelled block (text engine): a full opaque stand-in occludes as it does; the sky pixels are
# what the case is about. Returns (space, eye, look direction).
def sky_test_space(face: str):
    sp = flat.FlatSpace((0, 0, 0), (1, 1, 1))
    r = np.array([*from_srgb8((0x9E, 0x00, 0x00))], np.float32)   # Rgb01::UNIFORM_LUMINANCE_RED / GREEN / BLUE (color.rs:357-364)
    g = np.array([*from_srgb8((0x00, 0x59, 0x00))], np.float32)
    b = np.array([*from_srgb8((0x00, 0x00, 0xFF))], np.float32)
    zero = np.zeros(3, np.float32)
    sp.set_sky_octants(np.stack([zero, b, g, g + b, r, r + b, r + g, r + g + b]))
    sp.block_index[...] = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))
    sp.light[...] = (0, 0, 0, 128)  # a space filled with an opaque block starts all PackedLight::OPAQUE
    axis = "XYZ".index(face[1])
    sign = 1.0 if face[0] == "P" else -1.0
    eye = np.array([0.5, 0.5, 0.5])
    eye[axis] -= sign * 2.0          # on the side opposite the sky face looked at, 1.5 from the cube's surface
    if axis == 1:
        eye[2] -= 0.25               # "tilt the view a little"
    else:
        eye[1] += 0.25
    look = np.array([0.5, 0.5, 0.5]) - eye
    return sp, tuple(float(v) for v in eye), tuple(float(v) for v in look)
