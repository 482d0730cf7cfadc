"""Synthetic workloads of the benchmark and the parity tests: plain numpy generators of
`flat.FlatSpace` scenes. Product-side module (bench.py and __graft_entry__ import it; tests/scenes.py re-exports it):
it must not import anything from tests/ or oracle/."""
from __future__ import annotations

import numpy as np

from . import flat

#: palette::DAY_SKY_COLOR = srgb[243 243 255] decoded to linear (palette.rs:63; values of the reference's sRGB table)
DAY_SKY_LINEAR = (0.8962694406509399, 0.8962694406509399, 1.0)
#: palette::ALMOST_BLACK = srgb[0x3d 0x3d 0x3d] decoded (palette.rs:82)
ALMOST_BLACK_LINEAR = (0.046665072441101074,) * 3


# -- synthetic scenes for parity / bench (SURVEY.md 8d "S256") ------------------------------
def _splitmix64(state: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = state + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _hash3(x, y, z, seed) -> np.ndarray:
    with np.errstate(over="ignore"):
        h = (x.astype(np.uint64) * np.uint64(0x9E3779B1) ^ y.astype(np.uint64) * np.uint64(0x85EBCA77)
             ^ z.astype(np.uint64) * np.uint64(0xC2B2AE3D) ^ np.uint64(seed))
    return _splitmix64(h)


def synthetic_blocks(resolution: int, count: int, seed: int = 1, palette_size: int = 16, translucent: bool = True):
    """`count` distinct recursive blocks at `resolution`: spheres / slabs / lattices / 30% random fill."""
    r = resolution
    g = np.arange(r)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    blocks = []
    rng = np.random.default_rng(seed)
    for k in range(count):
        pal = np.zeros((palette_size, 8), np.float32)
        cols = rng.uniform(0.05, 1.0, (palette_size, 3)).astype(np.float32)
        pal[:, 0:3] = cols
        pal[:, 3] = 1.0
        pal[0] = 0.0  # index 0 = empty voxel
        if translucent and k % 4 == 3:
            pal[1:, 3] = np.float32(0.5)
        if k % 8 == 5:
            pal[1, 4:7] = (0.0, 0.8, 0.2)  # an emissive entry
        shape = k % 4
        hv = _hash3(X, Y, Z, seed * 1000 + k)
        colour = (1 + (hv >> np.uint64(40)) % np.uint64(palette_size - 1)).astype(np.uint16)
        if shape == 0:
            c = (r - 1) / 2.0
            mask = (X - c) ** 2 + (Y - c) ** 2 + (Z - c) ** 2 <= (0.48 * r) ** 2
        elif shape == 1:
            mask = Y < max(1, (r * (1 + k % 3)) // 4)
        elif shape == 2:
            q = max(1, r // 4)
            mask = ((X % q == 0) & (Y % q == 0)) | ((Y % q == 0) & (Z % q == 0)) | ((X % q == 0) & (Z % q == 0))
        else:
            mask = (hv % np.uint64(100)) < np.uint64(30)
        vox = np.where(mask, colour, 0).astype(np.uint16)
        blocks.append(flat.voxel_block(r, vox, pal, name=chr(ord("a") + k % 26)))
    return blocks


def synthetic_space(n: int = 256, resolution: int = 32, n_blocks: int = 64, seed: int = 1, light: str = "one") -> flat.FlatSpace:
    """S<n>: [0,n)^3, heightfield terrain <= n/2 high + 5% floating blocks (about 45% non-air)."""
    sp = flat.FlatSpace((0, 0, 0), (n, n, n))
    sp.set_sky_uniform((0.9, 0.9, 1.0))
    a = sp.add_block(flat.air())
    atoms = [sp.add_block(flat.atom(c)) for c in [(0.3, 0.6, 0.2, 1.0), (0.5, 0.45, 0.4, 1.0), (0.8, 0.75, 0.5, 1.0), (0.3, 0.5, 0.9, 0.5)]]
    recs = [sp.add_block(b) for b in synthetic_blocks(resolution, n_blocks, seed)]
    g = np.arange(n)
    # smooth heightfield from a few seeded sinusoids
    rng = np.random.default_rng(seed)
    H = np.zeros((n, n))
    for _ in range(6):
        fx, fz = rng.uniform(0.5, 4.0, 2) * 2 * np.pi / n
        ph = rng.uniform(0, 2 * np.pi, 2)
        H += rng.uniform(0.3, 1.0) * np.sin(g[:, None] * fx + ph[0]) * np.cos(g[None, :] * fz + ph[1])
    H = (H - H.min()) / (H.max() - H.min())
    height = (0.15 * n + 0.35 * n * H).astype(np.int64)  # <= n/2
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    hv = _hash3(X, Y, Z, seed)
    pick = (hv >> np.uint64(20)) % np.uint64(len(atoms) + len(recs))
    table = np.array(atoms + recs, dtype=np.uint16)
    solid = Y < height[:, None, :]
    # top two layers recursive-heavy, interior atoms
    surface_layer = Y >= (height[:, None, :] - 2)
    rec_pick = table[len(atoms) + ((hv >> np.uint64(33)) % np.uint64(len(recs))).astype(np.int64)]
    any_pick = table[pick.astype(np.int64)]
    grid = np.where(solid, np.where(surface_layer, rec_pick, any_pick), a)
    floating = (~solid) & ((hv % np.uint64(100)) < np.uint64(5)) & (Y < (3 * n) // 4)
    grid = np.where(floating, rec_pick, grid)
    sp.block_index[...] = grid.astype(np.uint16)
    if light == "field":
        nonair = sp.block_index != a
        lv = (120 + 40 * np.sin(X * 0.05) * np.cos(Z * 0.07) + 20 * (Y / n)).clip(1, 200).astype(np.uint8)
        sp.light[..., 0] = lv
        sp.light[..., 1] = lv
        sp.light[..., 2] = np.minimum(lv.astype(np.int64) + 6, 255).astype(np.uint8)
        sp.light[..., 3] = np.where(nonair, flat.STATUS_OPAQUE, flat.STATUS_VISIBLE)
        sp.light[nonair, 0:3] = 0
    return sp


def atrium_like_space(seed: int = 7) -> flat.FlatSpace:
    """S-atrium-like: 19x35x51 cubes, R16 blocks, an open hall with floors,
    arches and balconies; stands in for UniverseTemplate::Atrium (whose generator needs the
    un-vendored noise crate and the block-evaluation engine)."""
    lo = (-9, -1, -25)
    size = (19, 35, 51)
    sp = flat.FlatSpace(lo, size)
    sp.set_sky_uniform(DAY_SKY_LINEAR)  # DAY_SKY_COLOR palette.rs:63
    a = sp.add_block(flat.air())
    stone = sp.add_block(flat.atom((0.55, 0.53, 0.5, 1.0)))
    recs = [sp.add_block(b) for b in synthetic_blocks(16, 24, seed, translucent=True)]
    sx, sy, sz = size
    gx, gy, gz = np.arange(sx), np.arange(sy), np.arange(sz)
    X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
    hv = _hash3(X, Y, Z, seed)
    rec = np.array(recs, np.uint16)[((hv >> np.uint64(30)) % np.uint64(len(recs))).astype(np.int64)]
    grid = np.full(size, a, np.uint16)
    wall = (X == 0) | (X == sx - 1) | (Z == 0) | (Z == sz - 1)
    grid[wall] = stone
    grid[:, 0, :] = rec[:, 0, :]  # detailed floor
    for fy in (8, 16, 24):  # balconies along the walls
        ring = (Y == fy) & ((X < 4) | (X >= sx - 4) | (Z < 4) | (Z >= sz - 4))
        grid[ring] = rec[ring]
    pillars = ((X % 6 == 3) & (Z % 6 == 3)) & (Y < 25) & ((X < 5) | (X > sx - 6))
    grid[pillars] = rec[pillars]
    arches = (Y == 7) & (Z % 6 == 3) & (X > 3) & (X < sx - 4)
    grid[arches] = rec[arches]
    grid[:, sy - 1, :] = np.where((X[:, 0, :] + Z[:, 0, :]) % 3 == 0, a, stone)  # skylights
    sp.block_index[...] = grid
    # an evaluated-looking light field (the template is lit by its skylights): brighter towards the
    # roof, smooth elsewhere; cubes holding a block are STATUS_OPAQUE with no light of their own,
    # which is what exercises the ambient-occlusion weights and the light-leak rule of
    # get_interpolated_light (sr.rs:248-359)
    nonair = grid != a
    lv = (100 + 60 * (Y / sy) + 20 * np.sin(X * 0.4) * np.cos(Z * 0.3)).clip(1, 200).astype(np.uint8)
    sp.light[..., 0] = lv
    sp.light[..., 1] = lv
    sp.light[..., 2] = np.minimum(lv.astype(np.int64) + 4, 255).astype(np.uint8)
    sp.light[..., 3] = np.where(nonair, flat.STATUS_OPAQUE, flat.STATUS_VISIBLE)
    sp.light[nonair, 0:3] = 0
    return sp




# -- light_bench_space (all-is-cubes/src/content/testing.rs:26-141) --------------------------------------------
# The scene of the reference's only raytracer benchmark (all-is-cubes-render/benches/raytrace.rs:28-37: size 54x16x54,
# 64x64 viewport) and of the `template-light-bench` image test. Its sections are drawn with rand 0.10 / rand_xoshiro
# (Xoshiro256Plus::seed_from_u64, random_range, random_bool) -- crates that are not vendored with the reference; their
# published algorithms are restated below and pinned by the golden image template-light-bench-all.png
# (tests/test_oracle_goldens.py test_png_template_light_bench).

_M64 = (1 << 64) - 1


class _Xoshiro256Plus:
    def __init__(self, seed: int) -> None:  # seed_from_u64: SplitMix64 fills the state
        x = seed & _M64
        self.s = []
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & _M64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            self.s.append(z ^ (z >> 31))

    def next_u64(self) -> int:
        s = self.s
        result = (s[0] + s[3]) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & _M64
        return result

    def next_u32(self) -> int:
        return self.next_u64() >> 32

    def unit_f32_inclusive(self) -> np.float32:  # random_range(0.0..=1.0): [1,2) mantissa trick, scale 1, low 0
        bits = (self.next_u32() >> 9) | 0x3F800000
        return np.float32(np.array([bits], np.uint32).view(np.float32)[0] - np.float32(1.0))

    def random_bool(self, p: float) -> bool:  # Bernoulli: next_u64 < p * 2^64
        return self.next_u64() < int(p * 18446744073709551616.0)

    def below(self, n: int) -> int:  # random_range(0..n) for i32: widening multiply, one bias-reduction step
        m = self.next_u32() * n
        result, lo = m >> 32, m & 0xFFFFFFFF
        if lo > ((-n) & 0xFFFFFFFF):
            new_hi = (self.next_u32() * n) >> 32
            result += 1 if lo + new_hi > 0xFFFFFFFF else 0
        return result


def light_bench_layout(size=(54, 16, 54)):
    section_width, margin = 6, 4
    spacing = section_width + margin
    ax, az = min(255, (size[0] - margin) // spacing), min(255, (size[2] - margin) // spacing)
    height = min(255, size[1])
    section_height = max(0, height - 2)
    yup = section_height * 4 // 14
    ydown = section_height - yup
    lo = (0, -ydown - 1, 0)
    hi = (spacing * ax + margin, 1 + yup, spacing * az + margin)
    return dict(ax=ax, az=az, section_height=section_height, yup=yup, ydown=ydown, lo=lo, hi=hi, section_width=section_width,
                margin=margin, spacing=spacing)


def light_bench_space(size=(54, 16, 54)) -> flat.FlatSpace:
    """content::testing::light_bench_space(size), without light (LightPhysics::Rays { maximum_distance = max(width, depth) }
    is what the caller evaluates it with); the Spawn is looking_at_space(bounds, [0, 0.5, 1])."""
    L = light_bench_layout(size)
    lo, hi = L["lo"], L["hi"]
    sp = flat.FlatSpace(lo, tuple(h - l for l, h in zip(lo, hi)))
    ground_sky = np.float32(ALMOST_BLACK_LINEAR[0])
    day = np.array(DAY_SKY_LINEAR, np.float32)
    bright, dim = day * np.float32(2.0), day * np.float32(0.5)
    g = np.array([ground_sky] * 3, np.float32)
    # Sky::Octants index = x_pos << 2 | y_pos << 1 | z_pos (sky.rs:32-41)
    sp.set_sky_octants(np.stack([g, g, bright, bright, g, g, dim, dim]))
    air = sp.add_block(flat.air())
    ground = sp.add_block(flat.atom((0.5, 0.5, 0.5, 1.0)))

    def fill(lo3, hi3, index):
        x0, y0, z0 = (a - b for a, b in zip(lo3, lo))
        x1, y1, z1 = (a - b for a, b in zip(hi3, lo))
        sp.block_index[x0:x1, y0:y1, z0:z1] = index

    fill(lo, (hi[0], hi[1] - L["yup"], hi[2]), ground)  # bounds.shrink(PY by yup)
    for i in range(L["ax"] * L["az"]):
        sx, sz = divmod(i, L["az"])
        rng = _Xoshiro256Plus(sx + sz * L["ax"])
        s_lo = (L["margin"] + sx * L["spacing"], -L["ydown"] + 1, L["margin"] + sz * L["spacing"])
        s_hi = (s_lo[0] + L["section_width"], s_lo[1] + L["section_height"], s_lo[2] + L["section_width"])
        r, gch, b = rng.unit_f32_inclusive(), rng.unit_f32_inclusive(), rng.unit_f32_inclusive()
        alpha = 0.5 if rng.random_bool(0.125) else 1.0
        color = sp.add_block(flat.atom((float(r), float(gch), float(b), alpha)))
        kind = rng.below(3)
        if kind == 0:
            fill(s_lo, s_hi, color)
        elif kind == 1:
            fill(s_lo, (s_hi[0], s_hi[1] - L["yup"], s_hi[2]), color)
            fill((s_lo[0] + 1, s_lo[1], s_lo[2] + 1), (s_hi[0] - 1, s_hi[1], s_hi[2] - 1), air)
        else:
            for x in range(s_lo[0], s_hi[0]):       # GridAab::interior_iter: x, then y, then z fastest
                for y in range(s_lo[1], s_hi[1]):
                    for z in range(s_lo[2], s_hi[2]):
                        sp.set((x, y, z), color if rng.random_bool(0.25) else air)
    sp.light[...] = (0, 0, 0, flat.STATUS_UNINITIALIZED)
    return sp
