"""VERDICT r03 next 5 (time-boxed diagnostic of the fog-* golden hole): WHERE on the lamp-lit wall and pillars does the oracle's image of
fog_test_universe differ from fog-None-ray.png, cube by cube, and how does that relate to the lamps?

For every pixel whose first hit is a wall (x = 29) or pillar cube, the signed red difference ours - golden in sRGB levels is
accumulated on that cube (the first-hit cube comes from the oracle's aux records; the geometry is exact: debug_pixel_cost-ray, same
universe, matches pixel for pixel). Printed: the mean per cube as a (y, z) map of the wall, the mean by Chebyshev distance from the
cube to the nearest lamp, and the same for the pillars by height.

    python tools/fog_texel_map.py > profiles/r04_fog_texel_map.txt        (CPU only)
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402
from fog_experiments import GOLDEN, fog_space  # noqa: E402
from test_oracle_goldens import COMMON_VIEWPORT  # noqa: E402
from test_oracle_light import spawn_camera  # noqa: E402


def main():
    sp = fog_space()
    oracle.evaluate_light(sp, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16)
    cam = spawn_camera(COMMON_VIEWPORT, (0.0, 10.0, 0.0), (0.4, 0.0, -1.0), view_distance=50.0)
    r = oracle.render(oracle.Space(sp), oracle.unaltered_colors(lighting=3, fog=0, view_distance=50.0), cam, threads=os.cpu_count() or 4, want_aux=True)
    gold = np.load(os.path.join(GOLDEN, "png_fog-None-ray.npy")).astype(int)
    d = r["rgba8"].astype(int) - gold
    aux = r["aux"]
    z_length = 60
    lamps = np.array([((z * 19) % 60 - 30, 8, z + 1) for z in range(-z_length, 0, 2)])
    print("# signed difference ours - golden (sRGB levels), red channel, by first-hit cube; fog-None-ray.png; lamps at", len(lamps), "cubes (x, 8, z+1)")
    for name, block in (("wall x = 29 (1, .5, .5)", 2), ("pillars (ALMOST_BLACK)", 3), ("floor (0, 1, .5)", 1)):
        m = (aux["block_index"] == block) & (aux["hit"] == 1)
        cubes = aux["cube"][m]
        dr, dg, db = d[m][:, 0], d[m][:, 1], d[m][:, 2]
        gr = gold[m][:, 0]
        # per cube
        keys, inv = np.unique(cubes, axis=0, return_inverse=True)
        inv = inv.ravel()
        n = np.bincount(inv)
        mean_r = np.bincount(inv, dr) / n
        mean_g = np.bincount(inv, dg) / n
        dist = np.abs(keys[:, None, :] - lamps[None, :, :]).max(axis=2).min(axis=1)
        print(f"\n## {name}: {m.sum()} pixels on {len(keys)} cubes; mean signed rgb {dr.mean():+.2f} {dg.mean():+.2f} {db.mean():+.2f}; golden red level mean {gr.mean():.0f}")
        print("   by Chebyshev distance cube -> nearest lamp:  distance: cubes, pixels, mean red diff, mean green diff, mean golden red level")
        gr_c = np.bincount(inv, gr) / n
        for dd in sorted(set(dist)):
            k = dist == dd
            px = n[k].sum()
            print(f"     {dd:3d}: {k.sum():4d} cubes {px:5d} px   red {np.average(mean_r[k], weights=n[k]):+.2f}   green {np.average(mean_g[k], weights=n[k]):+.2f}   golden red {np.average(gr_c[k], weights=n[k]):6.1f}")
        if block == 2:
            print("   map of the wall, rows y = 19 .. 1, columns z = -1 .. -40 (mean red diff per cube x 10, '..' = not seen):")
            grid = {(int(c[1]), int(c[2])): v for c, v in zip(keys, mean_r)}
            for y in range(19, 0, -1):
                print("     y%2d " % y + " ".join(("%+3d" % round(10 * grid[(y, z)])) if (y, z) in grid else " .." for z in range(-1, -41, -1)))
    # relative error in linear light: is it a constant factor (one PackedLight unit is 2^(1/10) = +7.2 %)?
    m = (aux["block_index"] == 2) & (aux["hit"] == 1)

    def lin(v):
        v = v / 255.0
        return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    ours, g = lin(r["rgba8"][m][:, 0].astype(float)), lin(gold[m][:, 0].astype(float))
    ok = g > 0.02
    ratio = ours[ok] / g[ok]
    print(f"\n## wall, linear red ours / golden: median {np.median(ratio):.4f}, mean {ratio.mean():.4f}, 10th..90th percentile {np.percentile(ratio, 10):.4f} .. {np.percentile(ratio, 90):.4f}"
          f"  (one PackedLight unit = {2 ** 0.1:.4f})")


if __name__ == "__main__":
    main()


def hysteresis():
    """Is the converged light near a lamp a unique fixed point? A cube under a lamp sees the lamp's face lit by ITS OWN stored light
    (updater.rs:812-826: light_from_struck_face = emission + colour.reflect(stored light of the cube in front of the face) -- the cube
    itself), so its 8-bit log-scale texel feeds back into its own next value. For the cubes next to every lamp: lower / raise the
    converged red texel by one unit, recompute the cube, and see whether it returns (unique fixed point) or stays (two stable states:
    which one a run ends in depends on the side it is approached from, i.e. on the whole update history)."""
    import copy
    sp = fog_space()
    oracle.evaluate_light(sp, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16)
    lo = np.array(sp.lo)
    z_length = 60
    lamps = [((z * 19) % 60 - 30, 8, z + 1) for z in range(-z_length, 0, 2)]
    print("\n## hysteresis of the texels next to a lamp (red channel; converged value v; recomputed after setting it to v-1 / v+1)")
    stay_lo = stay_hi = n = 0
    rows = []
    for lamp in lamps:
        for off in ((0, -1, 0), (0, 1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, -2, 0), (0, 2, 0)):
            c = tuple(int(a + b) for a, b in zip(lamp, off))
            i = tuple(int(a - b) for a, b in zip(c, lo))
            if not all(0 <= i[k] < sp.size[k] for k in range(3)) or int(sp.block_index[i]) != 0:
                continue
            v = int(sp.light[i][0])
            res = []
            for dv in (-1, +1):
                t = copy.deepcopy(sp)
                t.light[i][0] = max(0, min(255, v + dv))
                out, _ = oracle.compute_light(oracle.Space(t), c, 30)
                res.append(int(out[0]))
            same, _ = oracle.compute_light(oracle.Space(sp), c, 30)
            n += 1
            stay_lo += res[0] == v - 1
            stay_hi += res[1] == v + 1
            rows.append((off, v, int(same[0]), res[0], res[1]))
    by_off = {}
    for off, v, same, lo_, hi_ in rows:
        by_off.setdefault(off, []).append((v, same, lo_, hi_))
    for off, lst in by_off.items():
        a = np.array(lst)
        print(f"   offset {off}: {len(lst):2d} cubes, converged red {a[:, 0].min()}..{a[:, 0].max()}; recomputed as is: unchanged {int((a[:, 1] == a[:, 0]).sum())}; "
              f"from v-1: stays at v-1 {int((a[:, 2] == a[:, 0] - 1).sum())}, returns to v {int((a[:, 2] == a[:, 0]).sum())}; "
              f"from v+1: stays at v+1 {int((a[:, 3] == a[:, 0] + 1).sum())}, returns to v {int((a[:, 3] == a[:, 0]).sum())}")
    print(f"   total {n} texels: {stay_lo} have a second stable state one unit lower, {stay_hi} one unit higher")


if __name__ == "__main__" and "--hysteresis" in sys.argv:
    hysteresis()
