"""VERDICT r02 "next" 1: what moves the four fog-* goldens? Runs the oracle's light updater on fog_test_universe
(test-renderers/cases/src/lib.rs:1354-1406) under every variant of update order that could differ from the reference build that
made the goldens -- hashbrown group width 8 / 16, batches of 1 / 32 (auto-threads off / on), epsilon 0 / 1, and the queue tables
grown by the scene's build-time history before fast_evaluate_light clears them (queue.rs:287-298 keeps their capacity) -- plus a
sweep of the lamp emission, and prints each image's difference histogram against fog-None-ray.png.

    python tools/fog_experiments.py > profiles/r03_fog_experiments.txt        (CPU only, about two minutes)
"""
import ctypes as C
import itertools
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import scenes  # noqa: E402
from all_is_cubes_amd import flat  # noqa: E402
from test_oracle_goldens import COMMON_VIEWPORT, neighbourhood_diff  # noqa: E402
from test_oracle_light import spawn_camera  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def histogram(sp, name="fog-None-ray", fog=0):
    cam = spawn_camera(COMMON_VIEWPORT, (0.0, 10.0, 0.0), (0.4, 0.0, -1.0), view_distance=50.0)
    img = oracle.render(oracle.Space(sp), oracle.unaltered_colors(lighting=3, fog=fog, view_distance=50.0), cam, threads=os.cpu_count() or 4)["rgba8"]
    ref = np.load(os.path.join(GOLDEN, f"png_{name}.npy"))
    d = neighbourhood_diff(img, ref).max(axis=-1).ravel()
    d0 = np.abs(img.astype(int) - ref.astype(int)).max(axis=-1).ravel()
    return (f"3x3-tolerant: equal {(d == 0).sum():5d}  1-2: {((d > 0) & (d <= 2)).sum():4d}  3-15: {((d > 2) & (d <= 15)).sum():4d}  >15: {(d > 15).sum():3d}  max {d.max():2d}"
            f"   | pixel to pixel: 1-2: {((d0 > 0) & (d0 <= 2)).sum():4d}  3-15: {((d0 > 2) & (d0 <= 15)).sum():4d}  >15: {(d0 > 15).sum():3d}")


def fog_space(emission=40.0):
    z_length = 60
    sp = flat.FlatSpace((-30, 0, -z_length), (60, 20, z_length))
    sp.set_sky_uniform(scenes.from_srgb8(scenes.DAY_SKY_COLOR))
    sp.add_block(flat.air())
    floor = sp.add_block(flat.atom((0.0, 1.0, 0.5, 1.0)))
    wall = sp.add_block(flat.atom((1.0, 0.5, 0.5, 1.0)))
    pillar = sp.add_block(flat.atom((*scenes.from_srgb8(scenes.ALMOST_BLACK), 1.0)))
    lamp = sp.add_block(flat.atom((1.0, 0.05, 0.05, 1.0), (emission, 0.05, 0.05)))
    scenes._fill(sp, (-30, 0, -z_length), (60, 1, z_length), floor)
    scenes._fill(sp, (29, 0, -z_length), (1, 20, z_length), wall)
    for z in range(-z_length, 0, 2):
        x = (z * 19) % 60 + (-30)
        scenes._fill(sp, (x, 1, z), (1, 10, 1), pillar)
        sp.set((x, 8, z + 1), lamp)
    return scenes._unlit(sp)


def main():
    lib = oracle.lib()
    print("# fog-None-ray.png against the oracle's image of fog_test_universe; the reference's threshold is [(2, 500), (15, 100)] of 12288 pixels")
    print("# (its comparison is 3x3-tolerant; rendiff 0.2.2's per-pixel metric is not in the tree: max over channels is used here)")
    print("## update order: queue capacity history x hashbrown group width x batch x epsilon")
    for history, batch, hb, eps in itertools.product([0, 1], [32, 1], [16, 8], [1, 0]):
        lib.orc_set_light_build_history(C.c_int32(history))
        sp = fog_space()
        n = oracle.evaluate_light(sp, maximum_distance=30, fast=True, epsilon=eps, batch=batch, hb_width=hb)
        print(f"build-history {history}  batch {batch:2d}  group width {hb:2d}  epsilon {eps}  updates {n:5d}:  {histogram(sp)}", flush=True)
    lib.orc_set_light_build_history(C.c_int32(0))
    print("## lamp emission (the scene says 40.0): is the golden from another version of the scene?")
    for e in (40.0, 36.0, 30.0, 20.0):
        sp = fog_space(e)
        oracle.evaluate_light(sp, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16)
        print(f"emission {e:4.1f}:  {histogram(sp)}", flush=True)
    print("## where the differences are (default variant): per block kind, mean signed difference ours - golden, sRGB levels")
    sp = fog_space()
    oracle.evaluate_light(sp, maximum_distance=30, fast=True, epsilon=1, batch=32, hb_width=16)
    cam = spawn_camera(COMMON_VIEWPORT, (0.0, 10.0, 0.0), (0.4, 0.0, -1.0), view_distance=50.0)
    r = oracle.render(oracle.Space(sp), oracle.unaltered_colors(lighting=3, fog=0, view_distance=50.0), cam, threads=os.cpu_count() or 4, want_aux=True)
    d = r["rgba8"].astype(int) - np.load(os.path.join(GOLDEN, "png_fog-None-ray.npy")).astype(int)
    dm = np.abs(d).max(axis=-1)
    names = {1: "floor (0, 1, .5)", 2: "wall (1, .5, .5)", 3: "pillar ALMOST_BLACK", 4: "lamp"}
    for b in (1, 2, 3, 4):
        m = (r["aux"]["block_index"] == b) & (r["aux"]["hit"] == 1)
        print(f"  {names[b]:22s} pixels {m.sum():5d}  differing {(dm[m] > 0).sum():5d}  mean |d| {dm[m].mean():.2f}  mean signed rgb {np.round(d[m][:, :3].mean(axis=0), 2)}")
    m = r["aux"]["hit"] != 1
    print(f"  {'sky':22s} pixels {m.sum():5d}  differing {(dm[m] > 0).sum():5d}")


if __name__ == "__main__":
    main()
