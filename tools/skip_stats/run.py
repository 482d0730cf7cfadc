"""Run-length / macro-step statistics of the bench workloads (table in profiles/r04_experiments.txt A).
CPU only: a plain f64 DDA over a sample of the workload's rays (tools/skip_stats/skip_stats.cpp, built here with g++).

    python tools/skip_stats/run.py atrium 4      # every 4th pixel in x and y
    python tools/skip_stats/run.py s256 8
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import bench  # noqa: E402
import oracle  # noqa: E402  (camera matrices only)

FAST_STEP, FULL = 38, 0  # wave-instructions of one fast step


def build_lib():
    so = os.path.join(tempfile.gettempdir(), "libskip_stats.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "skip_stats.cpp")])
    return ctypes.CDLL(so)


def cheb(empty, cap):
    """m(c) = min(cap, Chebyshev distance to the nearest non-empty or out-of-bounds cell - 1): the (2m+1)^3 cube around c is empty."""
    p = np.pad(empty, 1, constant_values=False)
    d = ndimage.distance_transform_cdt(p, metric="chessboard")[1:-1, 1:-1, 1:-1]
    return np.ascontiguousarray(np.clip(d - 1, 0, cap).astype(np.uint8) * empty)


def dirskip(empty, cap):
    """Per ray octant: m(c) = the largest m <= cap such that the (m+1)^3 cube anchored at c and extending along the octant is empty."""
    out = []
    for o in range(8):
        fx, fy, fz = (o >> 2) & 1, (o >> 1) & 1, o & 1
        e = empty
        if not fx: e = e[::-1]
        if not fy: e = e[:, ::-1]
        if not fz: e = e[:, :, ::-1]
        cur = np.pad(e.astype(np.int32), ((0, 1), (0, 1), (0, 1)))
        for _ in range(cap):
            mn = None
            for dx in (0, 1):
                for dy in (0, 1):
                    for dz in (0, 1):
                        if dx == dy == dz == 0:
                            continue
                        sh = cur[dx:cur.shape[0] - 1 + dx, dy:cur.shape[1] - 1 + dy, dz:cur.shape[2] - 1 + dz]
                        mn = sh if mn is None else np.minimum(mn, sh)
            cur[:-1, :-1, :-1] = np.where(e, 1 + mn, 0)
        m = np.clip(cur[:-1, :-1, :-1] - 1, 0, cap).astype(np.uint8)
        if not fx: m = m[::-1]
        if not fy: m = m[:, ::-1]
        if not fz: m = m[:, :, ::-1]
        out.append(np.ascontiguousarray(m))
    return np.ascontiguousarray(np.stack(out))


def brick_empty(empty, B):
    s = empty.shape
    pad = [(0, (-n) % B) for n in s]
    e = np.pad(empty, pad, constant_values=False)  # a partial brick at the edge is not empty (the bounds end inside it)
    e = e.reshape(e.shape[0] // B, B, e.shape[1] // B, B, e.shape[2] // B, B)
    return np.ascontiguousarray(e.all(axis=(1, 3, 5)).astype(np.uint8))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cap = 7
    lib = build_lib()
    sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    inv = np.asarray(inv, float).reshape(4, 4)
    xs = (np.arange(0, w, stride) + 0.5) / w * 2 - 1
    ys = -((np.arange(0, h, stride) + 0.5) / h * 2 - 1)
    X, Y = np.meshgrid(xs, ys)

    def unp(z):
        v = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z), np.ones(X.size)], 1) @ inv
        return v[:, :3] / v[:, 3:4]

    o, f = unp(0.0), unp(1.0)
    rays = np.ascontiguousarray(np.concatenate([o, f - o], 1))
    n = len(rays)
    nb = len(sp.blocks)
    bclass = np.zeros(nb, np.uint8)
    bcls, bres, blo, bsz = [], [], [], []
    for i, b in enumerate(sp.blocks):
        pal = b.palette
        vis = ~((pal[:, 3] == 0) & (pal[:, 4] == 0) & (pal[:, 5] == 0) & (pal[:, 6] == 0))
        opq = pal[:, 3] == 1.0
        if b.is_one:
            bclass[i] = 0 if not vis[0] else (1 if opq[0] else 2)
            bcls.append(np.zeros((1, 1, 1), np.uint8))
        else:
            bclass[i] = 3
            bcls.append(np.ascontiguousarray(np.where(vis[b.voxels], np.where(opq[b.voxels], 1, 2), 0).astype(np.uint8)))
        bres.append(b.resolution); blo.append(b.vlo); bsz.append(b.voxels.shape)
    gcls = np.ascontiguousarray(bclass[sp.block_index])
    gblk = np.ascontiguousarray(sp.block_index.astype(np.uint16))
    P = ctypes.POINTER(ctypes.c_uint8)
    V = ctypes.c_void_p
    bres = np.array(bres, np.int32); blo = np.ascontiguousarray(np.array(blo, np.int32)); bsz = np.ascontiguousarray(np.array(bsz, np.int32))
    gs = np.array(sp.size, np.int32); glo = np.array(sp.lo, np.int32)
    bc = (P * nb)(*[a.ctypes.data_as(P) for a in bcls])
    print(f"== {wl}: {label}; {n} rays (every {stride}th pixel); cube grid {(gcls == 0).mean():.3f} invisible")

    # ---- aligned empty bricks: how long does a ray stay inside one? ----
    for B in (2, 4, 8):
        ge = brick_empty(gcls == 0, B)
        bes = [brick_empty(c == 0, B) for c in bcls]
        be = (P * nb)(*[a.ctypes.data_as(P) for a in bes])
        out = np.zeros(84)
        lib.brick_runs(gcls.ctypes.data_as(P), gblk.ctypes.data_as(V), gs.ctypes.data_as(V), glo.ctypes.data_as(V), nb, bc, bres.ctypes.data_as(V),
                       blo.ctypes.data_as(V), bsz.ctypes.data_as(V), ge.ctypes.data_as(P), be, B, rays.ctypes.data_as(V), ctypes.c_long(n), out.ctypes.data_as(V))
        for lvl, o_ in (("cube grid", out[:42]), ("voxels", out[42:])):
            cells, inside = o_[0], o_[1]
            hist = o_[2:]
            runs = hist.sum()
            mean = inside / runs if runs else 0.0
            top = " ".join(f"{k}:{hist[k] / runs:.2f}" for k in range(1, min(40, 3 * B + 1)) if runs and hist[k] / runs >= 0.005)
            print(f"  aligned {B}^3 bricks, {lvl}: {cells / n:.2f} cells/ray, {inside / max(cells, 1):.1%} of them inside an all-invisible brick, "
                  f"{runs / n:.2f} runs/ray, mean run {mean:.2f} cells; run-length histogram {top}")

    # ---- macro steps: isotropic (Chebyshev) and per-octant fields ----
    for name, fn, dirn in (("isotropic (2m+1)^3 field", cheb, 0), ("per-octant (m+1)^3 field", dirskip, 1)):
        lib.set_dir(dirn)
        gskip = fn(gcls == 0, cap)
        bsk = [fn(c == 0, cap) if c.size > 1 else np.zeros((8 if dirn else 1,) + c.shape, np.uint8) for c in bcls]
        bs = (P * nb)(*[a.ctypes.data_as(P) for a in bsk])
        for nq in ([], [2], [2, 4], [2, 3, 4, 5, 6, 7, 8]):
            q = np.array(nq if nq else [99], np.int32); out = np.zeros(72)
            lib.skip_stats(gcls.ctypes.data_as(P), gskip.ctypes.data_as(P), gblk.ctypes.data_as(V), gs.ctypes.data_as(V), glo.ctypes.data_as(V), nb, bc, bs,
                           bres.ctypes.data_as(V), blo.ctypes.data_as(V), bsz.ctypes.data_as(V), rays.ctypes.data_as(V), ctypes.c_long(n),
                           q.ctypes.data_as(V), len(q), out.ctypes.data_as(V))
            line = f"  {name}, macro sizes {nq or 'none'}:"
            iters = 0.0; cells_all = 0.0; valu = 0.0
            for lvl, o_ in (("grid", out[:36]), ("voxels", out[36:])):
                cells, single = o_[0], o_[1]
                mac = o_[4:].reshape(16, 2)
                line += f" {lvl} {cells / n:.1f} cells/ray = {single / n:.1f} single"
                iters += single; cells_all += cells; valu += single * 16
                for k in range(16):
                    if mac[k, 0]:
                        line += f" + {mac[k, 0] / n:.2f} x n={k} ({mac[k, 1] / mac[k, 0]:.2f} cells each)"
                        iters += mac[k, 0]
                        valu += mac[k, 0] * (16 + 10 * k)  # VALU of an exact n-step: 3(n-1) adds, 2 min, 2+3(n-1)+1 compares, 3x4 + 3x4 updates (n = 2: 36)
                line += ";"
            line += f" dependent lookups per ray {cells_all / n:.1f} -> {iters / n:.1f}; VALU of the stepping per ray {cells_all * 16 / n:.0f} -> {valu / n:.0f}"
            print(line)


if __name__ == "__main__":
    main()
