"""Discrete-event model of trace_image_kernel's wave scheduler (tools/wave_sim/wave_sim.cpp): baseline and lane-exchange policies.
CPU only.   python tools/wave_sim/run.py atrium [stride]    /    python tools/wave_sim/run.py s256 2
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import bench  # noqa: E402
import oracle  # noqa: E402  (camera matrices only)


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("width height n_cus wg_per_cu waves_per_wg t_batch n_few frac_t frac_n step_reps fast_steps fast_min "
                                            "c_sched c_fast c_pass c_leave c_shade c_enter c_finish c_refill c_newray").split()] + \
               [(n, ctypes.c_double) for n in "cyc_per_inst cyc_lone lat_step".split()] + \
               [(n, ctypes.c_int) for n in "pool reservoir c_xchg_base c_xchg_move min_gain policy deposit_free keep_free cold_order hot retire".split()]


class Out(ctypes.Structure):
    _fields_ = [("makespan", ctypes.c_double), ("busy_inst", ctypes.c_double), ("phases", ctypes.c_double * 4), ("lanes", ctypes.c_double * 4)] + \
               [(n, ctypes.c_double) for n in "fast_iters fast_lanes pass_iters pass_lanes trips trip_lanes xchg_rounds xchg_moved sched_rounds".split()] + \
               [("inst_kind", ctypes.c_double * 6), ("dry_time_median", ctypes.c_double), ("inst_dry", ctypes.c_double), ("retired", ctypes.c_double)]


def build_lib():
    so = os.path.join(tempfile.gettempdir(), "libwave_sim.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "wave_sim.cpp")])
    return ctypes.CDLL(so)


def tokens(lib, wl, stride):
    sp, (w, h), eye, target, vd, label = bench.build_workload(wl)
    w //= stride; h //= stride
    _, _, inv = oracle.camera_matrices(90.0, vd, w / h, oracle.look_at_y_up(eye, target), eye)
    inv = np.asarray(inv, float).reshape(4, 4)
    xs = (np.arange(w) + 0.5) / w * 2 - 1
    ys = -((np.arange(h) + 0.5) / h * 2 - 1)
    X, Y = np.meshgrid(xs, ys)

    def unp(z):
        v = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z), np.ones(X.size)], 1) @ inv
        return v[:, :3] / v[:, 3:4]

    o, f = unp(0.0), unp(1.0)
    rays = np.ascontiguousarray(np.concatenate([o, f - o], 1))
    n = len(rays)
    nb = len(sp.blocks)
    bclass = np.zeros(nb, np.int8)
    balpha1 = np.zeros(nb, np.float32)
    bal, bres, blo, bsz = [], [], [], []
    for i, b in enumerate(sp.blocks):
        pal = b.palette
        vis = ~((pal[:, 3] == 0) & (pal[:, 4] == 0) & (pal[:, 5] == 0) & (pal[:, 6] == 0))
        al = np.where(vis, pal[:, 3], -1.0).astype(np.float32)
        if b.is_one or b.resolution == 1:
            bclass[i] = 1 if vis[0] else 0
            balpha1[i] = al[0]
            bal.append(np.zeros((1, 1, 1), np.float32))
        else:
            bclass[i] = 3
            bal.append(np.ascontiguousarray(al[b.voxels]))
        bres.append(b.resolution); blo.append(b.vlo); bsz.append(bal[-1].shape)
    gcls = np.ascontiguousarray(bclass[sp.block_index])
    galpha = np.ascontiguousarray(balpha1[sp.block_index])
    gblk = np.ascontiguousarray(sp.block_index.astype(np.uint16))
    PF = ctypes.POINTER(ctypes.c_float)
    V = ctypes.c_void_p
    bres = np.array(bres, np.int32); blo = np.ascontiguousarray(np.array(blo, np.int32)); bsz = np.ascontiguousarray(np.array(bsz, np.int32))
    gs = np.array(sp.size, np.int32); glo = np.array(sp.lo, np.int32)
    bc = (PF * nb)(*[a.ctypes.data_as(PF) for a in bal])
    cap = n * 1100
    cap = min(cap, 6_000_000_000)
    out = np.zeros(min(cap, n * 400), np.int8)
    off = np.zeros(n + 1, np.int64)
    lib.make_tokens.restype = ctypes.c_long
    got = lib.make_tokens(galpha.ctypes.data_as(V), gcls.ctypes.data_as(V), gblk.ctypes.data_as(V), gs.ctypes.data_as(V), glo.ctypes.data_as(V), nb, bc,
                          bres.ctypes.data_as(V), blo.ctypes.data_as(V), bsz.ctypes.data_as(V), rays.ctypes.data_as(V), ctypes.c_long(n),
                          out.ctypes.data_as(V), ctypes.c_long(len(out)), off.ctypes.data_as(V))
    assert got >= 0, "token buffer too small"
    return out[:got].copy(), off, w, h, label, (bal,)


def defaults(w, h):
    p = Params()
    p.width, p.height = w, h
    p.n_cus, p.wg_per_cu, p.waves_per_wg = 256, 4, 4
    p.t_batch, p.n_few, p.frac_t, p.frac_n = 32, 24, 4, 3
    p.step_reps, p.fast_steps, p.fast_min = 2, 16, 16
    # wave-instructions per phase, from profiles/r04_phase_cycles.txt (cycles per phase / 11.4 cycles per instruction at 4 waves per SIMD)
    p.c_sched, p.c_fast, p.c_pass, p.c_leave = 45, 38, 180, 35
    p.c_shade, p.c_enter, p.c_finish, p.c_refill, p.c_newray = 1070, 1000, 560, 420, 520
    p.cyc_per_inst, p.cyc_lone, p.lat_step = 2.84, 5.5, 600.0
    p.pool, p.reservoir, p.c_xchg_base, p.c_xchg_move, p.min_gain, p.policy, p.deposit_free, p.keep_free, p.cold_order = 0, 0, 40, 110, 4, 0, 0, 0, 0
    return p


def run(lib, tok, off, p, name, clock_ghz=2.3, scale=1.0):
    o = Out()
    t0 = time.time()
    rc = lib.simulate(tok.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), ctypes.byref(o))
    if rc: print(name, "SIM FAILED", rc); return o
    n_simd = p.n_cus * 4
    thr_ms = o.busy_inst * p.cyc_per_inst / n_simd / (clock_ghz * 1e6)   # issue-bound time of the frame's instructions
    lat_ms = o.makespan / (clock_ghz * 1e6)
    ph = [o.phases[k] for k in range(4)]
    ln = [o.lanes[k] / max(o.phases[k], 1) for k in range(4)]
    tot_lane_inst = (o.lanes[1] * p.c_shade + o.lanes[2] * p.c_enter + o.lanes[3] * (p.c_finish + p.c_refill + p.c_newray) + o.fast_lanes * p.c_fast + o.pass_lanes * p.c_pass)
    tot_inst = (o.phases[1] * p.c_shade + o.phases[2] * p.c_enter + o.phases[3] * (p.c_finish + p.c_refill + p.c_newray) + o.fast_iters * p.c_fast + o.pass_iters * p.c_pass)
    print(f"{name:44s} inst {o.busy_inst / 1e6 * scale:7.1f} M  issue-bound {thr_ms * scale:6.3f} ms  one frame {lat_ms:6.3f} ms | phases k: SHADE {ph[1] / 1e3 * scale:6.1f} @{ln[1]:4.1f}  ENTER {ph[2] / 1e3 * scale:5.1f} @{ln[2]:4.1f}  "
          f"RAY {ph[3] / 1e3 * scale:5.1f} @{ln[3]:4.1f}  trips {o.trips / 1e3 * scale:6.1f} @{o.trip_lanes / max(o.trips, 1):4.1f}  fast {o.fast_iters / 1e3 * scale:7.1f} @{o.fast_lanes / max(o.fast_iters, 1):4.1f}  "
          f"pass {o.pass_iters / 1e3 * scale:6.1f} @{o.pass_lanes / max(o.pass_iters, 1):4.1f} | lane util {tot_lane_inst / 64 / max(tot_inst, 1):.3f}  xchg rounds {o.xchg_rounds / 1e3 * scale:6.1f} k moved {o.xchg_moved / max(o.xchg_rounds, 1):4.1f}  inst after dry {o.inst_dry / max(o.busy_inst, 1):.3f}  retired {o.retired * scale:.0f}  ({time.time() - t0:.1f} s)",
          flush=True)
    return o


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = build_lib()
    t0 = time.time()
    tok, off, w, h, label, keep = tokens(lib, wl, stride)
    n = len(off) - 1
    ln = np.diff(off)
    t = tok.view(np.uint8)
    cnt = {c: int((t == ord(c)).sum()) for c in "flSOELX"}
    print(f"== {wl}: {label}; {w}x{h} ({n} rays, every {stride}th pixel), {time.time() - t0:.1f} s; tokens per ray {ln.mean():.1f} (max {ln.max()}): " +
          " ".join(f"{c} {v / n:.2f}" for c, v in cnt.items()))
    scale = stride * stride
    p = defaults(w, h)
    if stride > 1:
        p.n_cus = max(1, 256 // (stride * stride))
    run(lib, tok, off, p, "baseline (4 waves/WG, no exchange)", scale=scale)
    exec(os.environ.get("SIM_EXTRA", ""))
    return lib, tok, off, p, scale


if __name__ == "__main__":
    main()
