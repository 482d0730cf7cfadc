import os, sys, numpy as np
sys.path.insert(0, '/root/repo/tools/wave_sim'); sys.path.insert(0, '/root/repo')
import run as R
import bench, oracle
lib = R.build_lib()
# run.tokens builds rays from bench.build_workload's camera; patch the eye by rotating about the target
import types
orig = bench.build_workload
def rotated(deg):
    def bw(wl):
        sp, wh, eye, target, vd, label = orig(wl)
        a = np.radians(deg)
        dx, dz = eye[0] - target[0], eye[2] - target[2]
        e2 = (target[0] + dx * np.cos(a) + dz * np.sin(a), eye[1], target[2] - dx * np.sin(a) + dz * np.cos(a))
        return sp, wh, e2, target, vd, label
    return bw
def cost_of(off, w, h):
    ln = np.diff(off).reshape(h, w)
    my, mx = (h + 15) // 16, (w + 15) // 16
    c = np.zeros((my, mx), np.int32)
    for yy in range(my):
        blk = ln[yy * 16:(yy + 1) * 16]
        for xx in range(mx):
            m = blk[:, xx * 16:(xx + 1) * 16].max()
            c[yy, xx] = m if m > 48 else 0
    return c
res = {}
for deg in (0.0, 6.0, 12.0, 18.0):
    bench.build_workload = rotated(deg)
    tok, off, w, h, label, keep = R.tokens(lib, 'atrium', 1)
    res[deg] = (tok, off, w, h, cost_of(off, w, h))
bench.build_workload = orig
p = R.defaults(res[0.0][2], res[0.0][3])
p.pool=64; p.reservoir=1; p.policy=3; p.deposit_free=3; p.min_gain=8; p.c_xchg_base=75; p.c_xchg_move=450
p.c_shade, p.c_enter, p.c_finish, p.c_refill, p.c_newray = 1290, 815, 550, 456, 430
for deg in (6.0, 12.0, 18.0):
    tok, off, w, h, c_true = res[deg]
    prev = res[deg - 6.0][4]
    from scipy.stats import spearmanr
    print(f"== camera at {deg} degrees; the record of the frame 6 degrees before: rank correlation with this frame's own {spearmanr(prev.ravel(), c_true.ravel()).correlation:.3f}")
    for name, co, cost in (("own record (warm)", 0, None), ("index order", 1, None), ("the previous frame's record (6 degrees stale)", 0, prev)):
        p.cold_order = co
        if cost is not None:
            cost.ravel().astype(np.int32).tofile('/tmp/stale_cost.bin'); os.environ['SIM_COST'] = '/tmp/stale_cost.bin'
        elif 'SIM_COST' in os.environ: del os.environ['SIM_COST']
        R.run(lib, tok, off, p, name)
# --- the previous frame's record REPROJECTED: the point at distance D along the new camera's ray through a macro tile's centre, seen from the previous camera ---
def cam_inv(deg):
    sp, (W, H), eye, target, vd, label = rotated(deg)('atrium')
    _, _, inv = oracle.camera_matrices(90.0, vd, W / H, oracle.look_at_y_up(eye, target), eye)
    return np.asarray(inv, float).reshape(4, 4), np.array(eye, float), np.array(target, float), W, H
def reproject(prev_cost, deg_prev, deg_new, D):
    invB, eyeB, tgt, W, H = cam_inv(deg_new)
    invA, eyeA, _, _, _ = cam_inv(deg_prev)
    fwdA = np.linalg.inv(invA)
    my, mx = prev_cost.shape
    out = np.zeros_like(prev_cost)
    xs = (np.arange(mx) * 16 + 8) / W * 2 - 1
    ys = -((np.arange(my) * 16 + 8) / H * 2 - 1)
    X, Y = np.meshgrid(xs, ys)
    def unp(z):
        v = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z), np.ones(X.size)], 1) @ invB
        return v[:, :3] / v[:, 3:4]
    o, f = unp(0.0), unp(1.0)
    d = f - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    P = o + d * D
    q = np.concatenate([P, np.ones((len(P), 1))], 1) @ fwdA
    ok = q[:, 3] > 1e-9
    nd = q[:, :3] / np.where(ok, q[:, 3], 1.0)[:, None]
    px = ((nd[:, 0] + 1) / 2 * W); py = ((1 - nd[:, 1]) / 2 * H)
    inside = ok & (px >= 0) & (px < W) & (py >= 0) & (py < H)
    ix = np.clip((px // 16).astype(int), 0, mx - 1); iy = np.clip((py // 16).astype(int), 0, my - 1)
    vals = np.where(inside, prev_cost[iy, ix], 0)
    return vals.reshape(my, mx)
radius = float(np.linalg.norm(cam_inv(0.0)[1] - cam_inv(0.0)[2]))
for deg in (6.0, 12.0):
    tok, off, w, h, c_true = res[deg]
    for D in (radius * 0.5, radius, radius * 2.0, 50.0):
        rp = reproject(res[deg - 6.0][4], deg - 6.0, deg, D)
        rp.ravel().astype(np.int32).tofile('/tmp/stale_cost.bin'); os.environ['SIM_COST'] = '/tmp/stale_cost.bin'; p.cold_order = 0
        print(f"camera at {deg}: reprojected at distance {D:.1f}: rank correlation {spearmanr(rp.ravel(), c_true.ravel()).correlation:.3f}")
        R.run(lib, tok, off, p, f"reprojected, D = {D:.1f}")
