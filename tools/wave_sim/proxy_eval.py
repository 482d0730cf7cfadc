import os, sys, numpy as np
sys.path.insert(0, '/root/repo/tools/wave_sim'); sys.path.insert(0, '/root/repo')
import run as R
import bench, oracle
wl = sys.argv[1] if len(sys.argv) > 1 else 'atrium'
lib = R.build_lib()
ST = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tok, off, w, h, label, keep = R.tokens(lib, wl, ST)
sp, (W, H), eye, target, vd, _ = bench.build_workload(wl); W //= ST; H //= ST
proj_view, _, inv = oracle.camera_matrices(90.0, vd, W / H, oracle.look_at_y_up(eye, target), eye)[0:3]
inv = np.asarray(inv, float).reshape(4, 4)
fwd = np.linalg.inv(inv)   # world (row vector) -> clip
# recursive blocks
rec = np.array([not (b.is_one or b.resolution == 1) for b in sp.blocks])
g = rec[sp.block_index]
idx = np.argwhere(g)
c = idx + np.array(sp.lo) + 0.5
v = np.concatenate([c, np.ones((len(c), 1))], 1) @ fwd
wq = v[:, 3]
ok = wq > 1e-9
ndc = v[ok, :3] / wq[ok, None]
inside = (np.abs(ndc[:, 0]) < 1) & (np.abs(ndc[:, 1]) < 1) & (ndc[:, 2] > 0) & (ndc[:, 2] < 1)
ndc = ndc[inside]
px = ((ndc[:, 0] + 1) / 2 * W).astype(int).clip(0, W - 1)
py = ((1 - (ndc[:, 1] + 1) / 2) * H).astype(int).clip(0, H - 1)
tiles_x = (W + 7) // 8; tiles_y = (H + 7) // 8
mx_n = (tiles_x + 1) // 2; my_n = (tiles_y + 1) // 2
cost = np.zeros(mx_n * my_n, np.int32)
np.add.at(cost, (py // 16) * mx_n + (px // 16), 1)
print("recursive cubes", len(idx), "in view", len(ndc), "tiles with cost", (cost > 0).sum(), "of", len(cost), "max", cost.max())
# the true record for comparison
ln = np.diff(off).reshape(H, W)
true = np.zeros(mx_n * my_n, np.int32)
for yy in range(my_n):
    blk = ln[yy * 16:(yy + 1) * 16]
    for xx in range(mx_n):
        m = blk[:, xx * 16:(xx + 1) * 16].max()
        true[yy * mx_n + xx] = m if m > 48 else 0
from scipy.stats import spearmanr
print("spearman(proxy, true)", spearmanr(cost, true).correlation, " true>0 tiles", (true > 0).sum())
cost.tofile('/tmp/proxy_cost.bin')
# smoothed proxy: a 3x3 box sum (a cube covers more than its centre's tile)
c2 = cost.reshape(my_n, mx_n).astype(np.int64)
pad = np.pad(c2, 1)
sm = sum(pad[1 + dy:1 + dy + my_n, 1 + dx:1 + dx + mx_n] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
sm.astype(np.int32).ravel().tofile('/tmp/proxy_cost_sm.bin')
print("spearman(smoothed, true)", spearmanr(sm.ravel(), true).correlation)
p = R.defaults(w, h); p.n_cus = max(1, 256 // (ST * ST))
p.pool=64; p.reservoir=1; p.policy=3; p.deposit_free=3; p.min_gain=8; p.c_xchg_base=75; p.c_xchg_move=450
p.c_shade, p.c_enter, p.c_finish, p.c_refill, p.c_newray = 1290, 815, 550, 456, 430
for name, co, env in (("warm (true record)", 0, None), ("cold (index order)", 1, None), ("proxy: centre splat", 0, '/tmp/proxy_cost.bin'), ("proxy: 3x3 smoothed", 0, '/tmp/proxy_cost_sm.bin')):
    p.cold_order = co
    if env: os.environ['SIM_COST'] = env
    elif 'SIM_COST' in os.environ: del os.environ['SIM_COST']
    R.run(lib, tok, off, p, name)
# footprint splat: the bounding rectangle of the cube's eight projected corners, in macro tiles
corners = np.array([[dx, dy, dz] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], float)
base = idx + np.array(sp.lo)
cost3 = np.zeros((my_n, mx_n), np.int32)
n_used = 0
for b in base:
    pts = np.concatenate([b + corners, np.ones((8, 1))], 1) @ fwd
    if (pts[:, 3] <= 1e-9).any(): continue
    nd = pts[:, :3] / pts[:, 3:4]
    if nd[:, 0].max() < -1 or nd[:, 0].min() > 1 or nd[:, 1].max() < -1 or nd[:, 1].min() > 1 or nd[:, 2].max() < 0 or nd[:, 2].min() > 1: continue
    x0 = int(np.clip((nd[:, 0].min() + 1) / 2 * W, 0, W - 1)) // 16; x1 = int(np.clip((nd[:, 0].max() + 1) / 2 * W, 0, W - 1)) // 16
    y0 = int(np.clip((1 - (nd[:, 1].max() + 1) / 2) * H, 0, H - 1)) // 16; y1 = int(np.clip((1 - (nd[:, 1].min() + 1) / 2) * H, 0, H - 1)) // 16
    cost3[y0:y1 + 1, x0:x1 + 1] += 1
    n_used += 1
print("footprint splat: cubes used", n_used, "spearman", spearmanr(cost3.ravel(), true).correlation, "max", cost3.max())
cost3.ravel().tofile('/tmp/proxy_cost_fp.bin')
os.environ['SIM_COST'] = '/tmp/proxy_cost_fp.bin'; p.cold_order = 0
R.run(lib, tok, off, p, "proxy: footprint splat")
