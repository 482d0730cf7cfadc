"""Host-CPU probe for the cpu_baseline leg: what the box offers and how the oracle scales on it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import oracle
import bench

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)'; cat /proc/loadavg")
flat_space, (w, h), eye, target, vd, label = bench.build_workload("atrium")
sp = oracle.Space(flat_space)
oo = oracle.make_options(view_distance=vd)
q = oracle.look_at_y_up(eye, target)
_, _, inv = oracle.camera_matrices(90.0, vd, w / h, q, eye)
cam = oracle.make_camera(inv, w, h)
for th in (1, 8, 16, 32, 64, 128, 256):
    rows = (0, h) if th > 1 else (h // 2 - 20, h // 2 + 20)
    oracle.render(sp, oo, cam, rows=rows, threads=th)
    t = time.perf_counter(); n = 0
    while time.perf_counter() - t < 2.0:
        oracle.render(sp, oo, cam, rows=rows, threads=th); n += 1
    dt = time.perf_counter() - t
    print(f"threads {th:4d}: {n * (rows[1]-rows[0]) * w / dt / 1e6:8.3f} Mrays/s  ({dt / n * 1e3:.1f} ms/call)")
