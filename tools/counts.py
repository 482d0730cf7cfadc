import sys; sys.path.insert(0,'/root/repo')
import bench, oracle
from all_is_cubes_amd import abi
for wl in ("atrium","s256"):
    sp,(w,h),eye,target,vd,label = bench.build_workload(wl)
    _,_,inv = oracle.camera_matrices(90.0, vd, w/h, oracle.look_at_y_up(eye,target), eye)
    with abi.Context(0) as ctx:
        ctx.upload_space(0, sp); ctx.set_options(0, abi.make_options(view_distance=vd))
        i = ctx.render(ctx.make_frame(w,h,world_inv=inv), counters=True)["info"]
        n=w*h
        print(wl, "steps/ray", i.cubes_traced/n, "outer lookups/ray", i.n_outer/n, "inner/ray", i.n_inner/n, "hits/ray", i.n_hits/n, "light texels/ray", i.n_light/n)
