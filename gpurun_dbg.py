import numpy as np, oracle
from all_is_cubes_amd import abi
from tests import scenes
from tests.test_gpu_parity import render_both
ctx = abi.Context(0)
for size in [(128,96),(64,48),(16,16),(32,16),(17,9)]:
    got, ref = render_both(ctx, scenes.transparent_one_space(), oracle.unaltered_colors(transparency=0), size, (0.5,0.5,2.0))
    ga, ra = got["aux"], ref["aux"]
    print(size, "gpu total", got["info"].cubes_traced, "sum aux", int(ga["cubes_traced"].sum()), "oracle", int(ref["info"]["cubes_traced"]),
          "aux equal", bool((ga["cubes_traced"]==ra["cubes_traced"]).all()), "n_outer", got["info"].n_outer, int(ref["info"]["n_outer"]), "img diff", int(np.abs(got["rgba8"].astype(int)-ref["rgba8"].astype(int)).max()))
