"""The wave-scheduler model (tools/wave_sim): it builds, reproduces its own invariants, and the exchange policies it was used to choose still rank as recorded
(profiles/r05_experiments.txt A). CPU only; a small workload."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools", "wave_sim"))


def test_model_conserves_work_and_ranks_the_policies():
    import run as R

    lib = R.build_lib()
    tok, off, w, h, _, _ = R.tokens(lib, "small", 1)
    n = len(off) - 1
    assert n == w * h and off[-1] == len(tok)
    t = tok.view(np.uint8)
    ends = np.isin(t[off[1:][np.diff(off) > 0] - 1], [ord("O"), ord("X")])
    assert ends.all(), "every ray that enters the space ends in O (opaque) or X (left the grid / step cap)"

    def run(**kw):
        p = R.defaults(w, h)
        p.n_cus = 4
        for k, v in kw.items():
            setattr(p, k, v)
        o = R.Out()
        assert lib.simulate(tok.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), ctypes.byref(o)) == 0
        return o

    base = run()
    shade_tokens = int(((t == ord("S")) | (t == ord("O"))).sum())
    enter_tokens = int((t == ord("E")).sum())
    assert base.lanes[1] == shade_tokens and base.lanes[2] == enter_tokens, "every SHADE / ENTER token is served exactly once"
    pooled = run(pool=72, reservoir=1, policy=1, deposit_free=3, min_gain=2)
    assert pooled.lanes[1] == shade_tokens and pooled.lanes[2] == enter_tokens, "... with the exchange too: rays are moved, never lost or served twice"
    assert pooled.phases[1] <= base.phases[1] and pooled.trips <= base.trips, "a shared pool with a reservoir makes fuller (fewer) phases"
