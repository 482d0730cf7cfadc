"""Static checks on the compiled trace kernels (no GPU: hipcc cross-compiles): tools/check_pending_loads.py."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_register_of_a_lookup_in_flight_is_touched_before_its_wait():
    spec = importlib.util.spec_from_file_location("check_pending_loads", os.path.join(ROOT, "tools", "check_pending_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = mod.compile_to_asm()
    kernels, lookups, problems = mod.scan(asm)
    assert kernels == 44, "every <VOL, LMODE, DIAG, BIG, XC> instantiation of trace_image_kernel: 32, and the 12 production variants again with the lane exchange"
    assert lookups >= 5 * kernels
    assert not problems, problems
    # The production variants (no per-pixel diagnostics, not Bounce) are built for four waves per SIMD: 128 VGPRs, the CU's 160 KB of LDS shared by the workgroups that
    # make up its 16 waves (round 5: two 512-thread workgroups of 80 KB, a pool of parked rays each), and NOTHING in scratch. Round 4 lost that once without noticing (two more wave-uniform variables cost four spilled VGPRs and 1.2 GB of scratch traffic per C3 frame while
    # the frame got faster for other reasons): the budget is checked here from now on. Template arguments: <VOL, LMODE, DIAG, BIG>.
    res = mod.kernel_resources(asm)
    assert len(res) == 44
    import re
    def targs(k):  # <VOL, LMODE, DIAG, BIG, XC> out of the mangled name
        m = re.search(r"ILb([01])ELi(\d)ELb([01])ELb([01])ELb([01])EE", k)
        return tuple(int(x) for x in m.groups())
    production = {k: v for k, v in res.items() if targs(k)[2] == 0 and targs(k)[1] != 3}  # DIAG = false, not Bounce; with and without the exchange
    assert len(production) == 24, sorted(production)
    # No FLAT memory instruction in any trace kernel: every access names its address space (LDS, global, scalar). Round 5 read the pool's counts through a `volatile`
    # pointer, which address-space inference leaves alone: a flat load and an s_waitcnt vmcnt(0) at the top of every scheduler round (profiles/r05_experiments.txt J).
    flat = [line.strip() for line in asm.split("\n") if re.match(r"\s*flat_(load|store|atomic)", line)]
    assert not flat, flat[:5]
    for name, r in production.items():
        assert r["scratch_bytes"] == 0, (name, r)
        assert r["vgprs"] <= 128, (name, r)
        assert r["lds_bytes"] * (1024 // r["wg_threads"]) <= 160 * 1024, (name, r)  # 16 waves per CU
