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
    assert kernels == 32, "every <VOL, LMODE, DIAG, BIG> instantiation of trace_image_kernel"
    assert lookups >= 5 * kernels
    assert not problems, problems
    # The production variants (no per-pixel diagnostics, not Bounce) are built for four waves per SIMD: 128 VGPRs, the CU's 160 KB of LDS shared by the workgroups that
    # make up its 16 waves (round 5: two 512-thread workgroups of 80 KB, a pool of parked rays each), and NOTHING in scratch. Round 4 lost that once without noticing (two more wave-uniform variables cost four spilled VGPRs and 1.2 GB of scratch traffic per C3 frame while
    # the frame got faster for other reasons): the budget is checked here from now on. Template arguments: <VOL, LMODE, DIAG, BIG>.
    res = mod.kernel_resources(asm)
    assert len(res) == 32
    production = {k: v for k, v in res.items() if "Lb0ELb" in k.split("ELi")[1] and "ELi3E" not in k}
    assert len(production) == 12, sorted(production)
    for name, r in production.items():
        assert r["scratch_bytes"] == 0, (name, r)
        assert r["vgprs"] <= 128, (name, r)
        assert r["lds_bytes"] * (1024 // r["wg_threads"]) <= 160 * 1024, (name, r)  # 16 waves per CU
