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
    kernels, lookups, problems = mod.scan(mod.compile_to_asm())
    assert kernels == 32, "every <VOL, LMODE, DIAG, BIG> instantiation of trace_image_kernel"
    assert lookups >= 5 * kernels
    assert not problems, problems
