"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/aic_hip.h declares; no compute is attempted without a GPU."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from all_is_cubes_amd import abi, flat

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "aic_hip.h").read_text()
    declared = set(re.findall(r"\b(aic_[a-z0-9_]+)\s*\(", header))
    assert declared == set(abi.ABI_SYMBOLS)
    lib = abi.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert lib.aic_abi_version() == 3


def test_every_entry_point_taking_a_context_has_its_ctypes_signature():
    """A ctypes call without argtypes passes a Python int as a 32-bit C int: a 64-bit context handle is cut in half and the call
    crashes on the GPU box (round 4 met this in the first GPU run of aic_stream_wait_frame). Every context-taking entry point the
    binding calls must therefore declare its argument types."""
    header = (ROOT / "include" / "aic_hip.h").read_text()
    src = (ROOT / "all_is_cubes_amd" / "abi.py").read_text()
    takers = re.findall(r"\b(aic_[a-z0-9_]+)\s*\(\s*(?:const\s+)?aic_(?:ctx|multi)\s*\*", header)
    assert len(takers) > 30
    for name in takers:
        if f"_lib.{name}(" in src or f"lib.{name}(" in src:
            assert f"lib.{name}.argtypes" in src, f"{name}: called from abi.py without argtypes"


def test_struct_layouts_match_header():
    assert ctypes.sizeof(abi.BlockDesc) == 48 == flat.BLOCK_DTYPE.itemsize
    assert abi.PIXEL_AUX_DTYPE.itemsize == 56
    assert ctypes.sizeof(abi.Options) == 48
    assert ctypes.sizeof(abi.Camera) == 136
    assert ctypes.sizeof(abi.FrameInfo) == 64


def test_partition_rows_helper():
    lib = abi.load()
    for h, strip, n in [(1080, 16, 8), (37, 8, 3), (96, 16, 2), (5, 16, 4), (0, 16, 2)]:
        total = 0
        for p in range(n):
            part = abi.Partition(strip, n, p, 0)
            total += lib.aic_partition_rows(h, ctypes.byref(part))
        assert total == h


def test_no_gpu_fails_loudly():
    """Without a usable device the product must raise, never fall back to a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.AicError):
        abi.Context(0)


def test_block_sky_host_matches_oracle():
    import oracle
    from tests import scenes

    sp = scenes.one_cube_space()
    rng = np.random.default_rng(5)
    for kind in (0, 1):
        if kind == 1:
            sp.set_sky_octants(rng.uniform(0.0, 4.0, (8, 3)))
        got = abi.block_sky_texels(sp.sky_kind, sp.sky)
        ref = oracle.block_sky(oracle.Space(sp))
        assert (got == ref).all()


def test_loading_the_library_puts_a_hardware_queue_default_in_place():
    """Frame slots are HIP streams and the runtime's four hardware queues make slots share one as soon as the host has streams of its own
    (profiles/r06_hw_queues.txt): the library sets GPU_MAX_HW_QUEUES=8 when it is loaded, leaves a caller's value alone, and keeps its
    hands off under AIC_KEEP_HW_QUEUES=1."""
    import os
    import subprocess
    import sys

    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from all_is_cubes_amd import abi; abi.load(); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; print(libc.getenv(b'GPU_MAX_HW_QUEUES'))" % str(ROOT))
    for given, keep, want in [(None, None, "b'8'"), ("2", None, "b'2'"), (None, "1", "None")]:
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "AIC_KEEP_HW_QUEUES")}
        if given is not None:
            env["GPU_MAX_HW_QUEUES"] = given
        if keep is not None:
            env["AIC_KEEP_HW_QUEUES"] = keep
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        assert out.stdout.strip().splitlines()[-1] == want, (given, keep, out.stdout)
