"""The bench line's contract (the driver parses it), checked on the lines committed under profiles/ and on bench.py's own defaults.
No GPU: the lines were produced on an MI355X by tools/measure_round.sh."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def latest_round():
    rounds = sorted({os.path.basename(p).split("_")[0] for p in LINES})
    return rounds[-1] if rounds else None


def test_committed_lines_exist():
    assert LINES, "no bench lines under profiles/"
    r = latest_round()
    names = {os.path.basename(p) for p in LINES if os.path.basename(p).startswith(r)}
    assert f"{r}_bench_atrium.json" in names and f"{r}_bench_s256.json" in names


def round_no(path):
    return int(os.path.basename(path)[1:3])


@pytest.mark.parametrize("path", [p for p in LINES if round_no(p) >= 3] or LINES[-1:])
def test_line_has_the_contract_fields_and_is_consistent(path):
    text = open(path).read().strip()
    assert "\n" not in text, "one JSON line"
    d = json.loads(text)
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == "Mrays/s" and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert isinstance(d["config"].get("workload"), str) and "model" not in d["config"]
    assert d["n_gpus"] >= 1 and d["steps"] >= 1 and d["ms_per_step"] > 0
    # value = rays per frame / ms per step
    rays = d["config"]["rays_per_frame"]
    assert abs(d["value"] - rays / d["ms_per_step"] / 1e3) <= 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    if round_no(path) <= 5:
        # rounds 3-5: achieved = algorithmic bytes per launch / launch period (the frame period of overlapping launches)
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_period_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    else:
        # round 6 on: `frac` is the KERNEL's own fraction -- bytes / the duration of one launch alone, the figure rocprofv3's
        # per-kernel average of a --no-pipeline run reproduces -- and the period-based figure rides beside it
        for k in ("frac_streamed_period", "achieved_streamed_period", "kernel_ms_one_at_a_time", "rate_basis"):
            assert k in r, k
        basis_ms = r["kernel_ms_one_at_a_time"] or r["kernel_ms"]
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (basis_ms * 1e-3) / 1e9) <= 0.01 * r["achieved"]
        assert abs(r["achieved_streamed_period"] - r["algorithmic_bytes_per_launch"] / (r["launch_period_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved_streamed_period"]
        assert abs(r["frac_streamed_period"] - r["achieved_streamed_period"] / r["peak"]) < 1e-4
        if r["kernel_ms_one_at_a_time"]:
            assert r["frac_one_at_a_time"] == r["frac"]
    assert r["achieved"] < r["peak"]
    if r.get("kernel_ms_one_at_a_time"):
        f = r["algorithmic_bytes_per_launch"] / (r["kernel_ms_one_at_a_time"] * 1e-3) / 1e9 / r["peak"]
        assert abs(r["frac_one_at_a_time"] - f) <= 0.01 * f
    # round 4 on: the single-frame rate BASELINE.json's config 2 names rides beside the streamed value, and the default (atrium) line carries
    # a short s256 leg
    if not os.path.basename(path).startswith("r03"):
        sf = d.get("single_frame")
        if sf is not None:
            assert abs(d["value_single_frame"] - rays / sf["single_frame_cold_ms"] / 1e3) <= 0.01 * d["value_single_frame"]
            assert d["value_single_frame"] <= d["value_single_frame_warm"] * 1.10  # (64x64 frames of 0.08 ms: cold and warm are the same thing, within noise)
            assert "value_is" in d["config"]
        if os.path.basename(path).endswith("_bench_atrium.json"):
            s2 = d["secondary"]["s256"]
            assert "error" not in s2, s2
            assert s2["rays_per_frame"] == 3840 * 2160 and s2["ms_per_step"] > 0
            f2 = s2["algorithmic_bytes_per_launch"] / (s2["kernel_ms_one_at_a_time"] * 1e-3) / 1e9 / 8000.0
            assert abs(s2["frac_one_at_a_time"] - f2) <= 0.01 * f2
            if round_no(path) >= 6:
                assert abs(s2["frac"] - f2) <= 0.01 * f2 and "frac_streamed_period" in s2
                # configs[4] in the driver's own line: the orbit and the relight loop, short legs
                for leg in ("orbit", "relight"):
                    sl = d["secondary"][leg]
                    assert "error" not in sl, sl
                    assert sl["rays_per_frame"] == 1920 * 1080 and sl["ms_per_step"] > 0 and sl["frames"] >= 60
                assert d["secondary"]["relight"]["light_ms_per_frame"] > 0
    cb = d.get("cpu_baseline")
    if cb is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, k
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
        if round_no(path) >= 6:  # the port at -O3 as SURVEY.md 8(d) wrote; the -O2 build of rounds 1-5 timed beside it
            assert cb["flags"].startswith("-O3") and ("value" in cb["o2"] or "error" in cb["o2"])


def test_roofline_object_is_the_kernels_own_fraction():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod_r", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # streamed, a launch timed alone: frac = bytes / that duration; the period-based figure beside it
    r = mod.roofline_object(400_000_000, 1.3, 0.42, 400e6 / 0.42e-3 / 1e9, True, 4, 0.62, None, None, 75_000_000)
    assert abs(r["achieved"] - 400e6 / 0.62e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-6
    assert r["frac_one_at_a_time"] == r["frac"] and abs(r["frac_streamed_period"] - 400e6 / 0.42e-3 / 1e9 / 8000.0) < 1e-6
    # one launch at a time: the timed region's launches ARE alone
    r = mod.roofline_object(400_000_000, 0.62, 0.62, 400e6 / 0.62e-3 / 1e9, False, 1, None, None, None, 75_000_000)
    assert r["kernel_ms_one_at_a_time"] == 0.62 and abs(r["frac"] - r["frac_streamed_period"]) < 1e-6
    # streamed with --no-extras: no launch was timed alone, the overlapping launches' own durations bound the rate from below
    r = mod.roofline_object(400_000_000, 1.3, 0.42, 400e6 / 0.42e-3 / 1e9, True, 4, None, None, None, 75_000_000)
    assert r["kernel_ms_one_at_a_time"] is None and abs(r["achieved"] - 400e6 / 1.3e-3 / 1e9) < 0.01 and "lower bound" in r["rate_basis"]


def test_bench_defaults_are_one_gpu_and_short():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--no-pipeline", "--frames-per-gather", "--in-flight"):
        assert flag in out.stdout, flag
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    src = open(spec.origin).read()
    assert '"--gpus", type=int, default=1' in src.replace("'", '"')
