#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/measure.sh) into profiles/<tag>_*.

  python tools/summarize_profile.py r01

Outputs: profiles/<tag>_kernel_stats_<workload>.csv (rocprofv3 --kernel-trace --stats summary),
profiles/<tag>_pmc_<workload>.json (per-launch means of every collected counter for the trace
kernel, plus the derived HBM traffic that bench.py reports as roofline.traffic), and the bench
JSON lines."""
import csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(d, suffix):
    hits = glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # gpurun merges into the old directory: take the newest run


for wl in ("atrium", "s256"):
    p = find(f"stats_{wl}", "kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"))
    p = find(f"stats_{wl}_nopipe", "kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(dst, f"{tag}_kernel_stats_{wl}_nopipe.csv"))
    pmc = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq1", "pmc_sq2", "pmc_sq3"):
        p = find(f"{d}_{wl}", "counter_collection.csv")
        if not p:
            continue
        acc = {}
        with open(p) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                # the timed variant only: <VOL, LMODE, DIAG=false>; the DIAG=true launch is the
                # untimed counter-collecting pass bench.py issues once after the timed region
                m = re.search(r"trace_image_kernel<[^,>]+,[^,>]+, *(true|false)", name)
                if not m or m.group(1) == "true":
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    if pmc:
        out = {"workload": wl, "kernel": "trace_image_kernel", "counters": pmc}
        if "FETCH_SIZE" in pmc:
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. MI355X_MICROARCH.md (HBM section):
            # on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads -> doubled as the guide
            # prescribes. Calibrated in round 4 on this kernel's own access patterns (tools/ubench/fetch_calib.hip,
            # profiles/r04_fetch_calibration.txt): 2-byte and 4-byte gathers also cost one 128-byte line per L2 miss,
            # reported at 64 bytes, so the x2 holds for them as well; WRITE_SIZE is exact for coalesced stores.
            fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024.0
            write = pmc.get("WRITE_SIZE", {"mean_per_launch": 0.0})["mean_per_launch"] * 1024.0
            out["hbm_traffic_bytes_per_launch"] = 2.0 * fetch + write
            out["fetch_bytes_raw"] = fetch
            out["write_bytes_raw"] = write
        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
            h, m = pmc["TCC_HIT_sum"]["mean_per_launch"], pmc["TCC_MISS_sum"]["mean_per_launch"]
            out["l2_hit_rate"] = h / (h + m) if h + m else None
        with open(os.path.join(dst, f"{tag}_pmc_{wl}.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    b = os.path.join(src, f"bench_{wl}.json")
    if os.path.exists(b):
        lines = [l for l in open(b).read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w") as f:
                f.write(lines[-1] + "\n")
p = find("stats_lightbench", "kernel_stats.csv")
if p:
    shutil.copy(p, os.path.join(dst, f"{tag}_kernel_stats_lightbench.csv"))
# the light kernel's counters (tools/measure_round.sh): per-launch means of compute_light_wave_kernel
pmc = {}
for d in ("pmc_light_sq1", "pmc_light_sq2", "pmc_light_fetch", "pmc_light_write"):
    sj = os.path.join(src, d, "pmc_summary.json")  # written on the box by tools/reduce_pmc_csv.py (the CSV itself may be too large to keep)
    if os.path.exists(sj):
        for kern, ctrs in json.load(open(sj)).items():
            if "compute_light_wave_kernel" in kern and "dense" not in kern:
                pmc.update(ctrs)
        continue
    p = find(d, "counter_collection.csv")
    if not p:
        continue
    acc = {}
    with open(p) as f:
        for row in csv.DictReader(f):
            if "compute_light_wave_kernel" in row.get("Kernel_Name", ""):
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
if pmc:
    out = {"workload": "light-bench (light_bench_space, batches of 32)", "kernel": "compute_light_wave_kernel", "counters": pmc}
    if "FETCH_SIZE" in pmc:
        out["fetch_bytes_raw"] = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024.0
        out["write_bytes_raw"] = pmc.get("WRITE_SIZE", {"mean_per_launch": 0.0})["mean_per_launch"] * 1024.0
        out["hbm_traffic_bytes_per_launch"] = 2.0 * out["fetch_bytes_raw"] + out["write_bytes_raw"]
    with open(os.path.join(dst, f"{tag}_pmc_lightbench.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
for name in ("lightbench", "relight", "orbit"):
    b = os.path.join(src, f"bench_{name}.json")
    if os.path.exists(b):
        lines = [l for l in open(b).read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"{tag}_bench_{name}.json"), "w") as f:
                f.write(lines[-1] + "\n")
# the text profiles of tools/measure_round.sh's `profile` and `bench` phases: the measured lines are replaced, the commentary of the file in profiles/
# ('#' lines: those ahead of the first measured line stay ahead, the rest follow) is kept -- and has to be read again against the new numbers by whoever runs this
for raw, name, default_header in (
        ("prof.txt", "phase_cycles", "# In-kernel phase counters of trace_image_kernel<true,2,false,false> (-DAIC_PROFILE build; tools/measure_round.sh <tag> profile), one cold frame per workload"),
        ("wave_tail.txt", "wave_tail", "# Per-wave clocks of one frame (tools/wave_tail.py, tools/wave_rays.py on the -DAIC_PROFILE -DAIC_RAY_PROF build)"),
        ("rank_share.txt", "rank_share", "# One rank's share of the C2 frame traced on one GPU (tools/rank_share.py <n_parts> <frames in flight>)"),
        ("hip_handoff.txt", "hip_handoff", "# bench.py --gather-at-one under rocprofv3 --hip-trace --stats (tools/measure_round.sh <tag> bench; tools/hip_handoff_summary.py)")):
    rp = os.path.join(src, raw)
    if not os.path.exists(rp):
        continue
    body = [l for l in open(rp).read().splitlines() if l.strip() and not l.startswith("#")]
    out_path = os.path.join(dst, f"{tag}_{name}.txt")
    head, tail = [], []
    if os.path.exists(out_path):
        seen_body = False
        for l in open(out_path).read().splitlines():
            if l.startswith("#"):
                (tail if seen_body else head).append(l)
            elif l.strip():
                seen_body = True
    if not head:
        head = [default_header]
    with open(out_path, "w") as f:
        f.write("\n".join(head + body + tail) + "\n")
if os.path.exists(os.path.join(src, "issue_rate.txt")):
    shutil.copy(os.path.join(src, "issue_rate.txt"), os.path.join(dst, f"{tag}_issue_rate.txt"))
print(sorted(os.listdir(dst)))
