#!/usr/bin/env python3
"""Joins tools/ubench/fetch_calib's pattern list with the rocprofv3 counter CSVs of tools/measure_fetch_calib.sh:
per access pattern, what FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) amount to per read and per distinct
64-byte sector / 128-byte line. The factor to apply to the trace kernel's raw FETCH_SIZE is the gather rows'."""
import csv, glob, os, re, sys

d = sys.argv[1]
pat = {}
with open(os.path.join(d, "calib_patterns.csv")) as f:
    for row in csv.DictReader(f):
        pat[row["pattern"]] = {k: int(v) for k, v in row.items() if k != "pattern"}
kern = {"stream16": "stream16", "dense2": "dense2", "gather2_stride<64>": "gather2_64", "gather2_stride<128>": "gather2_128",
        "gather2_stride<256>": "gather2_256", "gather2_rnd": "gather2_rnd", "gather4_rnd": "gather4_rnd", "store4": "store4"}


def counters(sub, name):
    out = {}
    for p in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != name:
                    continue
                k = row.get("Kernel_Name", "")
                for key, label in kern.items():
                    if k.startswith("void " + key) or k.startswith(key):
                        out.setdefault(label, []).append(float(row["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = counters("calib_fetch", "FETCH_SIZE"), counters("calib_write", "WRITE_SIZE")
print("# FETCH_SIZE / WRITE_SIZE (rocprofv3, gfx950) on known access patterns over a 4 GiB buffer, each touched once (tools/ubench/fetch_calib.hip)")
print(f"{'pattern':<12} {'reads':>12} {'B/read':>7} {'requested B':>14} {'64B sectors x64':>16} {'128B lines x128':>16} {'FETCH_SIZE B':>14} {'per read':>9} {'/ sectors x64':>14} {'/ lines x128':>13} {'WRITE_SIZE B':>14}")
for name, p in pat.items():
    req = p["reads"] * p["elem_bytes"]
    s64, l128 = p["distinct_64B_sectors"] * 64, p["distinct_128B_lines"] * 128
    fv, wv = fetch.get(name), write.get(name)
    cells = [f"{name:<12}", f"{p['reads']:>12}", f"{p['elem_bytes']:>7}", f"{req:>14}", f"{s64:>16}", f"{l128:>16}"]
    if fv is not None:
        cells += [f"{fv:>14.0f}", f"{fv / p['reads']:>9.2f}", f"{fv / s64:>14.3f}", f"{fv / l128:>13.3f}"]
    else:
        cells += [f"{'-':>14}", f"{'-':>9}", f"{'-':>14}", f"{'-':>13}"]
    cells += [f"{wv:>14.0f}" if wv is not None else f"{'-':>14}"]
    print(" ".join(cells))
raw = {}
for sub, names in (("calib_raw", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_MISS_sum", "TCC_HIT_sum")), ("calib_raw2", ("TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_READ_sum"))):
    for nm in names:
        got = {}
        for p in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(p) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != nm:
                        continue
                    k = row.get("Kernel_Name", "")
                    for key, label in kern.items():
                        if k.startswith("void " + key) or k.startswith(key):
                            got.setdefault(label, []).append(float(row["Counter_Value"]))
        if got:
            raw[nm] = {k: sum(v) / len(v) for k, v in got.items()}
if raw:
    print("\n# raw request counters per read (the same patterns); TCC_EA0_RDREQ = L2 -> fabric read requests, _32B = those of 32 bytes")
    names = list(raw)
    print(f"{'pattern':<12} " + " ".join(f"{n.replace('_sum', ''):>20}" for n in names))
    for name, p in pat.items():
        print(f"{name:<12} " + " ".join((f"{raw[n][name] / p['reads']:>20.4f}" if name in raw[n] else f"{'-':>20}") for n in names))
