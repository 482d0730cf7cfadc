#!/usr/bin/env python3
"""Reduce rocprofv3 counter_collection.csv files under a directory to <dir>/pmc_summary.json: per kernel name and counter, the mean
per launch and the launch count. Run on the GPU box before large CSVs are dropped (gpurun merges at most 64 MiB back):
    python tools/reduce_pmc_csv.py gpurun_out/r03/pmc_light_sq1 [more dirs]"""
import csv, glob, json, os, sys

for d in sys.argv[1:]:
    acc = {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as f:
            for row in csv.DictReader(f):
                k = (row.get("Kernel_Name", ""), row["Counter_Name"])
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
    out = {}
    for (kern, ctr), (tot, n) in acc.items():
        out.setdefault(kern, {})[ctr] = {"mean_per_launch": tot / n, "launches": n}
    with open(os.path.join(d, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(d, len(out), "kernels")
