"""Counts the HIP API calls of a `rocprofv3 --hip-trace --stats` run of `bench.py --gather-at-one` (tools/measure_round.sh, bench phase):
the evidence that the device-side hand-off leaves no per-frame host synchronisation between a trace and its gather."""
import csv, glob, json, sys
d, j, mode = sys.argv[1], sys.argv[2], sys.argv[3]
calls = {}
for p in glob.glob(d + "/**/*hip_api_stats.csv", recursive=True) + glob.glob(d + "/**/*hip_stats.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        calls[row.get("Name", "")] = (int(float(row.get("Calls", 0))), float(row.get("TotalDurationNs", 0)) / 1e6)
line = json.loads([l for l in open(j).read().splitlines() if l.startswith("{")][-1])
frames = line["steps"] * line["timed_regions"] + line["warmup"]
keep = {k: v for k, v in calls.items() if k in ("hipStreamSynchronize", "hipEventSynchronize", "hipStreamWaitEvent", "hipLaunchKernel", "hipEventRecord", "hipDeviceSynchronize")}
print(f"gather-at-one {mode or '(device hand-off)'}: {line['ms_per_step']} ms/frame, handoff = {line['config']['handoff']}, frames ~{frames}; HIP calls (count, total ms): {keep}")
