#!/bin/bash
# round 5, GPU batch 12: the ray origin fetched in SHADE only where intersection_point uses it; the cold state as one 64-byte record per column (aos) against six
# arrays per workgroup (default): frame times and HBM-side traffic (C3: r04 4.0 GB, the first exchanging build 18 GB)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b12; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
  X="--no-pipeline --workload s256 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$1 -- python bench.py $X > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$1 -- python bench.py $X > /dev/null 2>&1
  python - <<PY
import csv, glob
for d in ("fetch", "write"):
    v = [float(r["Counter_Value"]) for f in glob.glob("$O/%s_$1/**/*counter_collection.csv" % d, recursive=True) for r in csv.DictReader(open(f)) if "trace_image_kernel" in r["Kernel_Name"] and ", false, false, " in r["Kernel_Name"]]
    print("$1 s256", d, "KiB per launch (sum over instances / launches)", sum(v) / 3 if v else None, len(v))
PY
}
echo "== default"; timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1; timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
run_bench default
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_aos.so all_is_cubes_amd/libaic_hip.so
timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/aos /"
run_bench aos
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
find $O -type f -size +4M -delete
