#!/bin/bash
# round 5, GPU batch 6: the swap on the round's common path (no register copies); 512- and 256-thread workgroups, the full-wave threshold; SQ instruction counts
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b6; mkdir -p $O
echo "== hashes (expected atrium d876fd8fde00ef83 74966856, s256 7912c59103550713 734379842)"
timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo "== bench"; run_bench default
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for n in x0 w256 w256f48 w256f64; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/$n /"
  run_bench $n
done
for n in w256; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  for W in atrium s256; do
    X="--no-pipeline --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0"
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq1_${n}_$W -- python bench.py $X > /dev/null 2>&1
    rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 --output-format csv -d $O/sq2_${n}_$W -- python bench.py $X > /dev/null 2>&1
  done
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
find $O -type f -size +4M -delete
python - <<PY
import csv, glob, collections
for n in ("w256",):
  for W in ("atrium", "s256"):
    acc = collections.defaultdict(list)
    for d in ("sq1", "sq2"):
        for f in glob.glob("$O/%s_%s_%s/**/*counter_collection.csv" % (d, n, W), recursive=True):
            for r in csv.DictReader(open(f)):
                if "trace_image_kernel" in r["Kernel_Name"] and ", false, false>" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(n, W, {k: "%.4g" % (sum(v) / max(1, len(v))) for k, v in sorted(acc.items())})
PY
