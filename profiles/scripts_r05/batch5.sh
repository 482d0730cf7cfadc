#!/bin/bash
# round 5, GPU batch 5: same-source baseline of the phase counters (profile build without the exchange) and SQ instruction counts, exchange off / on (256-thread workgroups)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b5; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_profx0.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== $wl"; timeout 300 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep PROF | tail -39; done > $O/profx0.txt 2>&1
grep "==\|cyc_\|_ph\|_ln\|iters\|lanes\|trips" $O/profx0.txt
for n in x0 w256; do
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
for n in ("x0", "w256"):
  for W in ("atrium", "s256"):
    acc = collections.defaultdict(list)
    for d in ("sq1", "sq2"):
        for f in glob.glob("$O/%s_%s_%s/**/*counter_collection.csv" % (d, n, W), recursive=True):
            for r in csv.DictReader(open(f)):
                if "trace_image_kernel" in r["Kernel_Name"] and ", false, false>" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # per dispatch: values are per (dispatch, dimension) rows; report the mean per dispatch = total / number of dispatches (rows of SQ_WAVES / dims)
    print(n, W, {k: "%.4g" % (sum(v) / max(1, len(v))) for k, v in sorted(acc.items())}, "rows", {k: len(v) for k, v in acc.items()}.get("SQ_WAVES"))
PY
