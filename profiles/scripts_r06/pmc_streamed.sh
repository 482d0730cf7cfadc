cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06y; rm -rf $O; mkdir -p $O
for mode in "" "4"; do
X="--workload atrium --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-extras --min-seconds 0.5"
AIC_TILES_PER_WAVE=$mode rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq1_$mode -- python bench.py $X > /dev/null 2>&1
AIC_TILES_PER_WAVE=$mode rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/sq2_$mode -- python bench.py $X > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for d in ("sq1_$mode", "sq2_$mode"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_image_kernel" in r["Kernel_Name"] and ", false, false, " in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("tiles/wave '$mode' (''=the part-grid rule, 4=the old sizing): launches", len(acc["SQ_INSTS_VALU"]), {k: round(v / 1e6, 2) for k, v in m.items()})
print("   lane utilisation %.3f  wait_any/wave_cycles %.3f  VALU per launch %.1f M" % (m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]), m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"] / 1e6))
PY
done
find $O -type f -size +4M -delete
