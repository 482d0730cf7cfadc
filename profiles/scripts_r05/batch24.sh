#!/bin/bash
# round 5, GPU batch 24: the light walk's kept pointers typed as global memory (15 of its 17 FLAT instructions become global ones) -- the light suite, 1500 light fuzz
# seeds, then light_bench_space in the reference's order against the library before (variants/libaic_hip_prev.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py tests/test_gpu_goldens2.py -m gpu -x -q 2>&1 | tail -1
AIC_LIGHT_FUZZ_N=1500 timeout 600 python -X faulthandler -m pytest tests/test_gpu_light_update.py -m gpu -x -q -k fuzz 2>&1 | tail -1
LB="python bench.py --workload light-bench --steps 60 --warmup 5 --no-cpu-baseline --no-extras"
one() { python - "$1" "$2" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).readlines()[-1])
lu = d.get("light_update", {})
print(sys.argv[2], "ms/frame", d["ms_per_step"], "light:", {k: lu.get(k) for k in ("total_ms", "device_ms", "launches", "updates")}, {k: d[k] for k in d if "reference_order" in k})
PY
}
run() { timeout 300 $LB > $O/$1.json 2> $O/$1.err; one $O/$1.json "$1" || tail -3 $O/$1.err; }
run new1
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_prev.so all_is_cubes_amd/libaic_hip.so
run prev1; run prev2
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
run new2
grep -o '"light[^}]*}' $O/new2.json | head -3
