#!/bin/bash
# round 5, GPU batch 15: issue priority by phase (s_setprio at the start of a stepping trip / of an event phase): step 3 event 0, step 0 event 3, step 1 event 0,
# against the committed library (no s_setprio). Data are untouched: frame hashes checked once per variant all the same.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b15; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  for k in 1 2; do timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p$k.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p$k.json "$1 atrium pipe"; done
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
run_bench default
for v in prio30 prio03 prio10; do
  cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
  run_bench $v
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
run_bench default
