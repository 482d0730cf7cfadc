#!/bin/bash
# round 5, GPU batch 7: tuning sweep of the exchange policy (streamed figures only; every variant's frame hash is checked)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b7; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo "== default"; timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1; timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
run_bench default
timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/default_atrium_np.json 2> $O/default_atrium_np.err; one $O/default_atrium_np.json "default atrium nopipe"
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for n in x0 mg4 mg12 mg16 pm4 pm16 pool64 f40 f56 reps1 reps3 fast24; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/$n /"
  run_bench $n
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
