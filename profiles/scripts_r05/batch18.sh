#!/bin/bash
# round 5, GPU batch 18: the tail patch's other half alone -- a wave that has seen the queue dry tops up from the pool by any number of lanes (AIC_XCHG_DRY_GAIN = 1 / 2 / 4
# instead of 8), with no wave retiring (AIC_XCHG_RETIRE=0). variants/libaic_hip_tail.so = HEAD + profiles/scripts_r05/tail_retire.patch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b18; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  for k in 1 2; do timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np$k.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np$k.json "$1 atrium nopipe"; done
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
run_bench default
cp variants/libaic_hip_tail.so all_is_cubes_amd/libaic_hip.so
export AIC_XCHG_RETIRE=0
for g in 1 2 4; do
  export AIC_XCHG_DRY_GAIN=$g
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
  run_bench "drygain$g"
done
unset AIC_XCHG_DRY_GAIN AIC_XCHG_RETIRE
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
run_bench default
