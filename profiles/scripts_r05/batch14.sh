#!/bin/bash
# round 5, GPU batch 14: the frame's tail -- a dry wave with few rays left retires (parks all of them and ends): AIC_XCHG_RETIRE = the most rays it parks,
# AIC_XCHG_DRY_GAIN = a dry wave's minimum top-up. Parity first (GPU suite + frame hashes with the rule on), then the sweep.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b14; mkdir -p $O
AIC_XCHG_RETIRE=16 AIC_XCHG_DRY_GAIN=1 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
AIC_XCHG_RETIRE=16 AIC_XCHG_DRY_GAIN=1 timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
AIC_XCHG_RETIRE=16 AIC_XCHG_DRY_GAIN=1 timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
AIC_XCHG_RETIRE=64 timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  for k in 1 2; do timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p$k.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p$k.json "$1 atrium pipe"; done
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
}
for cfg in "0 0" "16 0" "16 1" "32 1" "64 1" "0 0"; do
  set -- $cfg
  export AIC_XCHG_RETIRE=$1 AIC_XCHG_DRY_GAIN=$2
  run_bench "retire$1_gain$2"
done
