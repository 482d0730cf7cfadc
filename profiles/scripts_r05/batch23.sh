#!/bin/bash
# round 5, GPU batch 23: lvl_init without early exits at the image kernel's two call sites (ENTER, RAY events): the GPU suite, frame hashes, 600 fuzz seeds, then against
# the library before (variants/libaic_hip_prev.so = the second final pass's)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b23; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  for k in 1 2; do timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p$k.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p$k.json "$1 atrium pipe"; done
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
}
timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1; timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
AIC_FUZZ_N=600 timeout 900 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1
run_bench new
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_prev.so all_is_cubes_amd/libaic_hip.so
run_bench prev
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
run_bench new
