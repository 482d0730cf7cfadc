#!/bin/bash
# round 5, GPU batch 4 (timing experiment): cold loads the CU's L1 may serve (AIC_COLD_SCOPE = wavefront: frames may be WRONG) -- what do the sc0 loads cost?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b4; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for n in w256 w256plain; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/$n /"
  run_bench $n
done
cp variants/libaic_hip_prof256plain.so all_is_cubes_amd/libaic_hip.so
for wl in atrium; do echo "== $wl"; timeout 300 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep PROF | tail -39; done > $O/prof256plain.txt 2>&1
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
grep "cyc_\|xchg" $O/prof256plain.txt
