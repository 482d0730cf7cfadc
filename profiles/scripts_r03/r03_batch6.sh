#!/bin/bash
# Round 3, GPU batch 6: second sweep of fast steps x full passes per trip (same box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b6; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --min-seconds 1"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'warm/cold', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
bench3() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 200 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in f2r4 f2r3 f3r3 f4r3 f4r2 f6r2 f2r5 f3r5; do
  cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  timeout 120 python tools/check_frame_hash.py atrium 1 2>&1 | tail -1 | sed "s/^/$v /"
  bench3 $v
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
