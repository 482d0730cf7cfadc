#!/bin/bash
# Round 3, GPU batch 7: event batching thresholds re-swept on the fast-step kernel (default: 3 fast steps, 3 passes per trip).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b8; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --min-seconds 1"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'warm/cold', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
bench3() {
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 200 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo sweep2
bench3 f3r3
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in t32n24 t40n16 t48n16 t32n8 t48n24 t32n20 t36n16; do
  cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  timeout 120 python tools/check_frame_hash.py atrium 1 2>&1 | tail -1 | sed "s/^/$v /"
  bench3 $v
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
