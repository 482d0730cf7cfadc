#!/bin/bash
# Round 3, GPU batch 1: parity of the rebuilt SHADE / kernel-argument handling, A/B against round 2's trace kernel on the same
# box, the light kernel with LDS counters (fuzz + timing), the issue-rate micro-benchmark with pinned residency.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b1; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'warm/cold', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
bench3() {  # $1 = tag
  timeout 300 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 300 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo "== tests (new kernel)"
AIC_FUZZ_N=200 timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
echo "== bench new"
bench3 new
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
echo "== bench old (round 2 trace kernel)"
cp variants/libaic_hip_old.so all_is_cubes_amd/libaic_hip.so
bench3 old
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
echo "== light: default build"
timeout 300 $B --workload light-bench --steps 100 --warmup 10 > $O/lb_default.json 2> $O/lb_default.err; python -c "import json; d=json.loads(open('$O/lb_default.json').readlines()[-1]); print({k: d[k] for k in d if 'light' in k or 'criterion' in k})" 2>/dev/null | cut -c1-1500
echo "== light: LDS counters"
cp variants/libaic_hip_ldsc.so all_is_cubes_amd/libaic_hip.so
AIC_LIGHT_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_light_update.py tests/test_gpu_light.py -m gpu -x -q > $O/pytest_ldsc.log 2>&1; tail -3 $O/pytest_ldsc.log
timeout 300 $B --workload light-bench --steps 100 --warmup 10 > $O/lb_ldsc.json 2> $O/lb_ldsc.err; python -c "import json; d=json.loads(open('$O/lb_ldsc.json').readlines()[-1]); print({k: d[k] for k in d if 'light' in k or 'criterion' in k})" 2>/dev/null | cut -c1-1500
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
echo "== issue rate"
timeout 300 tools/ubench/issue_rate > $O/issue_rate.txt 2>&1; head -100 $O/issue_rate.txt
