#!/bin/bash
# Round 3, GPU batch 5: bookkeeping-free steps ahead of each full stepping pass (AIC_FAST_STEPS) -- parity of the default
# build (2), then the same-box sweep over the number of fast steps and of passes per trip; relight drain settings.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b5; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --min-seconds 1"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'warm/cold', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
bench3() {
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline --no-extras > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 200 $B --workload s256 --steps 8 --warmup 2 --no-extras > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo "== tests (default: 2 fast steps)"
AIC_FUZZ_N=300 AIC_LIGHT_FUZZ_N=16 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
echo "== hashes"; for wl in atrium s256; do timeout 120 python tools/check_frame_hash.py $wl 2>&1 | tail -1; done
echo "== sweep"
bench3 f2
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in f0 f1 f3 f2r4 f2r6 f1r6 f3r4; do
  cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  timeout 120 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/$v /"
  bench3 $v
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
echo "== relight"
for cfg in "30 2048" "30 1024" "60 1024" "16 2048"; do set -- $cfg
timeout 300 $B --workload relight --steps 120 --warmup 10 --relight-period $1 --light-budget $2 > $O/relight_$1_$2.json 2> $O/relight_$1_$2.err; python -c "
import json
d=json.loads(open('$O/relight_$1_$2.json').readlines()[-1]); print('period $1 budget $2', d['ms_per_step'], json.dumps(d['relight']))" || tail -3 $O/relight_$1_$2.err
done
