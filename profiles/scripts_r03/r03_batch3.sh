#!/bin/bash
# Round 3, GPU batch 3: ray migration in the frame's tail -- parity (full suite + fuzz), frame hashes equal with the
# migration off and on, and the timing sweep over the hand-over threshold.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b3; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); sf=d.get('single_frame',{}); print('$2', 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'warm/cold/moving', sf.get('single_frame_warm_ms'), sf.get('single_frame_cold_ms'), sf.get('single_frame_moving_camera_ms'))" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
echo "== frame hashes"
for k in 0 4 16 40; do for wl in atrium s256 small; do AIC_MIGRATE_K=$k timeout 120 python tools/check_frame_hash.py $wl 2>&1 | tail -1 | sed "s/^/K=$k /"; done; done
echo "== tests (K=16 default)"
AIC_FUZZ_N=300 timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
echo "== tests (K=40)"
AIC_MIGRATE_K=40 AIC_FUZZ_N=100 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest40.log 2>&1; tail -3 $O/pytest40.log
echo "== sweep"
for k in 0 8 16 24 32 48; do
  AIC_MIGRATE_K=$k timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/k${k}_atrium_np.json 2> $O/k${k}_atrium_np.err; one $O/k${k}_atrium_np.json "K=$k atrium nopipe"
  AIC_MIGRATE_K=$k timeout 200 $B --steps 40 --warmup 5 > $O/k${k}_atrium_p.json 2> $O/k${k}_atrium_p.err; one $O/k${k}_atrium_p.json "K=$k atrium pipe"
  AIC_MIGRATE_K=$k timeout 200 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/k${k}_s256_np.json 2> $O/k${k}_s256_np.err; one $O/k${k}_s256_np.json "K=$k s256 nopipe"
done
