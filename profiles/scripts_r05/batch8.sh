#!/bin/bash
# round 5, GPU batch 8: the full GPU suite, the randomised sweeps on the exchanging kernels (6000 ray-trace seeds, 1500 light seeds), a rank's share of the C2 frame at
# N = 2 / 4 / 8 with up to 32 frames in flight (AIC_MAX_IN_FLIGHT 8 -> 32)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b8; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
AIC_FUZZ_N=6000 timeout 900 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz6000.log 2>&1; tail -2 $O/fuzz6000.log
AIC_LIGHT_FUZZ_N=1500 timeout 600 python -X faulthandler -m pytest tests/test_gpu_light_update.py -m gpu -x -q -k fuzz > $O/lightfuzz.log 2>&1; tail -2 $O/lightfuzz.log
{
for np in 2 4 8; do
  for d in 1 8 16 32; do timeout 120 python tools/rank_share.py $np $d atrium 2>&1 | grep "^atrium"; done
done
} | tee $O/rank_share.txt
