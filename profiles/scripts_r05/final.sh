#!/bin/bash
# round 5, final pass on the final library: the GPU suite, the randomised sweeps (both production variants per seed), then the round's measurement recipe
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05final; mkdir -p $O
AIC_FUZZ_N=6000 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz6000.log 2>&1; tail -2 $O/fuzz6000.log
AIC_LIGHT_FUZZ_N=1500 timeout 600 python -X faulthandler -m pytest tests/test_gpu_light_update.py -m gpu -x -q -k fuzz > $O/lightfuzz.log 2>&1; tail -1 $O/lightfuzz.log
bash tools/measure_round.sh r05 counters 2>&1 | tail -4
bash tools/measure_round.sh r05 bench 2>&1 | tail -3
bash tools/measure_round.sh r05 profile > $O/profile.out 2>&1; tail -3 $O/profile.out
