#!/bin/bash
# round 5: the GPU suite and a 6000-seed sweep with the per-pixel step counts of the exchanging production variant checked per seed
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05verify; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
AIC_FUZZ_N=6000 timeout 1500 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz6000.log 2>&1; tail -2 $O/fuzz6000.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 3 2> $O/bench.err | tail -c 1500
