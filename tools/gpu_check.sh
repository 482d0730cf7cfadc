#!/bin/bash
# GPU-box check used while iterating on the kernel: parity suite, then the two bench workloads one frame at a time
# and streamed. Usage: gpurun -- 'bash tools/gpu_check.sh [tag] [pytest-args]'
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-chk}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python bench.py --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
timeout 300 $B --steps 40 --warmup 5 --no-pipeline > $O/atrium_np.json 2> $O/atrium_np.err; one $O/atrium_np.json atrium_nopipe
timeout 300 $B --steps 40 --warmup 5 > $O/atrium_p.json 2> $O/atrium_p.err; one $O/atrium_p.json atrium_pipe
timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/s256_np.json 2> $O/s256_np.err; one $O/s256_np.json s256_nopipe
