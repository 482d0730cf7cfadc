#!/bin/bash
# Round 4, GPU batch 2: the parity suite (Bounce proper, device-side hand-off), a 300-seed fuzz, the default bench line with its s256 leg,
# and the exchange step over the nccl backend on one GPU (--gather-at-one): device-side and host-side hand-off.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log )
( AIC_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz2.log 2>&1; tail -2 $O/fuzz2.log )
timeout 600 python bench.py --no-cpu-baseline > $O/b2_default.json 2> $O/b2_default.err; tail -c 1500 $O/b2_default.json; echo
for mode in "" "--host-handoff"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --gather-at-one $mode --steps 30 --warmup 3 > $O/b2_gather1$mode.json 2> $O/b2_gather1$mode.err
  python -c "
import json
d=json.loads(open('$O/b2_gather1$mode.json').read().strip().splitlines()[-1]); print('gather-at-one $mode', d['ms_per_step'], d['config'].get('handoff'), d['config'].get('assembled_frame_equals_single_rank_frame'))" || tail -5 $O/b2_gather1$mode.err
done
