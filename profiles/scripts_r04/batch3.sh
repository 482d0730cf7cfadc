#!/bin/bash
# Round 4, GPU batch 3: the parity suite (Bounce proper, device-side hand-off), a 300-seed fuzz, the default bench line with its s256 leg,
# and the exchange step over the nccl backend on one GPU (--gather-at-one): device-side and host-side hand-off.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest3.log 2>&1; tail -3 $O/pytest3.log )
( AIC_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz3.log 2>&1; tail -2 $O/fuzz3.log )
timeout 600 python bench.py --no-cpu-baseline > $O/b3_default.json 2> $O/b3_default.err; tail -c 1500 $O/b3_default.json; echo
for mode in "" "--host-handoff"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --gather-at-one $mode --steps 30 --warmup 3 > $O/b3_gather1$mode.json 2> $O/b3_gather1$mode.err
  python -c "
import json
d=json.loads(open('$O/b3_gather1$mode.json').read().strip().splitlines()[-1]); print('gather-at-one $mode', d['ms_per_step'], d['config'].get('handoff'), d['config'].get('assembled_frame_equals_single_rank_frame'))" || tail -5 $O/b3_gather1$mode.err
done
for wl in atrium s256; do timeout 200 python tools/check_frame_hash.py $wl 3; done 2>&1 | tee $O/hash3.txt
timeout 300 python bench.py --no-cpu-baseline --workload s256 --steps 8 --warmup 2 > $O/b3_s256.json 2> $O/b3_s256.err
python -c "
import json
for f in ('b3_default','b3_s256'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); s=d.get('single_frame') or {}
    print(f, 'streamed', d['ms_per_step'], 'warm', s.get('single_frame_warm_ms'), 'cold', s.get('single_frame_cold_ms'), 'kernel_warm', s.get('kernel_ms_warm'))"
