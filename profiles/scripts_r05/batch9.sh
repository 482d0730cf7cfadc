#!/bin/bash
# round 5, GPU batch 9: the new full-size parity tests and the driver's N = 8 command on one GPU; a rank's share with parking switched off for small frames
# (DevFrame::xchg_park: from 3 tiles per resident wave) against parking always (AIC_XCHG_PARK_TILES=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_production or bounce_frame or eight_ranks or two_ranks" --durations=6 > $O/pytest_new.log 2>&1; tail -12 $O/pytest_new.log
{
for np in 2 4 8; do
  for d in 1 8; do timeout 120 python tools/rank_share.py $np $d atrium 2>&1 | grep "^atrium"; done
done
echo "# parking always (AIC_XCHG_PARK_TILES=0)"
for np in 4 8; do
  for d in 1 8; do AIC_XCHG_PARK_TILES=0 timeout 120 python tools/rank_share.py $np $d atrium 2>&1 | grep "^atrium"; done
done
} | tee $O/rank_share.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
timeout 200 $B --steps 40 --warmup 5 > $O/atrium_p.json 2> $O/atrium_p.err; one $O/atrium_p.json "default atrium pipe"
timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/s256_p.json 2> $O/s256_p.err; one $O/s256_p.json "default s256 pipe"
