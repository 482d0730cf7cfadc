#!/bin/bash
# round 5, GPU batch 10: production variants with and without the exchange in one library, chosen per frame (DevFrame::exchange: >= 5 tiles per resident wave);
# the suite, a rank's share at N = 2 / 4 / 8 (C2) and at N = 8 (C3) by both variants, the bench lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
{
for np in 2 4 8; do
  for d in 1 8; do timeout 120 python tools/rank_share.py $np $d atrium 2>&1 | grep "^atrium"; done
done
echo "# the exchanging variant for every frame (AIC_XCHG_TILES=0)"
for np in 2 8; do
  for d in 1 8; do AIC_XCHG_TILES=0 timeout 120 python tools/rank_share.py $np $d atrium 2>&1 | grep "^atrium"; done
done
echo "# C3, an eighth of the frame"
for d in 1 8; do timeout 200 python tools/rank_share.py 8 $d s256 2>&1 | grep "^s256"; done
for d in 1 8; do AIC_XCHG_TILES=0 timeout 200 python tools/rank_share.py 8 $d s256 2>&1 | grep "^s256" | sed 's/$/ (exchanging variant)/'; done
} | tee $O/rank_share.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
timeout 200 $B --steps 40 --warmup 5 > $O/atrium_p.json 2> $O/atrium_p.err; one $O/atrium_p.json "default atrium pipe"
timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/s256_p.json 2> $O/s256_p.err; one $O/s256_p.json "default s256 pipe"
timeout 300 $B --workload orbit --steps 60 --warmup 5 > $O/orbit.json 2> $O/orbit.err; one $O/orbit.json "default orbit"
