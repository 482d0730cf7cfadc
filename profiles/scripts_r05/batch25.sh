#!/bin/bash
# round 5, GPU batch 25: a rank's share of the C2 frame at N = 4 / 8 (2 / 1 tiles per resident wave), streamed, with the plain variant (the launcher's choice below 3 tiles
# per wave) and with the exchanging variant forced (AIC_XCHG_TILES=0) -- on the trimmed scheduler round
cd ${GRAFT_REPO_ROOT:-/root/repo}
for np in 4 8; do
  python tools/rank_share.py $np 8 atrium 2>&1 | grep -v amdgpu
  AIC_XCHG_TILES=0 python tools/rank_share.py $np 8 atrium 2>&1 | grep -v amdgpu | sed 's/$/ (exchanging variant)/'
done
