#!/bin/bash
# a rank's share of the C2 frame on one GPU: one launch per share (round 5) against k shares per launch (aic_render_submit_batch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for np in 2 4 8; do
  python tools/rank_share.py $np 8 atrium 1 2>&1 | grep -v amdgpu
  for k in 2 4 8; do
    fl=$((8 / k)); [ $fl -lt 2 ] && fl=2
    python tools/rank_share.py $np $fl atrium $k 2>&1 | grep -v amdgpu
  done
done
python tools/rank_share.py 1 4 atrium 1 2>&1 | grep -v amdgpu | head -1
python tools/rank_share.py 1 2 atrium 2 2>&1 | grep -v amdgpu | head -1
python tools/rank_share.py 1 2 atrium 4 2>&1 | grep -v amdgpu | head -1
python tools/rank_share.py 8 2 s256 8 2>&1 | grep -v amdgpu | head -1
python tools/rank_share.py 8 8 s256 1 2>&1 | grep -v amdgpu | head -1
