#!/bin/bash
# The part-grid rule for streamed frames (aic_abi.cpp submit_frames) against round 5-6's fixed four tiles per wave (AIC_TILES_PER_WAVE=4: a whole frame then takes the whole chip)
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-secondary --no-extras --steps 40 --warmup 5"
one() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || echo "$1 FAILED"; }
for fl in 2 3 4 6 8; do for t in "" 4; do AIC_TILES_PER_WAVE=$t timeout 200 $B --in-flight $fl 2>/dev/null | one "atrium in-flight $fl rule ${t:-new}"; done; done
for fl in 4 8; do for t in "" 4; do AIC_TILES_PER_WAVE=$t timeout 300 $B --workload s256 --steps 8 --warmup 2 --in-flight $fl 2>/dev/null | one "s256 in-flight $fl rule ${t:-new}"; done; done
for w in orbit relight; do for t in "" 4; do AIC_TILES_PER_WAVE=$t timeout 300 $B --workload $w --steps 60 2>/dev/null | one "$w rule ${t:-new}"; done; done
for t in "" 4; do echo "rule ${t:-new}"; AIC_TILES_PER_WAVE=$t python tools/rank_share.py 2 4 atrium 1 2>&1 | grep -v amdgpu; done
python tools/check_frame_hash.py atrium | tail -1
timeout 1200 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | tail -2
