#!/bin/bash
# Round 6: a sweep of library variants on one box -- for each name (variants/libaic_hip_<name>.so, tools/build_variants.sh) the frame hash of both workloads and the
# two streamed figures; `default` = the tree's library. usage (GPU box): bash profiles/scripts_r06/sweep.sh <tag> name1 name2 ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for name in "$@"; do
  if [ $name = default ]; then cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so; else cp variants/libaic_hip_$name.so all_is_cubes_amd/libaic_hip.so; fi
  h1=$(timeout 300 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | awk '{print $4}'); h2=$(timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1 | awk '{print $4}')
  a=$(timeout 200 $B --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  a2=$(timeout 200 $B --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  s=$(timeout 300 $B --workload s256 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  n=$(timeout 200 $B --steps 40 --warmup 5 --no-pipeline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$name  hashes $h1 $h2  C2 streamed $a $a2  C3 streamed $s  C2 alone $n" | tee -a $O/sweep.txt
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
