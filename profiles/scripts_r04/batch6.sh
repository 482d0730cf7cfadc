#!/bin/bash
# Round 4, GPU batch 6: the trip's parameters swept again on the kernel without pending spans (streamed ms/frame, atrium / s256), and the
# FETCH_SIZE calibration with its u16 kernels fixed.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in default fs6 fs12 reps2 reps4 tb28 tb36 nf20 nf28 fm12 fm20 default; do
  [ $v = default ] && cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so || cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
  a=$(timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  s=$(timeout 200 python bench.py --no-cpu-baseline --no-extras --workload s256 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$v atrium $a s256 $s"
done 2>&1 | tee $O/sweep6.txt
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
timeout 900 bash tools/measure_fetch_calib.sh r04 2>&1 | tail -24
