#!/bin/bash
# experiment loop on the GPU box: for every prebuilt variants/libaic_hip_<name>.so (tools/build_variants.sh) run the
# parity suite (fast) and the bench workloads.  AIC_EXP_TESTS=0 skips the tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run_bench() {
for wl in atrium s256; do
  st=40; [ $wl = s256 ] && st=8
  for mode in "--no-pipeline" ""; do
    python bench.py --workload $wl --steps $st --warmup 3 --no-cpu-baseline $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 $wl ${mode:-pipelined}', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" || echo "$1 $wl $mode FAILED"
  done
done
}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in variants/libaic_hip_*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so | sed s/libaic_hip_//)
  cp "$v" all_is_cubes_amd/libaic_hip.so
  if [ "${AIC_EXP_TESTS:-1}" != 0 ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -1 | sed "s/^/$n tests: /"; fi
  run_bench $n
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
