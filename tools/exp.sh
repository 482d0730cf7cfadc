#!/bin/bash
# quick experiment loop on the GPU box: parity tests, then the headline bench for the default build
# and for every prebuilt variants/libaic_hip_<name>.so (tools/build_variants.sh)
run_bench() {
for wl in atrium s256; do
python bench.py --workload $wl --steps ${AIC_STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 $wl', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['gsteps_per_s'], d['config']['steps_per_ray'])"
done
}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run_bench default
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in variants/libaic_hip_*.so; do
  [ -f "$v" ] || continue
  cp "$v" all_is_cubes_amd/libaic_hip.so
  run_bench "$(basename $v .so | sed s/libaic_hip_//)"
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
if [ -n "$AIC_WITH_PROF" ]; then AIC_EXTRA="$AIC_WITH_PROF" bash tools/prof.sh; fi
