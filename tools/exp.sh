#!/bin/bash
# quick experiment loop on the GPU box: parity tests, then the headline bench; AIC_VARIANTS="-DX=1|-DX=2" rebuilds per variant
run_bench() {
for wl in atrium s256; do
python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 $wl', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['gsteps_per_s'], d['config']['steps_per_ray'])"
done
}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run_bench default
IFS='|' read -ra VARS <<< "$AIC_VARIANTS"
for v in "${VARS[@]}"; do
  make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $v" >/dev/null 2>&1 || echo "build failed: $v"
  run_bench "[$v]"
done
if [ -n "$AIC_WITH_PROF" ]; then AIC_EXTRA="$AIC_WITH_PROF" bash tools/prof.sh; fi
