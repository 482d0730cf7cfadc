mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "3 24 8" "2 24 8" "2 32 8" "4 24 8"; do
  set -- $cfg
  make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_MIN_WAVES=$1 -DAIC_T_BATCH=$2 -DAIC_N_FEW=$3" >/dev/null 2>&1
  echo "== waves=$1 T=$2 few=$3"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['gsteps_per_s'])"
done
