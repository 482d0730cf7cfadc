for cfg in "32 8" "40 8" "48 8" "32 4" "40 16"; do set -- $cfg
make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_T_BATCH=$1 -DAIC_N_FEW=$2" >/dev/null 2>&1
echo "== T=$1 few=$2"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"
done
make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_PROFILE" >/dev/null 2>&1
python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep PROF | tail -12
