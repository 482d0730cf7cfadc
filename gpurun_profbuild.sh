make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_PROFILE" >/dev/null 2>&1
python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep PROF | head -12
