#!/bin/bash
# in-kernel phase counters / cycle split (AIC_PROFILE build), then restore the normal build
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_PROFILE $AIC_EXTRA" >/dev/null 2>&1
for wl in atrium s256; do echo "== $wl"; python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep PROF | tail -16; done
