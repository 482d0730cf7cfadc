#!/bin/bash
# round 5, GPU batch 3: SoA cold state, decorrelated slot choice, full-wave fast path
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b3; mkdir -p $O
echo "== hashes (expected atrium d876fd8fde00ef83 74966856, s256 7912c59103550713 734379842)"
timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1
timeout 300 python tools/check_frame_hash.py s256 2>&1 | tail -1
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p.json 2> $O/$1_atrium_p.err; one $O/$1_atrium_p.json "$1 atrium pipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
}
echo "== bench"; run_bench default
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for n in x0 w256 w256mg8 full48; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python tools/check_frame_hash.py atrium 2>&1 | tail -1 | sed "s/^/$n /"
  run_bench $n
done
echo "== phase counters (profile build)"
cp variants/libaic_hip_prof.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== $wl"; timeout 300 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep PROF | tail -39; done > $O/prof.txt 2>&1
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
cat $O/prof.txt
echo "== phase counters (profile build, 256-thread workgroups)"
cp variants/libaic_hip_prof256.so all_is_cubes_amd/libaic_hip.so
for wl in atrium s256; do echo "== $wl"; timeout 300 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep PROF | tail -39; done > $O/prof256.txt 2>&1
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
cat $O/prof256.txt
