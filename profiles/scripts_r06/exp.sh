#!/bin/bash
# Round 6: ONE parameterised GPU batch script (round 5 left 25 batchNN.sh behind). Run on the GPU box:
#   gpurun --timeout T -- 'bash profiles/scripts_r06/exp.sh <tag> <step> [<step> ...]'
# Steps (each prints a few lines; raw files go to gpurun_out/<tag>/):
#   tests            the whole GPU suite (pytest -m gpu -x -q)
#   fuzz:N           N seeds of tests/test_gpu_fuzz.py (both production variants per seed)
#   hash             frame hash + step total of both full-size workloads (tools/check_frame_hash.py): the round's reference values are
#                    atrium d876fd8fde00ef83, s256 7912c59103550713
#   bench[:name]     the four figures (C2 / C3, streamed / one at a time) of the library in place, labelled `name`
#   lib:name         swap variants/libaic_hip_<name>.so in (tools/build_variants.sh); lib:default swaps the tree's own library back
#   pmc:workload     the quick counter passes of tools/pmc_quick.sh for one workload
#   cmd:...          any command (underscores for spaces are NOT translated: quote the whole step)
# The table of this round's invocations is in profiles/r06_experiments.txt.
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-extras"
one() { python -c "import sys,json; d=json.loads(open('$1').readlines()[-1]); print('$2', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || (echo "$2 FAILED"; tail -3 ${1%.json}.err); }
run_bench() {
  for k in 1 2; do timeout 200 $B --steps 40 --warmup 5 > $O/$1_atrium_p$k.json 2> $O/$1_atrium_p$k.err; one $O/$1_atrium_p$k.json "$1 atrium pipe"; done
  timeout 300 $B --workload s256 --steps 8 --warmup 2 > $O/$1_s256_p.json 2> $O/$1_s256_p.err; one $O/$1_s256_p.json "$1 s256 pipe"
  timeout 200 $B --steps 40 --warmup 5 --no-pipeline > $O/$1_atrium_np.json 2> $O/$1_atrium_np.err; one $O/$1_atrium_np.json "$1 atrium nopipe"
  timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/$1_s256_np.json 2> $O/$1_s256_np.err; one $O/$1_s256_np.json "$1 s256 nopipe"
}
[ -f /tmp/libaic_default.so ] || cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
LIB=default
for step in "$@"; do
  case "$step" in
    tests) timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > $O/pytest_$LIB.log 2>&1; echo "[$LIB] tests: $(tail -1 $O/pytest_$LIB.log)"; grep -E "^(FAILED|ERROR)" $O/pytest_$LIB.log | head -5 ;;
    fuzz:*) n=${step#fuzz:}; AIC_FUZZ_N=$n timeout 1500 python -X faulthandler -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_$LIB.log 2>&1; echo "[$LIB] fuzz $n: $(tail -1 $O/fuzz_$LIB.log)" ;;
    hash) for w in atrium s256; do echo "[$LIB] hash $(timeout 300 python tools/check_frame_hash.py $w 2>&1 | tail -1)"; done ;;
    bench) run_bench $LIB ;;
    bench:*) run_bench ${step#bench:} ;;
    lib:default) cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so; LIB=default ;;
    lib:*) LIB=${step#lib:}; cp variants/libaic_hip_$LIB.so all_is_cubes_amd/libaic_hip.so ;;
    pmc:*) bash tools/pmc_quick.sh $TAG ${step#pmc:} 2>&1 | tail -12 ;;
    cmd:*) eval "${step#cmd:}" ;;
    *) echo "unknown step $step" ;;
  esac
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
find "$O" -type f -size +4M -delete
