cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --no-secondary --no-extras --steps 40 --warmup 5 --no-pipeline"
one() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" 2>/dev/null || echo "$1 FAILED"; }
for t in "" 8 10 12 16 24; do AIC_TILES_PER_WAVE=$t timeout 200 $B 2>/dev/null | one "atrium alone tiles/wave ${t:-default}"; done
for t in "" 32 40 48 64; do AIC_TILES_PER_WAVE=$t timeout 300 $B --workload s256 --steps 8 --warmup 2 2>/dev/null | one "s256 alone tiles/wave ${t:-default}"; done
