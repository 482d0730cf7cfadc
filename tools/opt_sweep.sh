cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in "" "--lighting 1" "--lighting 0" "--transparency 0" "--fog 0" "--lighting 0 --transparency 0 --fog 0"; do
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-pipeline $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('atrium [$o]', d['ms_per_step'], d['config'].get('steps_per_ray'))"
done
