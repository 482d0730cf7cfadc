python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "3 1 1" "1 1 1"; do set -- $cfg
echo "== lighting=$1 fog=$2 transparency=$3"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --lighting $1 --fog $2 --transparency $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['gsteps_per_s'], d['config']['steps_per_ray'])"
done
