#!/bin/bash
# Round 3, GPU batch 4: the whole GPU suite (widened full-size parity, info text, light visits), smoke, the replay workload
# against the workload it was recorded from, the light-bench / relight lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b4; rm -rf $O; mkdir -p $O
echo "== tests"
AIC_FUZZ_N=100 AIC_LIGHT_FUZZ_N=64 timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -16 $O/pytest.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline"
echo "== replay vs atrium"
python tools/make_recording.py atrium /tmp/atrium_like.aic
timeout 300 $B --steps 30 --warmup 3 > $O/atrium.json 2> $O/atrium.err; tail -c 2600 $O/atrium.json; echo
timeout 300 $B --steps 30 --warmup 3 --workload replay:/tmp/atrium_like.aic > $O/replay.json 2> $O/replay.err; tail -3 $O/replay.err; python -c "
import json
a=json.loads(open('$O/atrium.json').readlines()[-1]); r=json.loads(open('$O/replay.json').readlines()[-1])
print('atrium', a['ms_per_step'], a['config']['steps_per_ray'], a['roofline']['algorithmic_bytes_per_launch'], a['roofline'].get('frac_one_at_a_time'))
print('replay', r['ms_per_step'], r['config']['steps_per_ray'], r['roofline']['algorithmic_bytes_per_launch'], r['config']['workload'])"
echo "== light-bench"
timeout 300 $B --workload light-bench --steps 100 --warmup 10 > $O/lightbench.json 2> $O/lightbench.err; python -c "
import json
d=json.loads(open('$O/lightbench.json').readlines()[-1]); print(json.dumps(d['light_update']))"
echo "== relight"
for cfg in "4 2048" "4 4096" "8 2048" "1 1024"; do set -- $cfg
timeout 300 $B --workload relight --steps 120 --warmup 10 --relight-period $1 --light-budget $2 > $O/relight_$1_$2.json 2> $O/relight_$1_$2.err; python -c "
import json
d=json.loads(open('$O/relight_$1_$2.json').readlines()[-1]); print('period $1 budget $2', d['ms_per_step'], json.dumps(d['relight']))" || tail -3 $O/relight_$1_$2.err
done
