#!/bin/bash
# round 5, GPU batch 11: (a) the context-lifecycle stress on the round-5 library; (b) the AIC_HURRY_STEPS sweep VERDICT r04 next 2 asks for,
# on the ROUND-4 kernel it was written for (variants built from commit 8025db6 + profiles/scripts_r05/hurry_rules.patch: the event rule and the trip rule apart):
# n = 0.5-0.8 x the frame's longest ray (C2: 205 steps, C3: the 1000-step cap), >= 3 s per point
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b11; mkdir -p $O
timeout 900 python -X faulthandler tools/lifecycle_stress.py 20000 $O/lifecycle.log > $O/lifecycle.out 2> $O/lifecycle.err; echo "lifecycle rc=$?"; tail -3 $O/lifecycle.out; tail -5 $O/lifecycle.err; tail -1 $O/lifecycle.log
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
row() { python - "$1" "$2" <<PY
import json, sys
d = json.loads(open(sys.argv[1]).readlines()[-1]); s = d.get("single_frame", {})
print(sys.argv[2], "streamed", d["ms_per_step"], "one frame warm", s.get("single_frame_warm_ms"), "cold", s.get("single_frame_cold_ms"), "moving", s.get("single_frame_moving_camera_ms"))
PY
}
for n in r04base h100both h100ev h100tr h125both h125ev h125tr h145both h145ev h145tr h165both h165ev h165tr; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 5 --min-seconds 3 > $O/${n}_atrium.json 2> $O/${n}_atrium.err; row $O/${n}_atrium.json "$n atrium" || tail -2 $O/${n}_atrium.err
done | tee $O/hurry_c2.txt
for n in r04base h500both h500ev h500tr h600both h600ev h600tr h700both h700ev h700tr h800both h800ev h800tr; do
  cp variants/libaic_hip_$n.so all_is_cubes_amd/libaic_hip.so
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --workload s256 --steps 8 --warmup 2 --min-seconds 3 > $O/${n}_s256.json 2> $O/${n}_s256.err; row $O/${n}_s256.json "$n s256" || tail -2 $O/${n}_s256.err
done | tee $O/hurry_c3.txt
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
