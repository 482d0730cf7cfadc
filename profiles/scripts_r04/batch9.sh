#!/bin/bash
# Round 4, batch 9: where the light updater's host time goes (-DAIC_LIGHT_TIMING build: host lap timers per call).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b9; mkdir -p $O
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
cp variants/libaic_hip_lighttiming.so all_is_cubes_amd/libaic_hip.so
python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 > $O/lb.json 2> $O/lb.err
grep "light host us\|light timing" $O/lb.err
python bench.py --workload relight --steps 120 --warmup 10 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 > $O/rl.json 2> $O/rl.err
grep "light host us" $O/rl.err | awk '{n++; u+=$5; b+=$7; for(i=9;i<=19;i+=2) s[i]+=$i; d+=$21; if ($5>=2000) {N++; for(i=9;i<=19;i+=2) S[i]+=$i; D+=$21}} END {printf "calls %d updates %d batches %d setup %.0f submit %.0f wait %.0f copy2 %.0f apply %.0f scatter %.0f device_ms %.1f (us totals)\n", n,u,b,s[9],s[11],s[13],s[15],s[17],s[19],d; if (N) printf "calls with >=2000 updates: %d; per call us: setup %.0f submit %.0f wait %.0f copy2 %.0f apply %.0f scatter %.0f device %.0f\n", N,S[9]/N,S[11]/N,S[13]/N,S[15]/N,S[17]/N,S[19]/N,1000*D/N}'
grep "light host us" $O/rl.err | sort -k5,5n | awk 'NR%40==1' | head -30
tail -c 600 $O/rl.json
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
