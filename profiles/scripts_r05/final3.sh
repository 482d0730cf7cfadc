#!/bin/bash
# round 5: the two bench lines again, now that profiles/r05_pmc_*.json (which their `issue` and `roofline.traffic` objects quote) are the final library's
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
python bench.py > "$O/bench_atrium.json" 2> "$O/bench_atrium.err"; tail -c 300 "$O/bench_atrium.json"; echo
python bench.py --workload s256 --steps 10 --warmup 2 --cpu-seconds 6 > "$O/bench_s256.json" 2> "$O/bench_s256.err"; tail -c 200 "$O/bench_s256.json"; echo
