#!/bin/bash
# round-2 call 1: instruction issue-rate ubench + baseline numbers of the round-1 kernel on this box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02c1; mkdir -p $O
( cd tools/ubench && timeout 300 ./issue_rate ) > $O/issue_rate.txt 2>&1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
timeout 300 $B > $O/bench_atrium_pipe.json 2>$O/err1.txt
timeout 300 $B --no-pipeline > $O/bench_atrium_nopipe.json 2>$O/err2.txt
timeout 300 $B --workload s256 --steps 8 --warmup 2 --no-pipeline > $O/bench_s256_nopipe.json 2>$O/err3.txt
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_atrium_nopipe -- $B --steps 20 --no-pipeline > $O/stats.log 2>&1
find $O -type f -size +2M -delete
tail -3 $O/issue_rate.txt; tail -c 600 $O/bench_atrium_nopipe.json
