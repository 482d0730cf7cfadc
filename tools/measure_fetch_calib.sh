#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known access patterns (tools/ubench/fetch_calib.hip).
#   gpurun -- 'bash tools/measure_fetch_calib.sh r04'   ->  gpurun_out/<tag>/fetch_calib.txt (copied to profiles/ by hand)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out/$TAG; mkdir -p "$O"
[ -x tools/ubench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
rm -rf "$O"/calib_fetch "$O"/calib_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- tools/ubench/fetch_calib > "$O/calib_patterns.csv" 2> "$O/calib_fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- tools/ubench/fetch_calib > /dev/null 2> "$O/calib_write.err"
# the raw request counters FETCH_SIZE is derived from (the derived metric turned out to report nothing for 2-byte loads): which of
# them see a global_load_ushort that misses L2?
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d "$O/calib_raw" -- tools/ubench/fetch_calib > /dev/null 2> "$O/calib_raw.err"
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d "$O/calib_raw2" -- tools/ubench/fetch_calib > /dev/null 2> "$O/calib_raw2.err"
python tools/fetch_calib_table.py "$O" | tee "$O/fetch_calib.txt"
find "$O"/calib_fetch "$O"/calib_write "$O"/calib_raw "$O"/calib_raw2 -type f -size +1M -delete
