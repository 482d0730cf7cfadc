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
python tools/fetch_calib_table.py "$O" | tee "$O/fetch_calib.txt"
find "$O"/calib_fetch "$O"/calib_write -type f -size +1M -delete
