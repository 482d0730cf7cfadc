cd ${GRAFT_REPO_ROOT:-/root/repo}
cp all_is_cubes_amd/libaic_hip.so /tmp/libaic_default.so
for v in lighttiming nostore; do
cp variants/libaic_hip_$v.so all_is_cubes_amd/libaic_hip.so
echo "== $v"
python bench.py --workload light-bench --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-secondary --min-seconds 0 2>&1 >/dev/null | grep "light host us\|light timing" | head -2
done
cp /tmp/libaic_default.so all_is_cubes_amd/libaic_hip.so
