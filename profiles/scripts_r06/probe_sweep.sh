for cap in 0 32 64 96 160 999; do
  for wl in atrium s256; do
    st=20; [ $wl = s256 ] && st=6
    AIC_PROBE_CAP=$cap timeout 300 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline --no-secondary --min-seconds 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['single_frame']; print('cap $cap $wl streamed', d['ms_per_step'], 'warm', s['single_frame_warm_ms'], 'cold', s['single_frame_cold_ms'], 'moving', s['single_frame_moving_camera_ms'], 'kernel warm/cold', s['kernel_ms_warm'], s['kernel_ms_cold'])"
  done
done
