for v in base nopn noat; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  bash tools/probe/kavg.sh bench.py --steps 20 --warmup 5 --windows 1 --schedule sequential --no-cpu-baseline --no-oracle-psnr 2>&1 | grep -E "track_prepare|track_eval_poll|icp_kernel" | sed "s/^/$v: /"
done
