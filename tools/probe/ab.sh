# A/B of two library builds on the same box: bench twice each, alternating
for i in 1 2; do for v in oldint newint; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-oracle-psnr > gpurun_out/ab_${v}_$i.log 2>&1
done; done
