#!/bin/bash
# A/B of kernel-library variants on one box through the C++ host (which links gps_slam_amd/libgpsslam_hip.so by rpath): the variant
# is copied over the shipped library for its runs.  tools/probe/ab_lib.sh <runs> default <name> [<name> ...]   (tools/probe/libs/libgps_<name>.so)
RUNS=$1; shift
mkdir -p gpurun_out
cp gps_slam_amd/libgpsslam_hip.so /tmp/libgps_default.so
for i in $(seq 1 $RUNS); do
  for v in "$@"; do
    if [ "$v" = default ]; then cp /tmp/libgps_default.so gps_slam_amd/libgpsslam_hip.so; else cp tools/probe/libs/libgps_$v.so gps_slam_amd/libgpsslam_hip.so; fi
    python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames ${WHOLE:-0} 2> gpurun_out/ablib_err.log | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; j=json.loads(l); c=j['config']
print('lib $v run $i: sequential %.1f overlap %.1f%s' % (c['sequential_fps'], c['overlap_fps'], (' whole-run %.1f / %.1f' % (c['whole_run_fps_sequential'], c['whole_run_fps'])) if 'whole_run_fps' in c else ''))"
  done
done
cp /tmp/libgps_default.so gps_slam_amd/libgpsslam_hip.so
