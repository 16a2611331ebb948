#!/bin/bash
# builds tools/probe/libgps_stats.so / libgps_sections.so: the library with the raycaster's instrumentation compiled in
# (GPS_RAYCAST_STATS: per-ray step counts + start/end ticks in the output image; _SECTIONS: per-section tick sums)
set -e
cd "$(dirname "$0")/../.."
OBJS=$(for f in gps_slam_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b = tsdf_render ] || echo gps_slam_amd/build/$b.o; done)
for v in stats sections; do
  FL="-DGPS_RAYCAST_STATS"; [ $v = sections ] && FL="$FL -DGPS_RAYCAST_STATS_SECTIONS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DGPS_BUILDING_DLL -ffp-contract=off $FL \
      -Iinclude -c gps_slam_amd/csrc/tsdf_render.hip -o /tmp/tsdf_render_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libgps_$v.so $OBJS /tmp/tsdf_render_$v.o
done
ls -la tools/probe/*.so
