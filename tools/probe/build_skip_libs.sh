#!/bin/bash
# library variants with different free-space look-ahead of the raycaster: tools/probe/libgps_skip<N>.so
set -e
cd "$(dirname "$0")/../.."
OBJS=$(for f in gps_slam_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b = tsdf_render ] || echo gps_slam_amd/build/$b.o; done)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DGPS_BUILDING_DLL -ffp-contract=off -DGPS_RAYCAST_SKIP=$n \
      -Iinclude -c gps_slam_amd/csrc/tsdf_render.hip -o /tmp/tsdf_render_skip$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libgps_skip$n.so $OBJS /tmp/tsdf_render_skip$n.o
done
ls tools/probe/*.so
