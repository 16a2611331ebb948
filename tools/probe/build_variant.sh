#!/bin/bash
# A variant of the kernel library for an A/B: tools/probe/build_variant.sh <name> <file.hip> "<extra hipcc flags>"
#   -> tools/probe/libs/libgps_<name>.so = the cached objects of gps_slam_amd/build/ with <file.hip> recompiled with the flags
# Use it through GPS_SLAM_HIP_LIB=<that path> (Python hosts) or by copying it over gps_slam_amd/libgpsslam_hip.so (C++ host).
set -e
NAME=$1; FILE=$2; EXTRA=$3
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
python -c "from gps_slam_amd import _build; _build.build()" >/dev/null
mkdir -p "$ROOT/tools/probe/libs"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DGPS_BUILDING_DLL"
case "$FILE" in tsdf_*) FLAGS="$FLAGS -ffp-contract=off";; esac
OBJ="$ROOT/tools/probe/libs/${NAME}_${FILE%.hip}.o"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c "$ROOT/gps_slam_amd/csrc/$FILE" -o "$OBJ"
OBJS=$(ls "$ROOT"/gps_slam_amd/build/*.o | grep -v "/host_" | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probe/libs/libgps_$NAME.so" $OBJS "$OBJ"
echo "$ROOT/tools/probe/libs/libgps_$NAME.so"
