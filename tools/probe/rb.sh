# tools/raster_bench.py with each named probe library as the second build (bit-for-bit forward comparison, backward difference, times)
for a in "$@"; do GPS_ALT_LIB=$PWD/tools/probe/libgps_$a.so timeout 600 python tools/raster_bench.py 2>&1 | grep "raster\|ward\|step"; done
