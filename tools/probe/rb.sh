for a in "$@"; do GPS_ALT_LIB=$PWD/tools/probe/libgps_$a.so timeout 600 python tools/raster_bench.py 2>&1 | grep "bwd\|ward"; done
