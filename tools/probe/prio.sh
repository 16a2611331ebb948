run() { echo "$1: $(env $2 python bench.py --steps 100 --warmup 20 --schedule overlap --no-cpu-baseline --no-oracle-psnr 2>&1 | grep -o '"value": [0-9.]*')"; }
run "frame hi, map normal (shipped)" "A=1"
run "frame hi, map hi" "GPS_MAP_HI=1"
run "frame normal, map hi" "GPS_MAP_HI=1 GPS_FRAME_LO=1"
run "frame normal, map normal" "GPS_FRAME_LO=1"
run "frame hi, map normal (shipped)" "A=1"
