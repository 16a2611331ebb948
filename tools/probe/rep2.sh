# repeated default bench runs (5 windows of 20 frames, overlap schedule only): every processFrame call that took > 2.5 ms of host
# time is listed with its frame index, next to the windows' ms per step -- finds the rare frame that costs one window ~4.7 ms
N=${1:-6}
mkdir -p gpurun_out
for i in $(seq 1 $N); do GPS_BENCH_PIPE_TIMES=4 GPS_BENCH_FRAME_TIMES=1 python bench.py --steps 20 --warmup 5 --windows 5 --schedule overlap --no-cpu-baseline --no-oracle-psnr 2> gpurun_out/rep_err_$i.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; print('overlap %.1f' % (j['value']), [round(x, 3) for x in c['windows_ms_per_step']], 'mallocs', c['schedules']['overlap'].get('windows_device_mallocs'))
"; grep "^frame\|^flush" gpurun_out/rep_err_$i.log | awk '$1 == "frame" && $3 > 2.5 {printf "   %s %s ms;", $2, $3} END {print ""}'; grep "^\[pipe\]\|gps_slam_hip" gpurun_out/rep_err_$i.log | grep -v "frame 1[01]:\|update 1:" | tail -12; done
