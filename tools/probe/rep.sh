# repeated default bench runs of the tree's build: spread and outliers of the overlap schedule (frames > 2 ms listed)
N=${1:-6}
for i in $(seq 1 $N); do GPS_BENCH_FRAME_TIMES=1 python bench.py --steps 100 --warmup 20 --schedule overlap --no-cpu-baseline --no-oracle-psnr 2> gpurun_out/rep_err_$i.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; print('overlap %.1f' % (j['value']))
"; grep "^frame\|^flush" gpurun_out/rep_err_$i.log | tail -101 | awk '$3 > 2.0 {printf "   %s %s ms;", $2, $3} END {print ""}'; done
