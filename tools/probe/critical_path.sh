# Which chain carries the overlap schedule's period?  The same bench with the map update's iteration count changed
# (GPS_BENCH_OPT_ITERS, a probe aid: NOT the metric's workload) and with tracking off (--gt-pose): if the period follows
# the iterations, the update is critical; if it follows the tracker, the frame chain is.
# run ON the GPU box:  bash tools/probe/critical_path.sh  -> gpurun_out/critical_path.txt
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); s=d['config']['schedules']
        print('$1: overlap %.1f sequential %.1f frames/s (period %.2f / %.2f ms)' % (s['overlap']['frames_per_s'], s['sequential']['frames_per_s'], 1e4/s['overlap']['frames_per_s'], 1e4/s['sequential']['frames_per_s']))
"; }
{
for it in 20 10 30 0; do GPS_BENCH_OPT_ITERS=$it $CMD 2>/dev/null | show "iters=$it"; done
$CMD --gt-pose 2>/dev/null | show "iters=20 gt-pose"
GPS_BENCH_OPT_ITERS=10 $CMD --gt-pose 2>/dev/null | show "iters=10 gt-pose"
} > gpurun_out/critical_path.txt 2>&1
cat gpurun_out/critical_path.txt
