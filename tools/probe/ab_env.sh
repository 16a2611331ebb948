# A/B of one environment switch on one box, same library: runs bench.py alternately without / with the variable set
# usage: bash tools/probe/ab_env.sh GPS_BENCH_PINNED_LINE [pairs]
VAR=$1; N=${2:-3}
mkdir -p gpurun_out
for i in $(seq 1 $N); do for v in off on; do
  if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr > gpurun_out/abenv_${v}_$i.log 2>&1
done; done
unset $VAR
python - <<'PY'
import json, glob
for v in ('off', 'on'):
    for f in sorted(glob.glob('gpurun_out/abenv_%s_*.log' % v)):
        l = [x for x in open(f) if x.startswith('{')]
        if not l: print(v, f, 'NO LINE'); continue
        j = json.loads(l[-1]); c = j['config']; r = j['roofline']
        k = [x for x in r['kernels'] if x['kernel'].startswith('track_eval')]
        print(v, "overlap %.1f sequential %.1f" % (j['value'], c['schedules']['sequential']['frames_per_s']),
              "tracking %.3f ms/frame, poll %.2f us" % (r['fusion']['tracking_ms_per_frame'], k[0]['avg_us'] if k else -1),
              {a: b for a, b in (k[0] if k else {}).items() if 'spin' in a or 'wait' in a or 'eval' in a})
PY
