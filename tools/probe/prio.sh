run() { echo "$1: $(env $2 python bench.py --steps 100 --warmup 20 --schedule overlap --no-cpu-baseline --no-oracle-psnr 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('overlap %.1f' % (j['value']))
")"; }
run "warm-up" "A=1"
for p in 0 36864 49152 0 36864 49152; do run "batch LDS pad $p" "GPS_BATCH_PAD=$p"; done
