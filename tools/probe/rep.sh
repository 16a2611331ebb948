# repeated default bench runs of the tree's build: spread and outliers of the two schedules
for i in 1 2 3 4 5 6; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-oracle-psnr 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; print('overlap %.1f sequential %.1f fusion_ms %.4f' % (j['value'], c['schedules']['sequential']['frames_per_s'], c['split']['fusion_ms_per_frame']))
"; done
