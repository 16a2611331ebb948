# A/B/n of one environment variable's VALUES on one box, same library: bash tools/probe/ab_envval.sh VAR "v1 v2 ..." [rounds]   ("-" = unset)
VAR=$1; VALS=$2; N=${3:-3}
mkdir -p gpurun_out
for i in $(seq 1 $N); do for v in $VALS; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs > gpurun_out/abval_${v}_$i.log 2>&1
done; done
unset $VAR
VALS="$VALS" N=$N python - <<'PY'
import json, os
for v in os.environ['VALS'].split():
    ov, sq = [], []
    for i in range(1, int(os.environ['N']) + 1):
        l = [x for x in open('gpurun_out/abval_%s_%d.log' % (v, i)) if x.startswith('{')]
        if not l: print(v, i, 'NO LINE'); continue
        j = json.loads(l[-1]); ov.append(j['value']); sq.append(j['config']['schedules']['sequential']['frames_per_s'])
    print("value %-3s overlap %s mean %.1f | sequential mean %.1f" % (v, [round(x) for x in ov], sum(ov) / len(ov), sum(sq) / len(sq)))
PY
