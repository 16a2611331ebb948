# A/B/n of several library builds on one box: tools/probe/libgps_<name>.so for every name given, bench alternating, ROUNDS times
# usage: bash tools/probe/abn.sh "base ed256 wgs128" [rounds]
NAMES=$1; ROUNDS=${2:-3}
mkdir -p gpurun_out
cp gps_slam_amd/libgpsslam_hip.so /tmp/libgps_shipped.so
for i in $(seq 1 $ROUNDS); do for v in $NAMES; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs > gpurun_out/abn_${v}_$i.log 2>&1
done; done
cp /tmp/libgps_shipped.so gps_slam_amd/libgpsslam_hip.so
NAMES="$NAMES" ROUNDS=$ROUNDS python - <<'PY'
import json, os
for v in os.environ['NAMES'].split():
    ov, sq = [], []
    for i in range(1, int(os.environ['ROUNDS']) + 1):
        l = [x for x in open('gpurun_out/abn_%s_%d.log' % (v, i)) if x.startswith('{')]
        if not l: print(v, i, 'NO LINE'); continue
        j = json.loads(l[-1]); c = j['config']
        ov.append(j['value']); sq.append(c['schedules']['sequential']['frames_per_s'])
        print(v, "overlap %.1f sequential %.1f" % (ov[-1], sq[-1]), [round(x, 3) for x in c.get('windows_ms_per_step', [])])
    if ov: print("==", v, "mean overlap %.1f sequential %.1f" % (sum(ov) / len(ov), sum(sq) / len(sq)))
PY
