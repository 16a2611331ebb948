mkdir -p gpurun_out
for i in 1 2 3; do for v in old new; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr > gpurun_out/ab_${v}_$i.log 2>&1
done; done
python - <<'PY'
import json
for v in ('old','new'):
  for i in (1,2,3):
    l=[x for x in open('gpurun_out/ab_%s_%d.log'%(v,i)) if x.startswith('{')][-1]; j=json.loads(l)
    c=j['config']; print(v, "overlap %.1f sequential %.1f" % (j['value'], c['schedules']['sequential']['frames_per_s']), [round(x,3) for x in c.get('windows_ms_per_step',[])])
PY
