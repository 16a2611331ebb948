# A/B of two library builds on the same box (tools/probe/libgps_old.so, libgps_new.so): bench twice each, alternating
for i in 1 2; do for v in old new; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-oracle-psnr > gpurun_out/ab_${v}_$i.log 2>&1
done; done
python - <<'PY'
import json
for v in ('old','new'):
  for i in (1,2):
    l=[x for x in open('gpurun_out/ab_%s_%d.log'%(v,i)) if x.startswith('{')][-1]; j=json.loads(l)
    c=j['config']; print(v, "overlap %.1f sequential %.1f fusion_ms %.4f residual_ms %.4f bwd_us %.1f" % (j['value'], c['schedules']['sequential']['frames_per_s'], c['split']['fusion_ms_per_frame'], c['split']['gaussian_ms_per_frame'], j['roofline'].get('avg_launch_us')))
PY
