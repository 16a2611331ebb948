# A/B of tools/probe/libgps_base.so vs libgps_new.so in the OVERLAP schedule only (4 runs each, alternating, one box)
cp gps_slam_amd/libgpsslam_hip.so /tmp/libgps_shipped.so
for i in 1 2 3 4; do for v in base new; do
  cp tools/probe/libgps_$v.so gps_slam_amd/libgpsslam_hip.so
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 5 --schedule overlap --no-cpu-baseline --no-oracle-psnr --no-other-configs > gpurun_out/abo_${v}_$i.log 2>/dev/null
done; done
cp /tmp/libgps_shipped.so gps_slam_amd/libgpsslam_hip.so
python - <<'PY'
import json, statistics
for v in ("base","new"):
    vals=[]
    for i in range(1,5):
        l=[x for x in open('gpurun_out/abo_%s_%d.log'%(v,i)) if x.startswith('{')]
        j=json.loads(l[-1]); vals.append(j['value'])
    print(v, [round(x,1) for x in vals], "median %.1f mean %.1f"%(statistics.median(vals), statistics.mean(vals)))
PY
