#!/bin/bash
# sb_scatter_kernel average at Replica's geometry for library variants: tools/probe/scatter_time.sh default rows8 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp gps_slam_amd/libgpsslam_hip.so /tmp/libgps_default.so
for v in "$@"; do
  if [ "$v" = default ]; then cp /tmp/libgps_default.so gps_slam_amd/libgpsslam_hip.so; else cp tools/probe/libs/libgps_$v.so gps_slam_amd/libgpsslam_hip.so; fi
  rm -rf /tmp/prof_sc && rocprofv3 --kernel-trace -d /tmp/prof_sc -o s -- python bench.py --width ${W:-1200} --height ${H:-680} --gaussians ${NG:-300000} --steps 20 --warmup 5 --windows 2 --schedule sequential --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 > /tmp/sc.log 2>&1
  python - <<PY
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/prof_sc/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if "spin_kernel" in n]
d = [e - s for n, s, e in rows if "sb_scatter" in n and marks[0] < s < marks[3]]
print("$v: sb_scatter_kernel %d launches in the timed windows, mean %.2f us, median %.2f" % (len(d), sum(d) / len(d) / 1e3, sorted(d)[len(d) // 2] / 1e3))
PY
done
cp /tmp/libgps_default.so gps_slam_amd/libgpsslam_hip.so
