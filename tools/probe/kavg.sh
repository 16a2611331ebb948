cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
cp $R/tools/probe/libgps_$v.so $R/gps_slam_amd/libgpsslam_hip.so
rm -rf /tmp/pk && rocprofv3 --kernel-trace -d /tmp/pk -o a -- python $R/bench.py --steps 60 --warmup 20 --schedule sequential --no-cpu-baseline --no-oracle-psnr > /dev/null 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
echo "== $v"
python $R/tools/frame_timeline.py $DB 5 | grep "$FILTER"
python $R/tools/kernel_avgs.py $DB "$KFILTER"
done
