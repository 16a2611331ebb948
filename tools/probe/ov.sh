cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk && rocprofv3 --kernel-trace -d /tmp/pk -o a -- python $R/bench.py --steps 40 --warmup 10 --schedule overlap --no-cpu-baseline --no-oracle-psnr > /tmp/b.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
grep -o '"value": [0-9.]*' /tmp/b.log
python $R/tools/stream_overlap.py $DB 40
python $R/tools/keyframe_timeline.py $DB 500 900 -1
