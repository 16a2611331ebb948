cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 2 --schedule overlap --no-cpu-baseline --no-oracle-psnr --no-other-configs"
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace -d /tmp/prof_kt -o t -- $CMD > gpurun_out/kt_bench.log 2>&1
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python tools/keyframe_timeline.py "$DB" 700 1700 -3 > gpurun_out/kt_timeline.txt 2>&1
wc -l gpurun_out/kt_timeline.txt
