# The 5x outlier launches on the map stream (round-3 review item 3): which calls are they and what runs beside them?
#   1. kernel trace of the bench (both schedules, one window each) -> tools/outlier_overlap.py
#   2. timeline of one keyframe update of the sequential schedule -> tools/keyframe_timeline.py
#   3. A/B: the keyframe views' batched raycast beside the update's first kernels (async_raycasts = 1, the default) or before them (0)
# run ON the GPU box:  bash tools/probe/outliers.sh   -> gpurun_out/outliers_*.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 2 --no-cpu-baseline --no-oracle-psnr --no-other-configs"
rm -rf /tmp/prof_outl && rocprofv3 --kernel-trace -d /tmp/prof_outl -o t -- $CMD > gpurun_out/outliers_bench.log 2>&1
DB=$(find /tmp/prof_outl -name '*.db' | head -1)
python tools/outlier_overlap.py "$DB" 2.0 > gpurun_out/outliers_overlap.txt 2>&1
python tools/keyframe_timeline.py "$DB" 50 900 -2 > gpurun_out/outliers_timeline.txt 2>&1
for v in 1 0 1 0 1 0; do
  GPS_BENCH_ASYNC_RAYCASTS=$v $CMD 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); s=d['config']['schedules']
        print('async_raycasts=$v overlap %.1f sequential %.1f frames/s' % (s['overlap']['frames_per_s'], s['sequential']['frames_per_s']))
"
done > gpurun_out/outliers_ab.txt 2>&1
head -60 gpurun_out/outliers_overlap.txt; cat gpurun_out/outliers_ab.txt
