#!/bin/bash
# only part 1 of tools/profile.sh: rocprofv3 --kernel-trace summary of the driver's bench command per schedule -> gpurun_out/<tag>_bench_kernel_stats.md,
# with the SAME run's in-loop launch timings (gps_launch_timing_*, what the line's roofline is priced with) printed beside the tables
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 --full-line"
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- $CMD > gpurun_out/bench_prof.log 2>&1
{
  echo "# $TAG -- rocprofv3 --kernel-trace --stats summary (MI355X, gfx950)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`; tools/prof_summary.py, one table per schedule over 5 windows of 20 frames."
  python tools/prof_summary.py "$(find /tmp/prof_stats -name '*.db' | head -1)" 40 --frames 20 --windows 5 --in-loop
  echo
  echo "### the same run's in-loop launch timings (device-clock stamps of each launch's first wave start / last wave end, one extra 20-frame window per schedule)"
  echo
  grep '^{"metric"' gpurun_out/bench_prof.log | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
for sched, r in d['config']['schedules'].items():
    il = r.get('in_loop')
    if not il:
        continue
    print('schedule %s (window %.3f ms/frame):' % (sched, il['window_ms_per_step']))
    print('| kernel | launches | avg_us | max_us | us/frame | flagged launches | avg_us flagged | avg_us unflagged |')
    print('|---|---|---|---|---|---|---|---|')
    for k, v in sorted(il['kernels'].items(), key=lambda kv: -kv[1]['us_per_frame']):
        f = lambda x: '-' if x is None else '%.2f' % x
        print('| %s | %d | %.2f | %.2f | %.1f | %d | %s | %s |' % (k, v['launches'], v['avg_us'], v['max_us'], v['us_per_frame'], v['launches_flagged'], f(v['avg_us_flagged']), f(v['avg_us_unflagged'])))
    print()
r = d['roofline']
print('roofline of the line: kernel %s, avg_launch_us %.2f (alone %.2f), bytes %.0f, frac %.4f (alone %.4f), timed in: %s' % (r['kernel'], r['avg_launch_us'], r['avg_launch_us_alone'], r['algorithmic_bytes'], r['frac'], r['frac_alone'], r['timed_in']))
print('value %.1f frames/s, sequential %.1f' % (d['value'], d['config'].get('sequential_fps', 0)))
"
} > gpurun_out/${TAG}_bench_kernel_stats.md
