#!/bin/bash
# only part 1 of tools/profile.sh: rocprofv3 --kernel-trace summary of the driver's bench command per schedule -> gpurun_out/<tag>_bench_kernel_stats.md
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0"
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- $CMD > gpurun_out/bench_prof.log 2>&1
{
  echo "# $TAG -- rocprofv3 --kernel-trace --stats summary (MI355X, gfx950)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`; tools/prof_summary.py, one table per schedule over 5 windows of 20 frames."
  python tools/prof_summary.py "$(find /tmp/prof_stats -name '*.db' | head -1)" 40 --frames 20 --windows 5
  echo
  grep '^{"metric"' gpurun_out/bench_prof.log | tail -1 | cut -c1-400
} > gpurun_out/${TAG}_bench_kernel_stats.md
