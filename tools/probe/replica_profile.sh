#!/bin/bash
# per-kernel table of the SLAM loop at Replica's own geometry (1200x680, fx = fy = 600, c = (599.5, 339.5): bench.py's default pinhole at
# that size IS Replica's camera), ~300 k Gaussians -> gpurun_out/<tag>_replica_kernel_stats.md
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --width 1200 --height 680 --gaussians 300000 --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0"
rm -rf /tmp/prof_rep && rocprofv3 --kernel-trace --stats -d /tmp/prof_rep -o bench -- $CMD > gpurun_out/replica_prof.log 2>&1
{
  echo "# $TAG -- Replica-native geometry (1200x680, f = 600), rocprofv3 --kernel-trace --stats summary (MI355X)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`; tools/prof_summary.py, one table per schedule over 3 windows of 20 frames."
  python tools/prof_summary.py "$(find /tmp/prof_rep -name '*.db' | head -1)" 30 --frames 20 --windows 3
  echo
  echo "bench line of the profiled run:"
  grep '^{"metric"' gpurun_out/replica_prof.log | tail -1 | cut -c1-700
} > gpurun_out/${TAG}_replica_kernel_stats.md
