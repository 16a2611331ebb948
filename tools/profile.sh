#!/bin/bash
# Regenerates the artefacts kept under profiles/ (run ON the GPU box:  gpurun -- 'bash tools/profile.sh rNN').
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command, summarised PER SCHEDULE over its timed windows
#      (tools/prof_summary.py) -> gpurun_out/<tag>_bench_kernel_stats.md
#   2. HBM traffic + instruction mix of the kernels that carry a frame, on THE SAME scene: FETCH_SIZE, WRITE_SIZE and two SQ
#      counter groups in SEPARATE --pmc passes (one counter group each, MI355X guide "rocprofv3 PMC slots"; never combined with
#      trace domains other than --kernel-trace) over the same bench command; the counters of bench_kernels.roofline_section's
#      micro-benchmark loops (each behind its own marker kernel) -> gpurun_out/pmc_<kernel>.json (tools/pmc_extract.py)
#   3. the bench line itself (which then picks the PMC files up) -> gpurun_out/<tag>_bench_line.json (compact, as printed) and
#      gpurun_out/<tag>_bench_full.json (the full record)
# Copy the files into profiles/ and commit them.
TAG=${1:-rXX}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
WINDOWS=${WINDOWS:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps $STEPS --warmup $WARMUP --windows $WINDOWS --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 --full-line"
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- $CMD > gpurun_out/bench_prof.log 2>&1
{
  echo "# $TAG -- rocprofv3 --kernel-trace --stats summary (MI355X, gfx950)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`.  Summarised from the rocpd database with tools/prof_summary.py:"
  echo "one table per schedule over its $WINDOWS timed windows of $STEPS frames (phase markers: bench.py's spin_kernel launches);"
  echo "\`sequential\` = the reference's schedule, \`overlap\` = the one \`value\` reports; us/frame = total / ($WINDOWS x $STEPS) frames."
  python tools/prof_summary.py "$(find /tmp/prof_stats -name '*.db' | head -1)" 40 --frames $STEPS --windows $WINDOWS --in-loop
  echo
  echo "bench line of the profiled run:"
  grep '^{"metric"' gpurun_out/bench_prof.log | tail -1 | cut -c1-600
} > gpurun_out/${TAG}_bench_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c && rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o a -- $CMD > gpurun_out/pmc_$c.log 2>&1
done
rm -rf /tmp/prof_SQ && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/prof_SQ -o a -- $CMD > gpurun_out/pmc_SQ.log 2>&1
rm -rf /tmp/prof_SQ2 && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/prof_SQ2 -o a -- $CMD > gpurun_out/pmc_SQ2.log 2>&1
python tools/pmc_extract.py gpurun_out/pmc_FETCH_SIZE.log /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE /tmp/prof_SQ /tmp/prof_SQ2 gpurun_out/pmc > gpurun_out/pmc_extract.log 2>&1
cat gpurun_out/pmc_extract.log
# the bench line with the fresh PMC files in place (bench_kernels reads profiles/pmc_*.json)
mkdir -p profiles && cp gpurun_out/pmc/pmc_*.json profiles/ 2>/dev/null
# (the driver's command; stdout = ONE compact line, the full record goes to gpurun_out/bench_full.json)
python bench.py --steps $STEPS --warmup $WARMUP --windows $WINDOWS > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log
tail -1 gpurun_out/bench_stdout.log > gpurun_out/${TAG}_bench_line.json
cp gpurun_out/bench_full.json gpurun_out/${TAG}_bench_full.json
wc -c gpurun_out/${TAG}_bench_line.json gpurun_out/${TAG}_bench_full.json
cat gpurun_out/${TAG}_bench_line.json
