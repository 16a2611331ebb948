cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 2 --schedule overlap --no-cpu-baseline --no-oracle-psnr --no-other-configs"
rm -rf /tmp/prof_ht && rocprofv3 --kernel-trace --hip-runtime-trace -d /tmp/prof_ht -o t -- $CMD > gpurun_out/ht_bench.log 2>&1
DB=$(find /tmp/prof_ht -name '*.db' | head -1)
python tools/launch_vs_start.py "$DB" ${1:-12} 2 > gpurun_out/launch_vs_start.txt 2>&1
cat gpurun_out/launch_vs_start.txt | cut -c1-230
cat gpurun_out/launch_vs_start.txt | cut -c1-220
