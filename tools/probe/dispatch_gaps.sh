# kernel trace of the overlap schedule -> tools/dispatch_gaps.py (run ON the GPU box) -> gpurun_out/dispatch_gaps.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 3 --schedule overlap --no-cpu-baseline --no-oracle-psnr --no-other-configs"
rm -rf /tmp/prof_dg && rocprofv3 --kernel-trace -d /tmp/prof_dg -o t -- $CMD > gpurun_out/dg_bench.log 2>&1
python tools/dispatch_gaps.py "$(find /tmp/prof_dg -name '*.db' | head -1)" ${1:-12} > gpurun_out/dispatch_gaps.txt 2>&1
cat gpurun_out/dispatch_gaps.txt
