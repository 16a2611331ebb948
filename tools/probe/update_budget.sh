# One kernel trace of the bench (both schedules, two windows each) -> tools/update_budget.py for each schedule:
# where a keyframe update's time goes on the map stream (raycasts / mask render / addGaussians, the 20 iterations, prune).
# run ON the GPU box:  bash tools/probe/update_budget.sh   -> gpurun_out/update_budget_{sequential,overlap}.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 20 --warmup 5 --windows 2 --no-cpu-baseline --no-oracle-psnr --no-other-configs"
rm -rf /tmp/prof_ub && rocprofv3 --kernel-trace -d /tmp/prof_ub -o t -- $CMD > gpurun_out/update_budget_bench.log 2>&1
DB=$(find /tmp/prof_ub -name '*.db' | head -1)
python tools/update_budget.py "$DB" 0 2 > gpurun_out/update_budget_sequential.txt 2>&1
python tools/update_budget.py "$DB" 1 2 > gpurun_out/update_budget_overlap.txt 2>&1
tail -5 gpurun_out/update_budget_sequential.txt; tail -5 gpurun_out/update_budget_overlap.txt
