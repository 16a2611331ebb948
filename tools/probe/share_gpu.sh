# Rehearsal of the driver's N > 1 launch on a 1-GPU box: N ranks share the GPU (gloo group), each runs its own scene.
# NOT a scaling measurement -- it checks that N processes of the bench coexist (threads, pinned buffers, BAR lines, barriers)
# and that rank 0 prints the aggregate line.  usage: bash tools/probe/share_gpu.sh [N]
N=${1:-2}
mkdir -p gpurun_out
GPS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 20 --warmup 5 --windows 3 > gpurun_out/share_gpu_$N.log 2>&1
echo "rc $?"
grep '^{"metric"' gpurun_out/share_gpu_$N.log | tail -1 | cut -c1-500
grep -i "error\|traceback\|tracker:" gpurun_out/share_gpu_$N.log | head -10
