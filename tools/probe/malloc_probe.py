"""Which allocations still reach hipMalloc in the steady state of the overlap schedule?  Runs the bench scene and prints, per
keyframe period, the caching allocator's segment statistics (num_device_alloc, reserved bytes per pool) -- a segment allocated
in a timed window costs the allocating thread a hipMalloc and can stall the other streams.
usage (GPU box): python tools/probe/malloc_probe.py [periods]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = "cuda:0"
n = 10 * periods + 1
seq = bench.synthetic_sequence(640, 480, n, 1234)
seeds = bench.seed_gaussians(seq, 200000, 1234, dev)
bench.prime(dev)
scene = bench.Scene(seq, seeds, 1234, False, overlap=True, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
torch.cuda.synchronize()
prev = None
for p in range(periods):
    scene.run(10 * p, 10 * p + 10)
    torch.cuda.synchronize()
    s = torch.cuda.memory_stats()
    cur = (s["num_device_alloc"], s["reserved_bytes.large_pool.current"], s["reserved_bytes.small_pool.current"],
           s["allocated_bytes.all.current"], s["segment.large_pool.current"], s["segment.small_pool.current"])
    if prev:
        print("period %2d: N=%d  +%d hipMalloc  large pool %+.1f MB (%d segments)  small pool %+.1f MB (%d segments)  live %.1f MB" % (
            p, scene.model.getGaussianNum(), cur[0] - prev[0], (cur[1] - prev[1]) / 1e6, cur[4], (cur[2] - prev[2]) / 1e6, cur[5], cur[3] / 1e6))
    prev = cur
snap = torch.cuda.memory_snapshot()
big = sorted([(seg["total_size"], seg["segment_type"], seg["stream"], len(seg["blocks"])) for seg in snap], reverse=True)
from collections import Counter
print("segments by (type, size MB, stream):", sorted(Counter((t, round(sz / 1e6, 1), st) for sz, t, st, _ in big).items(), key=lambda kv: -kv[0][1])[:40])
