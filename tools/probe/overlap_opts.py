"""Overlap schedule of the bench scene with SLAMPipeline options overridden: python tools/probe/overlap_opts.py [attr=value ...]
(e.g. merge_keyframe_raycasts=1 async_raycasts=0); prints frames/s of 100 timed frames after 40 settle frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
W, H, NG, K, PRO = 640, 480, 200000, 100, 40
seq = bench.synthetic_sequence(W, H, PRO + K, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, False, True, PRO + K, 1.0, 0.02)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    setattr(scene.pipe, k, type(getattr(scene.pipe, k))(int(v)))
scene.run(0, PRO)
torch.cuda.synchronize(); t0 = time.perf_counter()
scene.run(PRO, PRO + K)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%s: %.1f frames/s" % (" ".join(sys.argv[1:]) or "defaults", K / dt))
