"""Which stream carries the overlap schedule?  Frames/s of the bench scene with the map update's iteration count varied
(the frame stream's work stays the same): python tools/probe/iters.py <iters> [gt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
iters = int(sys.argv[1]); gt = len(sys.argv) > 2
W, H, NG, K, PRO = 640, 480, 200000, 100, 40
seq = bench.synthetic_sequence(W, H, PRO + K, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, gt, True, PRO + K, 1.0, 0.02)
scene.pipe.loadConfig(dict(local_opt_iters=iters))
scene.run(0, PRO)
torch.cuda.synchronize(); t0 = time.perf_counter()
scene.run(PRO, PRO + K)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("overlap schedule, %2d iterations per update%s: %.1f frames/s (%.3f ms per frame)" % (iters, " (given poses)" if gt else "", K / dt, 1e3 * dt / K))
