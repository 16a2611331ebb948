"""How many of the model's Gaussians does an optimise view see (radius > 0) as the whole-sequence run grows the model?  (A zero-
gradient Gaussian's Adam step does not depend on the iteration's rasterization: how much of the dense step could run beside it.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
sc = bench.Scene(seq, None, 1234, False, overlap=False, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
for lo in range(0, n, 100):
    sc.run(lo, min(n, lo + 100))
    torch.cuda.synchronize()
    views = list(zip(sc.pipe.optCams(), sc.pipe.optRaycasts()))
    N = sc.model.getGaussianNum()
    fr = []
    with torch.no_grad():
        for cam, rc in views:
            res = sc.model.forward(cam, rc["depth_map"], rc["color_map"])
            fr.append(float((res["radiis"] > 0).sum().item()) / N)
    print("after frame %4d: N = %6d, %d views, visible fraction min %.3f mean %.3f max %.3f" % (lo + 99, N, len(views), min(fr), np.mean(fr), max(fr)), flush=True)
sc.close()
