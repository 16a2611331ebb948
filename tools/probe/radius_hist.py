"""Radius histogram of the Gaussians an optimise view sees, late in a whole-sequence run (what the strip backward's tasks are made of).
python tools/probe/radius_hist.py [frames] [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
seq = bench.synthetic_sequence_device(W, H, n, 1234, dev)
sc = bench.Scene(seq, None, 1234, False, overlap=False, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
sc.run(0, n)
torch.cuda.synchronize()
N = sc.model.getGaussianNum()
edges = [0, 4, 8, 16, 32, 48, 64, 80, 99, 100]
with torch.no_grad():
    for cam, rc in list(zip(sc.pipe.optCams(), sc.pipe.optRaycasts()))[:3]:
        r = sc.model.forward(cam, rc["depth_map"], rc["color_map"])["radiis"].cpu().numpy()
        v = r[r > 0]
        if os.environ.get("SAVE_RADII"):
            np.save(os.environ["SAVE_RADII"], r.astype(np.int16)); os.environ.pop("SAVE_RADII")
        h = [int(((v > lo) & (v <= hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        area = [(float((2.0 * v[(v > lo) & (v <= hi)]) ** 2).sum()) if False else float(((2.0 * v[(v > lo) & (v <= hi)].astype(np.float64)) ** 2).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        tot = sum(area)
        print("N %d visible %d; radius bins %s: counts %s; share of the summed box area %s" %
              (N, v.size, ["%d-%d" % (lo + 1, hi) for lo, hi in zip(edges[:-1], edges[1:])], h, ["%.1f%%" % (100 * a / tot) for a in area]), flush=True)
sc.close()
