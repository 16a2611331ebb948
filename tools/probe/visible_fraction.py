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
    fr, tiles = [], []
    T = 128
    with torch.no_grad():
        for cam, rc in views:
            res = sc.model.forward(cam, rc["depth_map"], rc["color_map"])
            vis = (res["radiis"] > 0)
            fr.append(float(vis.sum().item()) / N)
            pad = (-N) % T
            tiles.append(torch.nn.functional.pad(vis, (0, pad)).view(-1, T).any(1).cpu().numpy())
    tiles = np.stack(tiles)                      # [views, n_tiles]: the tile holds a Gaussian the view sees
    rng = np.random.default_rng(0)
    saved = []
    for trial in range(200):                     # 20 iterations, each a random view of the update (RandomSelector: without replacement per round)
        order = np.concatenate([rng.permutation(len(views)) for _ in range(3)])[:20]
        seen = np.zeros(tiles.shape[1], bool)
        b = 0.0
        for v in order:
            now = tiles[v]
            dark = ~seen & ~now                  # never seen in this update, not seen now: nothing moves but the parameter read for the next view
            first = ~seen & now                  # first touch: no moment reads
            b += dark.mean() * (5.0 / 6.0) + first.mean() * (2.0 / 6.0)
            seen |= now
        saved.append(b / 20.0)
    print("after frame %4d: N = %6d, %d views, visible fraction min %.3f mean %.3f max %.3f; tiles of %d with a visible Gaussian per view %.3f; "
          "Adam bytes a tile-level skip saves over an update's 20 iterations: %.1f %%" % (lo + 99, N, len(views), min(fr), np.mean(fr), max(fr), T, tiles.mean(), 100 * np.mean(saved)), flush=True)
sc.close()
