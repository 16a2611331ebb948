#!/usr/bin/env python
"""How many Gaussians does a map update touch at all?  After the whole-sequence run (bench.whole_run's scene: 1,000 frames from an
empty model) the last update's <= 9 optimise cameras are rendered once each on a Python twin of the model: per camera the visible
fraction (radius > 0), and the fraction of the model that is visible in NONE of them -- Gaussians whose gradient is exactly zero in
every iteration of the update, so that Adam (state re-created per update: m = v = 0) leaves them bit-for-bit unchanged."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from bench_kernels import _python_twin  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dev = "cuda:0"
    torch.cuda.set_device(0)
    bench.prime(dev)
    seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
    sc = bench.Scene(seq, None, 1234, False, overlap=False, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
    for stop in range(200, n + 1, 200):
        sc.run(stop - 200, stop)
        torch.cuda.synchronize()
        from gps_slam_amd.gs_model import Camera
        model, _, _ = _python_twin(sc, dev)
        N = model.opt_gs_params.N if hasattr(model.opt_gs_params, "N") else sc.model.getGaussianNum()
        union = torch.zeros(N, dtype=torch.bool, device=dev)
        fr = []
        with torch.no_grad():
            for c, rc in zip(sc.pipe.optCams(), sc.pipe.optRaycasts()):
                cam = Camera(c.id, c.width, c.height, c.fx, c.fy, c.cx, c.cy, c.c2w.cpu().numpy(), image=c.image, device=dev)
                cam.c2w_slam = c.c2w_slam.cpu()
                cam.invalidate()
                model.forward(cam, rc["depth_map"], rc["color_map"])
                vis = model._B["radii"][:N] > 0
                union |= vis
                fr.append(float(vis.float().mean()))
        print("frame %d: N = %d, visible per camera %s, union %.3f -> untouched by the whole update %.3f"
              % (stop, N, " ".join("%.2f" % f for f in fr), float(union.float().mean()), 1.0 - float(union.float().mean())), flush=True)
        del model
    sc.close()


if __name__ == "__main__":
    main()
