"""Diagnostic: per-Gaussian intermediates of the prefetched forward (backward kernel's tail) vs the stand-alone forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import scenes
from tests.test_model_gpu import _two_cameras, DEV
from gps_slam_amd.gs_model import SLAMGaussianModel
N, W, H = 20000, 320, 240
g = scenes.random_gaussians(N, seed=21, scale_range=(0.004, 0.03))
T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
ms = []
for _ in range(2):
    m = SLAMGaussianModel(dict(fuse_sh_rest_adam=2), device=DEV)
    m.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                      featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"])))
    m.initOptimizers(-1, 1.0)
    ms.append(m)
cams = _two_cameras(W, H, seed=4)
gen = torch.Generator().manual_seed(9)
gts = [torch.rand((H, W, 3), generator=gen).to(DEV) for _ in range(2)]
base = torch.rand((H, W, 3), generator=gen).to(DEV)
ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
a, b = ms
a.train_step(cams[0], ref, base, gts[0], next_cam=cams[1])
b.train_step(cams[0], ref, base, gts[0])
torch.cuda.synchronize()
inter = ("radii", "means2d", "depths", "conics", "colors", "opacities", "records", "tiles_per_gauss")
held = {n: a._B[n][:N].clone() for n in inter}
# b's stand-alone forward for camera 1 on the same (equal) parameters
for n in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"):
    print("params equal after step 0:", n, torch.equal(getattr(a.opt_gs_params, n), getattr(b.opt_gs_params, n)))
b.forward(cams[1], ref, base)
torch.cuda.synchronize()
for n in inter:
    x, y = held[n].float(), b._B[n][:N].float()
    d = (x - y).abs()
    print("%-16s equal %s  max abs diff %.3e  rows differing %d  (scale %.3e)" % (n, torch.equal(held[n], b._B[n][:N]), float(d.max()), int((d.reshape(N, -1).max(1).values > 0).sum()), float(y.abs().max())))
