"""distCUDA2: tiled brute force vs uniform-grid search over P (surface samples of the synthetic room), to place the switch-over
(GPS_KNN_GRID_MIN_POINTS).  usage (GPU box): python tools/probe/knn_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gps_slam_amd.gs_model import knn_mean_dist2
from tests.test_init_prune_raycast_gpu import _surface_points

pts, gen = _surface_points(1280, 720, seed=3)
for P in (500, 1000, 2000, 3000, 4096, 6000, 8000, 12000, 20000, 40000, 76800, 230400, 600000):
    x = pts[torch.randperm(pts.shape[0], generator=gen)[:P].sort().values].contiguous().cuda()
    out = {}
    for m in ("brute", "grid"):
        if m == "brute" and P > 250000:
            continue
        knn_mean_dist2(x, method=m)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 3 if (m == "brute" and P > 50000) else 20
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(n):
            knn_mean_dist2(x, method=m)
        ev[1].record()
        torch.cuda.synchronize()
        out[m] = ev[0].elapsed_time(ev[1]) / n * 1e3
    print("P = %7d: brute %10.1f us   grid %8.1f us" % (x.shape[0], out.get("brute", float("nan")), out["grid"]), flush=True)
