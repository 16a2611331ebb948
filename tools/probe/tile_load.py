"""Per-tile list lengths of the bench scene's optimise iteration (what bounds the forward rasterizer: one workgroup per tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from bench_kernels import _python_twin
W, H, NG = 640, 480, 200000
seq = bench.synthetic_sequence(W, H, 31, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, "cuda:0")
model.initOptimizers(-1, 1.0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B = model._B
n_isects = int(B["counts"][0])
off = B["tile_offsets"][:1200].cpu().long()
lens = torch.diff(torch.cat([off, torch.tensor([n_isects])])).float()
q = lambda p: float(torch.quantile(lens, p))
print("tiles %d  isects %d  mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (lens.numel(), n_isects, lens.mean(), q(.5), q(.9), q(.99), lens.max()))
g = lens.view(30, 40)
print("per tile row (mean):", [int(x) for x in g.mean(1)])
# one workgroup per tile, dealt round-robin to 8 XCDs x 32 CUs: work per CU if tile i lands on CU (i % 8, (i // 8) % 32)
cu = torch.zeros(256)
for i, l in enumerate(lens.tolist()):
    cu[(i % 8) * 32 + (i // 8) % 32] += l
print("per-CU sum of list lengths: mean %.0f max %.0f min %.0f" % (cu.mean(), cu.max(), cu.min()))
