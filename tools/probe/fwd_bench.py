"""Forward rasterizer (gps_raster_ges_fwd_rec) of the bench scene's optimise iteration: the shipped build and variant builds
(tools/probe/variant.py) timed alternately on one box (HIP events, 50 launches, best of 3), outputs compared with the shipped one.
usage: python tools/probe/fwd_bench.py [variant .so ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib, load_library

W, H, NG = 640, 480, 200000
seq = bench.synthetic_sequence(W, H, 31, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, "cuda:0")
model.initOptimizers(-1, 1.0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
print("N %d counts %s" % (N, B["counts"].tolist()))
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ref = rc["depth_map_clamped"]


def fwd(L, out_rc, out_ws):
    return lambda: L.gps_raster_ges_fwd_rec(N, p(B["records"]), p(ref), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]), p(B["counts"]),
                                            model.delta_depth, p(out_rc), p(out_ws), sp)


libs = [("shipped", lib)] + [(os.path.basename(a), load_library(a)) for a in sys.argv[1:]]
outs = {}
for name, L in libs:
    o = (torch.zeros_like(B["render_colors"]), torch.zeros_like(B["weight_sum"]))
    assert fwd(L, *o)() == 0
    torch.cuda.synchronize()
    outs[name] = o
for name, o in outs.items():
    d = (o[0] - outs["shipped"][0]).abs().max().item()
    print("%-28s max |render - shipped| %.3e (max |render| %.3e) bit-equal %s" % (name, d, outs["shipped"][0].abs().max().item(),
                                                                                 torch.equal(o[0], outs["shipped"][0])))
for _ in range(2):
    line = ""
    for name, L in libs:
        o = outs[name]
        line += " fwd[%s] %.1f us |" % (name, min(1e6 * _time_launches(fwd(L, *o), 50, stream) for _ in range(3)))
    print(line)
# how much of the time is load imbalance: the tile list lengths, and the same kernel on a table with all lists equally long
to = B["tile_offsets"]
n_is = int(B["counts"][0])
nt = to.numel()
cnt = torch.diff(torch.cat([to, torch.tensor([n_is], device=to.device, dtype=to.dtype)])).float()
q = torch.quantile(cnt, torch.tensor([0.0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], device=cnt.device)).tolist()
print("tile list lengths: mean %.0f  quantiles 0/25/50/75/90/99/100 %s  batches of 512 per tile: max %d" %
      (cnt.mean().item(), [int(v) for v in q], int((cnt.max().item() + 511) // 512)))
uni = (torch.arange(nt, device=to.device, dtype=torch.int64) * n_is // nt).to(to.dtype)
keep = to.clone()
o = outs["shipped"]
o2 = (o[0].clone(), o[1].clone())
t_real = min(1e6 * _time_launches(fwd(lib, *o2), 50, stream) for _ in range(3))
to.copy_(uni)
t_uni = min(1e6 * _time_launches(fwd(lib, *o2), 50, stream) for _ in range(3))
to.copy_(keep)
print("forward %.1f us; with all %d tile lists equally long (same total) %.1f us" % (t_real, nt, t_uni))
# latency- or throughput-bound?  The same kernel with only the first M tiles' lists non-empty (the others return at once): if the
# time does not grow with M up to one resident round (768 workgroups), a wave's own dependent chain sets it
cnt0 = B["counts"].clone()
line = "non-empty tiles -> us:"
for M in (64, 256, 512, 768, 1024, 1200):
    t2 = keep.clone()
    if M < nt:
        t2[M:] = keep[M]
        B["counts"][0] = int(keep[M])
    to.copy_(t2)
    line += "  %d: %.1f" % (M, min(1e6 * _time_launches(fwd(lib, *o2), 50, stream) for _ in range(3)))
    B["counts"].copy_(cnt0)
to.copy_(keep)
print(line)
step = lambda: model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
print("whole train step (shipped) %.1f us" % (1e6 * _time_launches(step, 20, stream)))
scene.close()
