"""A/B of the forward rasterizer: persistent launch (raster_ges_fwd_pp_kernel) against one workgroup per tile
(raster_ges_fwd_pk_kernel) on the bench scene's state, alternating, best of 3 x 50 launches (HIP events), with and without
the binning's tile_order; the two outputs compared bit for bit.
usage: python tools/probe/fwd_pp_ab.py [width height gaussians [fx fy cx cy]]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib

a = sys.argv[1:]
W, H, NG = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (640, 480, 200000)
intr = tuple(float(x) for x in a[3:7]) if len(a) >= 7 else None
dev = "cuda:0"
bench.prime(dev)
seq = bench.synthetic_sequence_device(W, H, 31, 1234, dev, intrinsics=intr)
seeds = bench.seed_gaussians(seq, NG, 1234, dev)
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, dev)
model.initOptimizers(-1, 1.0)
lib.gps_set_frame_chain_reserve(0)
step = lambda: model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
step()
torch.cuda.synchronize()
B, st = model._B, model._step
counts = B["counts"].cpu().tolist()
N = st.N
print("%dx%d  N %d  n_isects %d  visible %d  tiles %d" % (W, H, N, counts[0], counts[3], ((W + 15) // 16) * ((H + 15) // 16)))
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ref = rc["depth_map_clamped"]
order = C.c_void_p(lib.gps_isect_workspace_tile_order(p(B["workspace"]), N, st.isect_capacity))


def fwd(persistent, ordered):
    def f():
        lib.gps_set_raster_fwd_persistent(persistent)
        lib.gps_raster_ges_fwd_rec_ordered(N, p(B["records"]), p(ref), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]), p(B["counts"]),
                                           model.delta_depth, p(B["render_colors"]), p(B["weight_sum"]), order if ordered else C.c_void_p(0), sp)
    return f


outs = {}
for key in ((0, 0), (1, 0), (1, 1), (0, 1)):
    B["render_colors"].fill_(float("nan")); B["weight_sum"].fill_(float("nan"))
    fwd(*key)()
    torch.cuda.synchronize()
    outs[key] = (B["render_colors"].clone(), B["weight_sum"].clone())
for key in ((1, 0), (1, 1), (0, 1)):
    print("persistent=%d ordered=%d == per-tile row-major bit for bit: %s" % (key[0], key[1], bool(torch.equal(outs[key][0], outs[(0, 0)][0]) and torch.equal(outs[key][1], outs[(0, 0)][1]))))
best = {}
for _ in range(3):
    for key in ((0, 0), (1, 0), (0, 1), (1, 1)):
        t = 1e6 * _time_launches(fwd(*key), 50, stream)
        best[key] = min(best.get(key, 1e9), t)
for key, t in best.items():
    print("forward persistent=%d ordered=%d: %.1f us" % (key[0], key[1], t))
lib.gps_set_raster_fwd_persistent(1)
t_on = 1e6 * _time_launches(step, 20, stream)
lib.gps_set_raster_fwd_persistent(0)
t_off = 1e6 * _time_launches(step, 20, stream)
lib.gps_set_raster_fwd_persistent(0)   # (the shipped default)
print("whole train step: persistent %.1f us | per-tile %.1f us" % (t_on, t_off))
scene.close()
