"""Per-kernel times of one optimise iteration on the bench scene after its settle frames (HIP events, 50 launches each).
usage: python tools/raster_bench.py [width height gaussians]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib

W, H, NG = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480, 200000)
seq = bench.synthetic_sequence(W, H, 31, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, "cuda:0", strip_backward=False)   # (group tables for the group kernel)
model.initOptimizers(-1, 1.0)
step = lambda: model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
step()
torch.cuda.synchronize()
B, st = model._B, model._step
counts = B["counts"].cpu().tolist()
print("N %d  n_isects %d  n_groups %d  visible %d" % (st.N, counts[0], counts[1], counts[3]))
# how much of the 2r x 2r boxes the backward walks can contribute at all: area of {alpha >= 1/255} = 2 pi T / sqrt(det Q),
# T = ln(255 opac), against the box area 4 r^2 (visible Gaussians)
r_ = B["radii"][:st.N].float()
con = B["conics"][:st.N]
T_ = torch.log(255.0 * B["opacities"][:st.N]).clamp_min(0)
det = (con[:, 0] * con[:, 2] - con[:, 1] ** 2).clamp_min(1e-12)
ell = 2 * 3.14159265 * T_ / det.sqrt()
vis = r_ > 0
print("sum ellipse area / sum box area = %.3f   (mean radius %.1f px, mean T %.2f)" % (float(ell[vis].sum() / (4 * r_[vis] ** 2).sum()),
                                                                                      float(r_[vis].mean()), float(T_[vis].mean())))
rv = B["radii"][:st.N][vis].long()
hist = torch.bincount(rv.clamp_max(40), minlength=41).tolist()
tot = sum(hist)
print("radius histogram (visible, clamped at 40): " + " ".join("%d:%d" % (r, c) for r, c in enumerate(hist) if c))
slots = [(4 * r * r) * c for r, c in enumerate(hist)]
print("share of pixel slots by radius: " + " ".join("%d:%.1f%%" % (r, 100.0 * s / max(1, sum(slots))) for r, s in enumerate(slots) if s))
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ref, N = rc["depth_map_clamped"], st.N
fwd = lambda: lib.gps_raster_ges_fwd_rec(N, p(B["records"]), p(ref), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]), p(B["counts"]),
                                         model.delta_depth, p(B["render_colors"]), p(B["weight_sum"]), sp)
bwd_of = lambda L: lambda: L.gps_raster_ges_bwd_gs(N, p(B["means2d"]), p(B["conics"]), p(B["colors"]), p(B["opacities"]), p(B["radii"]), p(ref), W, H,
                                        p(B["group_gs_ids"]), p(B["group_starts"]), p(B["counts"]), model.delta_depth,
                                        p(B["v_render_colors"]), p(B["v_render_alphas"]), p(B["v_means2d"]), p(B["v_conics"]),
                                        p(B["v_colors"]), p(B["v_opacities"]), 1, sp)
bwd = bwd_of(lib)
alt = os.environ.get("GPS_ALT_LIB")  # a second build of the library: same inputs through both, outputs compared bit for bit
if alt:
    from gps_slam_amd._lib import load_library
    lib2 = load_library(alt)
    fwd(); torch.cuda.synchronize()
    a_rc, a_ws = B["render_colors"].clone(), B["weight_sum"].clone()
    lib2.gps_raster_ges_fwd_rec(N, p(B["records"]), p(ref), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]), p(B["counts"]),
                                model.delta_depth, p(B["render_colors"]), p(B["weight_sum"]), sp)
    torch.cuda.synchronize()
    print("forward: %s == %s bit for bit: %s" % (os.environ.get("GPS_SLAM_HIP_LIB", "default"), alt,
                                                 bool(torch.equal(a_rc, B["render_colors"]) and torch.equal(a_ws, B["weight_sum"]))))
    grads = lambda: [B[k][:st.N].clone() for k in ("v_means2d", "v_conics", "v_colors", "v_opacities")]
    def zero():
        for k in ("v_means2d", "v_conics", "v_colors", "v_opacities"): B[k].zero_()
    zero(); bwd(); torch.cuda.synchronize(); ga = grads()
    zero(); bwd_of(lib2)(); torch.cuda.synchronize(); gb = grads()
    print("backward: max |difference| per array (float atomics: order-dependent last bits) %s" %
          ["%.2e / %.2e" % (float((x - y).abs().max()), float(x.abs().max())) for x, y in zip(ga, gb)])
    fwd2 = lambda: lib2.gps_raster_ges_fwd_rec(N, p(B["records"]), p(ref), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]), p(B["counts"]),
                                               model.delta_depth, p(B["render_colors"]), p(B["weight_sum"]), sp)
    # alternate the two builds (whichever is timed first after a pause reads ~10 % slow): best of three rounds each
    for what, f_a, f_b in (("raster bwd", bwd_of(lib2), bwd), ("raster fwd", fwd2, fwd)):
        ta, tb = [], []
        for _ in range(3):
            ta.append(1e6 * _time_launches(f_a, 50, stream)); tb.append(1e6 * _time_launches(f_b, 50, stream))
        print("%-12s %s %.1f us | default %.1f us" % (what, os.path.basename(alt), min(ta), min(tb)))
for name, fn, n in (("raster fwd (records)", fwd, 50), ("raster bwd (operator entry: 3 gathers)", bwd, 50), ("whole train step", step, 20)):
    print("%-42s %.1f us" % (name, 1e6 * _time_launches(fn, n, stream)))
scene.close()
