"""Times the `raw` render method kernels on the bench scene's Gaussians (640x480)."""
import sys, torch, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gps_slam_amd.dist_util import scene_seed
from gps_slam_amd import gsplat_ops as ops
from bench_kernels import _time_launches
W, H = 640, 480
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W, H, 32, 200000, scene_seed(0), 'cuda:0')
for i in range(31):
    pipe.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])
cam = pipe.opt_cam_list[-1]; rc = pipe.opt_raycast_list[-1]
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
stream = torch.cuda.current_stream()
m2, con, col, op = B["means2d"][:N][None], B["conics"][:N][None], B["colors"][:N][None], B["opacities"][:N].view(1, N)
radii, depths = B["radii"][:N][None], B["depths"][:N][None]
isect = ops.isect_tiles(m2, radii, depths, 16, 40, 30)
ni = isect.sizes()[0]
print("N", N, "n_isects", ni)
rcol, ra, last = ops.rasterize_to_pixels_fwd(m2, con, col, op, None, W, H, 16, isect)
v_rc = torch.randn_like(rcol); v_ra = torch.randn_like(ra)
depth_used = (last[0].view(30, 16, 40, 16).amax((1, 3)) - isect.isect_offsets[0]).clamp_min(0).float()
d = np.diff(np.concatenate([isect.isect_offsets.cpu().numpy().ravel().astype(np.int64), [ni]]))
print('tile list: mean %.0f max %d; walked (to last contributor): mean %.0f' % (d.mean(), d.max(), float(depth_used.mean())))
print('alpha mean %.3f' % float(ra.mean()))
fns = dict(
    isect=lambda: ops.isect_tiles(m2, radii, depths, 16, 40, 30),
    isect_nodepth=lambda: ops.isect_tiles_no_depth(m2, radii, 16, 40, 30),
    fwd=lambda: ops.rasterize_to_pixels_fwd(m2, con, col, op, None, W, H, 16, isect),
    bwd=lambda: ops.rasterize_to_pixels_bwd(m2, con, col, op, None, W, H, 16, isect, ra, last, v_rc, v_ra),
    bwd_abs=lambda: ops.rasterize_to_pixels_bwd(m2, con, col, op, None, W, H, 16, isect, ra, last, v_rc, v_ra, absgrad=True),
)
for name, fn in fns.items():
    print('%-14s %.1f us' % (name, 1e6 * _time_launches(fn, 30, stream)))
