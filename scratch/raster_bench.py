import sys, time, torch, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gps_slam_amd.dist_util import scene_seed
from gps_slam_amd._lib import lib
from bench_kernels import _time_launches
W,H=640,480
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W,H,32,200000,scene_seed(0),'cuda:0')
for i in range(31):
    pipe.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])
cam = pipe.opt_cam_list[-1]; rc = pipe.opt_raycast_list[-1]
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
counts = B["counts"].cpu().tolist(); print('counts', counts, 'N', st.N)
stream = torch.cuda.current_stream(); sp = C.c_void_p(stream.cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
ref = rc["depth_map_clamped"]; N=st.N
def fwd1(): lib.gps_raster_ges_fwd(N, ptr(B["means2d"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]), ptr(ref), W, H, 16, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]), ptr(B["counts"]), model.delta_depth, ptr(B["render_colors"]), ptr(B["weight_sum"]), None, sp)
def fwd2(): lib.gps_raster_ges_fwd_rec(N, ptr(B["records"]), ptr(ref), W, H, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]), ptr(B["counts"]), model.delta_depth, ptr(B["render_colors"]), ptr(B["weight_sum"]), sp)
def bwd(): lib.gps_raster_ges_bwd_gs(N, ptr(B["means2d"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]), ptr(B["radii"]), ptr(ref), W, H, ptr(B["group_gs_ids"]), ptr(B["group_starts"]), ptr(B["counts"]), model.delta_depth, ptr(B["v_render_colors"]), ptr(B["v_render_alphas"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]), ptr(B["v_opacities"]), 1, sp)
pp = model.opt_gs_params
cam_d = cam.toGPU()
def pre_f(): lib.gps_gauss_preprocess_fwd(N, pp.K, 3, ptr(pp._buf["means"]), ptr(pp._buf["scales"]), ptr(pp._buf["quats"]), ptr(pp._buf["opacities"]), ptr(pp._buf["featuresDc"]), ptr(pp._buf["featuresRest"]), ptr(cam_d["viewmat"]), ptr(cam_d["K"]), ptr(cam_d["cam_pos"]), W, H, 0.3, 0.01, 1e10, 0.0, 100, ptr(B["radii"]), ptr(B["means2d"]), ptr(B["depths"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]), ptr(B["records"]), sp)
o = model._opt
def pre_b(): lib.gps_gauss_preprocess_bwd(N, pp.K, 3, ptr(pp._buf["means"]), ptr(pp._buf["scales"]), ptr(pp._buf["quats"]), ptr(pp._buf["opacities"]), ptr(pp._buf["featuresDc"]), ptr(pp._buf["featuresRest"]), ptr(cam_d["viewmat"]), ptr(cam_d["K"]), ptr(cam_d["cam_pos"]), W, H, 0.3, ptr(B["radii"]), ptr(B["conics"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]), ptr(B["v_opacities"]), ptr(o["g"][0]), ptr(o["g"][1]), ptr(o["g"][2]), ptr(o["g"][5]), ptr(o["g"][3]), ptr(o["g"][4]), sp)
for name, fn in (('fwd_lds', fwd1), ('fwd_rec_var', fwd2), ('bwd', bwd), ('pre_fwd', pre_f), ('pre_bwd', pre_b)):
    print('%-10s %.1f us' % (name, 1e6*_time_launches(fn, 50, stream)))
offs = B["tile_offsets"].cpu().numpy().astype(np.int64); ni=int(counts[0])
d = np.diff(np.concatenate([offs, [ni]]))
print('tile depth: mean %.0f max %d p99 %.0f' % (d.mean(), d.max(), np.percentile(d,99)))
r = B["radii"][:N].cpu().numpy(); print('radius mean %.1f max %d; visible %d' % (r[r>0].mean(), r.max(), (r>0).sum()))
