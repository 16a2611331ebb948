import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, numpy as np
import bench
from gps_slam_amd.dist_util import scene_seed
from gps_slam_amd._lib import lib, check
from bench_kernels import _time_launches
W,H=640,480
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W,H,32,200000,scene_seed(0),'cuda:0')
for i in range(31):
    pipe.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])
cam = pipe.opt_cam_list[-1]; rc = pipe.opt_raycast_list[-1]
model.initOptimizers(-1, 3.3)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
st = model._step; stream = torch.cuda.current_stream(); sp = C.c_void_p(stream.cuda_stream)
print("N", st.N)
for mode in (0, 1, 2, 0, 1, 2):
    st.fuse_sh_rest_adam = mode
    k = [1]
    def fn():
        k[0] += 1
        check(lib.gps_splat_train_step(C.byref(st), k[0], sp), "ts")
    print("mode", mode, "%.1f us per iteration" % (1e6 * _time_launches(fn, 40, stream)))
