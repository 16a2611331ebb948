import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gps_slam_amd.dist_util import scene_seed
from gps_slam_amd.slam_pipeline import compute_normal_map
W,H=640,480
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W,H,32,200000,scene_seed(0),'cuda:0')
for i in range(31):
    cam=cams[i]
    pipe.curr_frame_id=i
    M,invM=pipe.tsdf.ProcessFrame(rgb_dev[i],depth_dev[i],cam.c2w.numpy())
    cam.c2w_slam=torch.from_numpy(invM.reshape(4,4).T.copy()); cam.invalidate(); pipe.curr_cam=cam; cam.toGPU(); pipe.updateFrameList()
    if i in (10,20):
        pipe.localFrameRaycast(); pipe.keyFrameRaycast(); pipe.initNewGaussians(pipe.localframe_raycast_window[-1]); pipe.localOptimize(); pipe.removeRedundantGs()
pipe.localFrameRaycast(); pipe.keyFrameRaycast()
rm=pipe.localframe_raycast_window[-1]
def tm(name, fn, n=1):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=fn()
    torch.cuda.synchronize(); print('%-28s %.3f ms'%(name,1000*(time.perf_counter()-t)/n)); return r
c=pipe.cfg; cam=pipe.curr_cam
depth, color, vertex = rm["depth_map"], rm["color_map"], rm["vertex_map"]
res=tm('model.forward', lambda: model.forward(cam, depth, color))
def masks():
    valid = (depth > c["depth_vis_min"]) & (depth < c["depth_vis_max"])
    valid = valid & ~((vertex.sum(2) == 0).unsqueeze(-1))
    err = torch.mean(torch.abs(res["rgb"] - cam.image), -1, True)
    return (err > c["color_error_thres"]) & valid & (res["alpha"] < c["alpha_vis_max"])
mask=tm('masks', masks)
rm["normal_map"]=tm('normal_map', lambda: compute_normal_map(vertex))
m = mask.expand(H, W, 3)
verts=tm('masked_select x3', lambda: [torch.masked_select(t, m).reshape(-1,3) for t in (vertex, cam.image, rm["normal_map"])])
n=verts[0].shape[0]; print('n masked', n)
perm=tm('randperm', lambda: torch.randperm(n, device=verts[0].device, generator=pipe.gen)[:int(n*0.25)])
sel=tm('index', lambda: [v[perm] for v in verts])
new=tm('init_params', lambda: model.init_params(*sel))
tm('add_params', lambda: model.add_params(new))
tm('whole initNewGaussians', lambda: pipe.initNewGaussians(rm))
