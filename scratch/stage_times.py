import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gps_slam_amd.dist_util import scene_seed
W,H=640,480
K=60
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W,H,K+1,200000,scene_seed(0),'cuda:0')
T={}
def timed(name, fn):
    torch.cuda.synchronize(); t=time.perf_counter(); r=fn(); torch.cuda.synchronize(); T[name]=T.get(name,0)+time.perf_counter()-t; return r
import math
for i in range(K):
    cam=cams[i]
    def tsdf():
        pipe.curr_frame_id=i
        M,invM=pipe.tsdf.ProcessFrame(rgb_dev[i],depth_dev[i],cam.c2w.numpy())
        cam.c2w_slam=torch.from_numpy(invM.reshape(4,4).T.copy()); cam.invalidate(); pipe.curr_cam=cam; cam.toGPU(); pipe.updateFrameList()
    if i>=20: timed('tsdf_frame', tsdf)
    else: tsdf()
    if i%10==0 and i>0:
        if i>=20:
            timed('localFrameRaycast', pipe.localFrameRaycast); timed('keyFrameRaycast', pipe.keyFrameRaycast)
            rm=pipe.localframe_raycast_window[-1]; c=pipe.cfg
            depth, color, vertex = rm["depth_map"], rm["color_map"], rm["vertex_map"]
            res=timed('  ing.forward', lambda: model.forward(cam, depth, color))
            def masks():
                valid = (depth > c["depth_vis_min"]) & (depth < c["depth_vis_max"])
                valid = valid & ~((vertex.sum(2) == 0).unsqueeze(-1))
                err = torch.mean(torch.abs(res["rgb"] - cam.image), -1, True)
                return (err > c["color_error_thres"]) & valid & (res["alpha"] < c["alpha_vis_max"])
            mask=timed('  ing.masks', masks)
            from gps_slam_amd.slam_pipeline import compute_normal_map
            rm["normal_map"]=timed('  ing.normal', lambda: compute_normal_map(vertex))
            m = mask.expand(H, W, 3)
            verts=timed('  ing.masked_select', lambda: [torch.masked_select(t, m).reshape(-1,3) for t in (vertex, cam.image, rm["normal_map"])])
            nn=verts[0].shape[0]
            perm=timed('  ing.randperm', lambda: torch.randperm(nn, device=verts[0].device, generator=pipe.gen)[:int(nn*0.25)])
            sel=timed('  ing.index', lambda: [v[perm].contiguous() for v in verts])
            new=timed('  ing.init_params', lambda: model.init_params(*sel))
            timed('  ing.add_params', lambda: model.add_params(new))
            timed('localOptimize', pipe.localOptimize); timed('removeRedundantGs', pipe.removeRedundantGs)
        else:
            pipe.localFrameRaycast(); pipe.keyFrameRaycast(); pipe.initNewGaussians(pipe.localframe_raycast_window[-1]); pipe.localOptimize(); pipe.removeRedundantGs()
n=K-20
for k,v in T.items(): print('%-22s %.3f ms/frame' % (k, 1000*v/n))
print('sum %.3f ms/frame' % (1000*sum(T.values())/n), 'N', model.getGaussianNum(), pipe.stats)
# async (no per-stage sync) whole-loop time for the same frames is what bench.py reports
