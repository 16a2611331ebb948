"""Times gps_tsdf_mesh_scene on the bench scene (640x480, 5 mm voxels, 120 fused frames)."""
import sys, os, time, torch, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gps_slam_amd.dist_util import scene_seed
W, H = 640, 480
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W, H, 32, 1000, scene_seed(0), 'cuda:0')
pipe.work_mode = "recon"
for i in range(32):
    pipe.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])
torch.cuda.synchronize()
tri, counts = eng.MeshScene(1 << 24)
torch.cuda.synchronize()
n, gen = counts.cpu().tolist()
blocks = eng.n_blocks - 1 - int(eng.counters_host()[0])
print("allocated blocks", blocks, "triangles", n, gen)
ws = torch.empty(8 << 20, dtype=torch.uint8, device='cuda:0')
from gps_slam_amd._lib import lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.gps_tsdf_mesh_scene(C.byref(eng.state), 1 << 24, tri.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
out_bytes = n * 84
in_bytes = blocks * 4096
print("mesh_scene %.3f ms; output %.1f MB -> %.0f GB/s written, voxels read %.1f MB" % (ms, out_bytes / 1e6, out_bytes / ms / 1e6, in_bytes / 1e6))
