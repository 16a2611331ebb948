"""How many of a frame's visible blocks does IntegrateIntoScene leave untouched?  (Would a block-level skip of the 4 KB read pay?)
The bench scene after n frames: the volume before / after one more frame, blocks with any changed byte against the visible count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gps_slam_amd.tsdf_engine import TsdfEngine

W, H, n = 640, 480, int(os.environ.get("NFRAMES", 60))
seq = bench.synthetic_sequence_device(W, H, n + 3, 1234, "cuda:0")
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device="cuda:0")
up = lambda i: (torch.from_numpy(seq["rgb"][i]).cuda().contiguous(), torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous())
for i in range(n):
    eng.ProcessFrame(*up(i), seq["c2w"][i])
for i in range(n, n + 3):
    before = eng.vba.clone()
    eng.ProcessFrame(*up(i), seq["c2w"][i])
    torch.cuda.synchronize()
    changed = (eng.vba.view(-1, 4096) != before.view(-1, 4096)).any(1)
    vox = (eng.vba.view(-1, 8) != before.view(-1, 8)).any(1).view(-1, 512)
    nvis = int(eng.counters.cpu()[2])
    nb = int(changed.sum())
    per = vox[changed].float().sum(1)
    print("frame %d: visible blocks %d, blocks with a changed voxel %d (%.1f %%), changed voxels per changed block: mean %.0f of 512, blocks with < 64 changed voxels %.1f %%"
          % (i, nvis, nb, 100.0 * nb / nvis, float(per.mean()), 100.0 * float((per < 64).float().mean())))
