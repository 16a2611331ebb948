"""Which stage of building a scene is followed, ~100 ms later, by the release of the process's held queues?  The stages of
bench.Scene.__init__ are run with time stamps; the run starts right after; the stall's END time minus each stage's end time is printed
(the stage whose distance is a constant ~100 ms is the trigger).  SLEEP_AFTER=<stage letter> inserts 0.15 s after that stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import gps_slam_amd._host as H_

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = int(os.environ.get("FRAMES", "300"))
reps = int(os.environ.get("REPS", "10"))
after = os.environ.get("SLEEP_AFTER", "")
seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
W, H = 640, 480
for rep in range(reps):
    T = {}
    def mark(k):
        torch.cuda.synchronize(); T[k] = time.perf_counter()
        if k in after: time.sleep(0.15)
    reader = H_.DatasetReader(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    cams = []
    for k in range(n):
        c = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
        c.id = k
        c.image = torch.as_tensor(seq["rgb"][k]); c.depth = torch.as_tensor(seq["depth"][k].view(np.int16))
        reader.addTrainCamera(c)
        pc = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
        pc.id = k
        cams.append(pc)
    mark("A")
    cli = H_.createTsdfEngine(reader, dict(voxel_size=0.005, trunc_dist=0.02, viewFrustum_min=0.2, viewFrustum_max=10.0, use_gt_pose=0))
    mark("B")
    model = H_.SLAMGaussianModel()
    model.loadConfig(dict(capacity=1 << 19, isect_capacity=8 << 20))
    mark("C")
    pipe = H_.SLAMPipeline(1234)
    pipe.setTsdfEngine(cli); pipe.setModel(model)
    pipe.loadConfig(dict(keyframe_theta_thres=1.0, keyframe_trans_thres=0.02))
    pipe.overlap_mapping = True; pipe.mapping_thread = True
    mark("D")
    model.reserveWorkspace(W, H)
    mark("E")
    pipe.keep_frame_ms = True
    t_run = time.perf_counter()
    tm = pipe.SLAMTrainCamsTimed(model, cams)
    ms = np.asarray(pipe.frame_ms)
    w = int(ms[:60].argmax())
    end = t_run + 1e-3 * ms[:w + 1].sum()
    print("rep %2d: %.0f frames/s, stall %.1f ms at frame %d; its end is %s ms after the stage ends" %
          (rep, tm.fps(), ms[w], w, " ".join("%s:%.0f" % (k, 1e3 * (end - v)) for k, v in T.items())), flush=True)
    cli.Shutdown(); del cli, pipe, model, reader, cams
    torch.cuda.empty_cache()
