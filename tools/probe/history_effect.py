"""Does what a process did BEFORE change the overlap schedule's whole-run speed?  (bench.py's whole-run leg measured 900 frames/s after
its other_configs leg and 1,100-1,150 without it.)  PRE = none | small:<n> (n tiny overlap scenes first) | big (one 1280x720 / 400 k scene)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
pre = os.environ.get("PRE", "none")
kf = dict(keyframe_theta=1.0, keyframe_trans=0.02)
if pre.startswith("small"):
    n = int(pre.split(":")[1])
    seq = bench.synthetic_sequence(64, 48, 21, 1)
    seeds = bench.seed_gaussians(seq, 500, 1, dev)
    for k in range(n):
        s = bench.Scene(seq, seeds, 1, False, overlap=True, n_frames=21, **kf)
        s.run(0, 21); torch.cuda.synchronize(); s.close(); del s
elif pre == "big":
    seq3 = bench.synthetic_sequence_device(1280, 720, 60, 1234, dev, intrinsics=(605.0, 605.0, 635.3, 366.5))
    seeds3 = bench.seed_gaussians(seq3, 400000, 1234, dev)
    for ov in (False, True):
        s = bench.Scene(seq3, seeds3, 1234, False, overlap=ov, n_frames=60, **kf)
        s.run(0, 60); torch.cuda.synchronize(); s.close(); del s
        torch.cuda.empty_cache()
    del seq3, seeds3
torch.cuda.empty_cache()
n = 1000
seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
for rep in range(2):
    sc = bench.Scene(seq, None, 1234, False, overlap=True, n_frames=n, **kf)
    torch.cuda.synchronize()
    tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
    print("PRE=%s rep %d: whole-run overlap %.1f frames/s, slowest frame %.1f ms (frame %d), reserved %.1f GB" %
          (pre, rep, tm.fps(), tm.max_frame_after_30, tm.max_frame_id, torch.cuda.memory_reserved() / 2**30), flush=True)
    sc.close(); del sc
    torch.cuda.empty_cache()
