import sys, json, time
sys.path.insert(0, '/root/repo')
import argparse, torch
import bench
dev = "cuda:0"
bench.prime(dev)
args = argparse.Namespace(width=640, height=480, gt_pose=False, keyframe_theta=1.0, keyframe_trans=0.02)
t=time.time(); d = bench.detail_run(args, 1234, dev); print(json.dumps(d), time.time()-t)
t=time.time(); r = bench.ref_threshold_run(args, 1234, dev); print(json.dumps(r), time.time()-t)
args.gt_pose=True
d = bench.detail_run(args, 1234, dev); print("gt pose:", json.dumps(d))
