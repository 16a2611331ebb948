#!/usr/bin/env python
"""Where the time of the whole-sequence run goes (bench.whole_run): host wall of every processFrame call of SLAMTrainCams from
frame 0, per schedule, summed per block of 100 frames, plus the slowest calls.  python tools/whole_run_trace.py [frames] [W H]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
    device = "cuda:0"
    torch.cuda.set_device(0)
    bench.prime(device)
    seq = bench.synthetic_sequence_device(W, H, n, 1234, device)
    out = {}
    for sched in (os.environ.get("TRACE_SCHEDULES", "sequential,overlap").split(",")):
        torch.cuda._sleep(1)   # phase marker for tools/prof_summary.py
        sc = bench.Scene(seq, None, 1234, False, overlap=sched == "overlap", n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
        sc.pipe.keep_frame_ms = True
        if os.environ.get("TRACE_REPORT_MS"):
            sc.pipe.frame_report_ms = float(os.environ["TRACE_REPORT_MS"])
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
        ms = np.asarray(sc.pipe.frame_ms, np.float64)
        blocks = [float(ms[k:k + 100].sum()) for k in range(0, n, 100)]
        worst = np.argsort(-ms)[:12]
        out[sched] = dict(fps=tm.fps(), total_ms=tm.slam_total, sum_frame_ms=float(ms.sum()), block_ms=[round(b, 1) for b in blocks],
                          worst=[(int(i), round(float(ms[i]), 2)) for i in worst], gaussians=int(sc.model.getGaussianNum()),
                          mallocs=int(torch.cuda.memory_stats().get("num_device_alloc", 0) - m0),
                          kf_mean=float(ms[10::10].mean()), nonkf_mean=float(np.delete(ms, np.arange(0, n, 10)).mean()))
        print(sched, json.dumps(out[sched]), flush=True)
        torch.cuda._sleep(1)
        sc.close()
        del sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
