"""Where a tracker evaluation's time goes (gps_track_poll_phases) on the bench sequence, tracker alone (no map update beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json, time
import numpy as np, torch
import bench
from bench_kernels import _fusion_timings
seq = bench.synthetic_sequence_device(640, 480, int(os.environ.get("NFRAMES", 60)), 1234, "cuda:0")
f = _fusion_timings(seq, False, "cuda:0", n_sub=int(os.environ.get("NSUB", 40)))
print(json.dumps({k: f[k] for k in ("tracked_ms_per_frame", "untracked_ms_per_frame", "tracking_ms_per_frame", "evals_per_frame", "poll_spin_s", "poll_eval_s", "poll_phases")}, indent=1))
