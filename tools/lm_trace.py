"""Accept / reject pattern of the depth tracker's LM loop on the bench's sequence, from the CPU oracle (oracle/tsdf_oracle.c:
orc_track_trace): per frame one string "3:AAR 2:AA 1:AARA 0:AA" (level : decisions).  Counts how many evaluations follow a
rejection (those are the ones whose pose is known BEFORE the rejected evaluation returns: the reject branch reads only the last
good state), split by whether the follower runs at the same level.  usage: lm_trace.py [frames, default 40]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import bench
from oracle import tsdf_ref as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = 640, 480
seq = bench.synthetic_sequence(W, H, n, 1234)
o = R.TsdfOracle(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
cfg = R.track_config(0.2, 10.0)
lib = R._lib()
lib.orc_track_trace.restype = C.c_char_p
tot = rej = follow_same = follow_next = follow_none = 0
chains = {}
for k in range(n):
    o.process_frame_tracked(seq["rgb"][k], seq["depth"][k], cfg)
    tr = lib.orc_track_trace().decode()
    if k == 0:
        continue
    print("frame %3d  %s" % (k, tr))
    levels = [p.split(":")[1] for p in tr.split()]
    flat = [(li, ch) for li, s in enumerate(levels) for ch in s]
    tot += len(flat)
    run = 0
    for i, (li, ch) in enumerate(flat):
        if ch == "R":
            rej += 1
            run += 1
            if i + 1 == len(flat):
                follow_none += 1
            elif flat[i + 1][0] == li:
                follow_same += 1
            else:
                follow_next += 1
        else:
            if run:
                chains[run] = chains.get(run, 0) + 1
            run = 0
    if run:
        chains[run] = chains.get(run, 0) + 1
f = n - 1
print("per frame: %.1f evaluations, %.1f rejections; the evaluation after a rejection runs at the same level %.1f, at the next level %.1f, "
      "does not exist %.1f" % (tot / f, rej / f, follow_same / f, follow_next / f, follow_none / f))
print("chains of consecutive rejections (length: count):", dict(sorted(chains.items())))
