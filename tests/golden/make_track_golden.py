"""Generates tests/golden/track_320x240.npz: poses estimated by the REFERENCE's own ITMLib CPU engine with its default
depth-only ExtendedTracker switched on (oracle/_ref/itm_ref in `track` mode, built by oracle/ref_build.sh from
/root/reference), on a synthetic sequence.  Only data is stored: per-frame M / invM (ORUtils layout), trackerScore and the
tracker's framesProcessed counter.  Run from the repo root:  python tests/golden/make_track_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdf_ref as R  # noqa: E402
from tests import synth  # noqa: E402

W, H, N, STEP = 320, 240, 8, 0.5
VOXEL, MU, VFMIN, VFMAX = 0.01, 0.04, 0.2, 10.0
assert R.available(), "oracle/_ref/itm_ref missing: bash oracle/ref_build.sh"
seq = synth.make_sequence(W, H, N, step_deg=STEP)
ref = R.run(seq, VOXEL, MU, VFMIN, VFMAX, track=True)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "track_320x240.npz")
np.savez_compressed(out, W=W, H=H, n_frames=N, step_deg=STEP, voxel=VOXEL, mu=MU, vf_min=VFMIN, vf_max=VFMAX,
                    M=np.stack([ref[("M", f)].reshape(-1) for f in range(N)]),
                    invM=np.stack([ref[("invM", f)].reshape(-1) for f in range(N)]),
                    score=np.stack([ref[("trk_score", f)] for f in range(N)]),
                    n_visible=np.array([ref[("counts", f)][0] for f in range(N)]),
                    last_free_block=np.array([ref[("counts", f)][1] for f in range(N)]))
print("wrote", out, os.path.getsize(out), "bytes")
