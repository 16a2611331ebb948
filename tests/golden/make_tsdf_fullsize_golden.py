"""Generates tests/golden/refdigest_tsdf_640x480_v5mm.npz: the REFERENCE's own ITMLib CPU engine (oracle/_ref/itm_ref, built by
oracle/ref_build.sh from /root/reference) at BASELINE size -- 640x480, 5 mm voxels, mu = 2 cm -- on the synthetic sequence
tests/test_tsdf_gpu.py::test_engine_matches_oracle_full_size fuses (synth.make_sequence(640, 480, 3, step_deg=1.0)).

Only digests are stored (the images are 5 MB each): per frame the engine counters and CRC-32s of the visible list, the hash
rows, the visible-type words, the float depth image, the min/max window the raycaster consumes, the raycast / ICP images and
the voxel payload; the same for one free view (runRaycast) after the last frame; and, from a second run with the depth-only
ExtendedTracker ON, the estimated poses, tracker score and allocation counters of the same three frames.  With it the CPU
restatement AND the HIP engine are compared with the reference engine itself at full size, not only with each other.
Run from the repo root:  python tests/golden/make_tsdf_fullsize_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdf_ref as R  # noqa: E402
from tests import synth  # noqa: E402
from tests.digests import frame_digest, free_view_digest  # noqa: E402

W, H, N, STEP = 640, 480, 3, 1.0
VOXEL, MU, VFMIN, VFMAX = 0.005, 0.02, 0.2, 10.0
assert R.available(), "oracle/_ref/itm_ref missing: bash oracle/ref_build.sh"
seq = synth.make_sequence(W, H, N, step_deg=STEP)
free = [(N - 1, seq["c2w"][0])]
ref = R.run(seq, VOXEL, MU, VFMIN, VFMAX, free_views=free)
get = lambda k, f: ref[(k, f)]
out = {"W": W, "H": H, "n_frames": N, "step_deg": STEP, "voxel": VOXEL, "mu": MU, "vf_min": VFMIN, "vf_max": VFMAX}
for f in range(N):
    for k, v in frame_digest(get, f, W, H).items():
        out["%s@%d" % (k, f)] = v
    out["M@%d" % f] = get("M", f).reshape(-1)
    out["invM@%d" % f] = get("invM", f).reshape(-1)
tag = (N - 1) * 1000
for k, v in free_view_digest(get, tag, W, H).items():
    out["%s@%d" % (k, tag)] = v
trk = R.run(seq, VOXEL, MU, VFMIN, VFMAX, track=True)
out["trk_M"] = np.stack([trk[("M", f)].reshape(-1) for f in range(N)])
out["trk_invM"] = np.stack([trk[("invM", f)].reshape(-1) for f in range(N)])
out["trk_score"] = np.stack([trk[("trk_score", f)] for f in range(N)])
out["trk_counts"] = np.stack([trk[("counts", f)] for f in range(N)])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refdigest_tsdf_640x480_v5mm.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
