"""Generates tests/golden/refdigest_tsdf_exhaustion_320x240_v2mm.npz: the REFERENCE's own ITMLib CPU engine (oracle/_ref/itm_ref)
driven until its voxel-block array is EXHAUSTED -- SDF_LOCAL_BLOCK_NUM = 0x40000 blocks are compile-time constants of the
reference (ITMLibDefines / ITMVoxelBlockHash.h), so the scene is chosen to fill them: 2 mm voxels (1.6 cm blocks), mu = 8 mm,
eight 320x240 frames of a 40-degree-per-frame orbit.  Frame 5 runs out of blocks (last free block id -1); frames 6 and 7 fuse
with nothing left to allocate: requests are dropped, their excess-list offsets restored, the visible list still grows
(AllocateSceneFromDepth, ITMSceneReconstructionEngine_CPU.tpp:120-230).  Digests as in make_tsdf_fullsize_golden.py, for every
frame.  Run from the repo root:  python tests/golden/make_tsdf_exhaustion_golden.py   (~1 minute)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdf_ref as R  # noqa: E402
from tests import synth  # noqa: E402
from tests.digests import frame_digest  # noqa: E402

W, H, N, STEP = 320, 240, 8, 40.0
VOXEL, MU, VFMIN, VFMAX = 0.002, 0.008, 0.2, 10.0
assert R.available(), "oracle/_ref/itm_ref missing: bash oracle/ref_build.sh"
seq = synth.make_sequence(W, H, N, step_deg=STEP)
ref = R.run(seq, VOXEL, MU, VFMIN, VFMAX)
get = lambda k, f: ref[(k, f)]
out = {"W": W, "H": H, "n_frames": N, "step_deg": STEP, "voxel": VOXEL, "mu": MU, "vf_min": VFMIN, "vf_max": VFMAX}
for f in range(N):
    for k, v in frame_digest(get, f, W, H).items():
        out["%s@%d" % (k, f)] = v
assert int(out["counts@%d" % (N - 1)][1]) == -1 and int(out["counts@4"][1]) > 0, "the scene no longer exhausts the block array"
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refdigest_tsdf_exhaustion_320x240_v2mm.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", [out["counts@%d" % f].tolist() for f in range(N)])
