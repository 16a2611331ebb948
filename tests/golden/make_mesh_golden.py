"""Generates tests/golden/mesh_64x48.npz: the triangles the REFERENCE's own CPU meshing engine (ITMMeshingEngine_CPU via
oracle/_ref/itm_ref in `mesh` mode, built by oracle/ref_build.sh from /root/reference) produces for the scene fused from
a small synthetic sequence, plus digests of the reference's SaveToDirectory files and WritePLY output for the same scene.
Only data is stored.  Run from the repo root:  python tests/golden/make_mesh_golden.py"""
import os
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdf_ref as R  # noqa: E402
from tests import synth  # noqa: E402

W, H, N, STEP = 64, 48, 3, 1.0
VOXEL, MU, VFMIN, VFMAX = 0.04, 0.16, 0.2, 10.0
assert R.available(), "oracle/_ref/itm_ref missing: bash oracle/ref_build.sh"
seq = synth.make_sequence(W, H, N, step_deg=STEP)
with tempfile.TemporaryDirectory() as d:
    ref = R.run(seq, VOXEL, MU, VFMIN, VFMAX, mesh=True, save_dir=d)
    ply = open(os.path.join(d, "mesh.ply")).read().split("\n")
    files = {}
    for name in ("alloc.dat", "vba.txt", "hash.dat", "excess.dat", "last.txt"):
        files[name] = zlib.crc32(open(os.path.join(d, name), "rb").read())
    vox = np.fromfile(os.path.join(d, "voxel.dat"), np.uint8)
    files["voxel.dat.count"] = int(np.frombuffer(vox[:8].tobytes(), np.uint64)[0])
    files["voxel.dat.payload7"] = zlib.crc32(np.ascontiguousarray(vox[8:].reshape(-1, 8)[:, :7]).tobytes())  # pad byte dropped
mesh = ref[("mesh", N - 1)]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh_64x48.npz")
np.savez_compressed(out, W=W, H=H, n_frames=N, step_deg=STEP, voxel=VOXEL, mu=MU, vf_min=VFMIN, vf_max=VFMAX,
                    triangles=mesh, ply_header=np.array(ply[:12]), ply_vertex_lines=np.array(ply[12:12 + 30]),
                    ply_lines=len(ply), file_names=np.array(sorted(files)), file_digests=np.array([files[k] for k in sorted(files)], np.int64))
print("wrote", out, os.path.getsize(out), "bytes;", mesh.shape[0], "triangles")
