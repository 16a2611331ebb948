"""Generates tests/golden/tsdf_*.npz with the REFERENCE's own ITMLib CPU engine
(oracle/_ref/itm_ref, built by oracle/ref_build.sh from /root/reference).

Run in the build container only (needs /root/reference):  python tests/golden/make_tsdf_golden.py
The fixtures are data: inputs (rgb, depth, GT poses, free-view poses, parameters) and the engine's
outputs (poses after Coerce, hash table, visible lists, voxel payloads, raycast / ICP / colour images).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdf_ref as R  # noqa: E402
from tests import synth  # noqa: E402

CASES = {
    # name: (W, H, frames, voxel, mu, vf_min, vf_max, orbit step in degrees, free views [(after frame, pose of frame)])
    "tsdf_64x48_v20mm": (64, 48, 4, 0.02, 0.08, 0.2, 10.0, 0.4, [(3, 1), (3, 3)]),
    "tsdf_32x24_v5mm_fast_orbit": (32, 24, 5, 0.005, 0.02, 0.2, 10.0, 2.5, [(4, 0)]),
}


def main():
    assert R.available(), "build oracle/_ref first (bash oracle/ref_build.sh)"
    for name, (W, H, n, voxel, mu, vmin, vmax, step, fvs) in CASES.items():
        seq = synth.make_sequence(W, H, n, step_deg=step)
        free = [(f, seq["c2w"][k]) for f, k in fvs]
        out = R.run(seq, voxel, mu, vmin, vmax, free_views=free)
        save = dict(W=W, H=H, fx=seq["fx"], fy=seq["fy"], cx=seq["cx"], cy=seq["cy"], voxel=voxel, mu=mu, vf_min=vmin,
                    vf_max=vmax, rgb=seq["rgb"], depth=seq["depth"], c2w=seq["c2w"],
                    free_frames=np.array([f for f, _ in free], np.int32), free_c2w=np.stack([c for _, c in free]))
        for (k, fr), a in out.items():
            if k == "vba":
                # keep every 8th allocated block (hash-index order) + the CRC of all of them ("vba_crc")
                a = a[::8, :, :7]  # [..., :7] drops the struct padding byte
            save["%s@%d" % (k, fr)] = a
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **save)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
