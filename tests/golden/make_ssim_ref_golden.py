"""Writes tests/golden/ssim_ref_gfx950.npz: outputs of the REFERENCE's fused-SSIM kernels (gsplat/rasterizer/ssim.cu compiled for
gfx950 by oracle/ref_ssim_build.py) on two small seeded image pairs.  Run ON the GPU box (the kernel needs a device):

    gpurun -- 'python tests/golden/make_ssim_ref_golden.py gpurun_out/ssim_ref_gfx950.npz'   then copy the file into tests/golden/

Stored: the inputs (img1, img2, dL_dmap), the four forward maps and the backward's dL_dimg1 (float32, as the kernel wrote them).
tests/test_oracle_splat.py checks the C restatement against them on CPU; tests/test_reference_ssim_gpu.py regenerates them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_ssim_build  # noqa: E402


def main(out):
    ref = ref_ssim_build.load()
    C1, C2 = float(np.float32(0.01 * 0.01)), float(np.float32(0.03 * 0.03))
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")
    N_ = lambda t: t.detach().cpu().numpy()
    z = {}
    for tag, (B, CH, H, W), seed in (("a", (1, 3, 24, 40), 11), ("b", (2, 1, 33, 31), 12)):  # 'b': ragged, wider than one 16x16 block
        rng = np.random.default_rng(seed)
        img2 = rng.uniform(0, 1, (B, CH, H, W)).astype(np.float32)
        img1 = np.clip(img2 + rng.normal(0, 0.15, img2.shape), 0, 1).astype(np.float32)
        dL = rng.normal(size=img1.shape).astype(np.float32)
        m, d1, d2, d3 = ref.fusedssim(C1, C2, T(img1), T(img2), True)
        g = ref.fusedssim_backward(C1, C2, T(img1), T(img2), T(dL), d1, d2, d3)
        for name, v in (("img1", img1), ("img2", img2), ("dL", dL), ("map", N_(m)), ("dm_dmu1", N_(d1)), ("dm_dsigma1_sq", N_(d2)),
                        ("dm_dsigma12", N_(d3)), ("grad", N_(g))):
            z[tag + "_" + name] = v
    np.savez_compressed(out, **z)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ssim_ref_gfx950.npz"))
