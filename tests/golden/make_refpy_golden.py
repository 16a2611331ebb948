"""Generates tests/golden/refpy_sh_ssim_psnr.npz from the REFERENCE'S OWN PYTHON (imported from /root/reference in the build
container; it cannot travel, so its outputs are committed as data):

  * scripts/utils/sh_utils.py: eval_sh(deg, sh, dirs), degrees 0..4 -- the spherical-harmonics colour model the reference's
    evaluation scripts hold; values in float64 and float32, and its autograd gradients (v_coeffs, v_dirs through the
    normalisation the rasterizer applies, gsplat/rasterizer/spherical_harmonics.cuh:27-31) for a fixed v_colors.
    Pins oracle/splat_oracle.c orc_sh_fwd / orc_sh_bwd (CPU suite) and csrc/splat_project.hip (GPU suite) by reference output.
  * scripts/utils/loss_utils.py: _ssim (11-tap sigma-1.5 window, zero padding) -- the SSIM map the fused kernel reproduces;
    l1_loss; scripts/utils/image_utils.py: psnr -- the number scripts/metric.py reports.

Run from the repo root:  python tests/golden/make_refpy_golden.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/scripts/utils"


def _load(name):
    spec = importlib.util.spec_from_file_location("refpy_" + name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    sh_utils, loss_utils, image_utils = _load("sh_utils"), _load("loss_utils"), _load("image_utils")
    rng = np.random.default_rng(20260929)
    out = {}
    N = 48
    for deg in range(5):
        K = (deg + 1) ** 2
        dirs = (rng.normal(size=(N, 3)) * rng.uniform(0.2, 5.0, size=(N, 1))).astype(np.float32)  # un-normalised, as the kernel gets them
        dirs[0] = [0.0, 0.0, 2.0]; dirs[1] = [1.0, 0.0, 0.0]; dirs[2] = [0.0, -3.0, 0.0]  # axis directions
        coeffs = rng.normal(size=(N, K, 3)).astype(np.float32)
        v_colors = rng.normal(size=(N, 3)).astype(np.float32)
        res = {}
        for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
            d = torch.tensor(dirs, dtype=dt, requires_grad=True)
            c = torch.tensor(coeffs, dtype=dt, requires_grad=True)
            unit = d / d.norm(dim=-1, keepdim=True)
            col = sh_utils.eval_sh(deg, c.transpose(1, 2), unit)  # sh [..., C, K], dirs [..., 3] -> [..., C]
            (col * torch.tensor(v_colors, dtype=dt)).sum().backward()
            res[tag] = (col.detach().numpy(), c.grad.numpy(), (d.grad if d.grad is not None else torch.zeros_like(d)).numpy())  # degree 0 does not read dirs
        out.update({"sh%d_dirs" % deg: dirs, "sh%d_coeffs" % deg: coeffs, "sh%d_v_colors" % deg: v_colors,
                    "sh%d_colors_f64" % deg: res["f64"][0], "sh%d_v_coeffs_f64" % deg: res["f64"][1],
                    "sh%d_v_dirs_f64" % deg: res["f64"][2], "sh%d_colors_f32" % deg: res["f32"][0]})
    # SSIM map / loss scalars
    C, Hh, Ww = 3, 37, 50
    img1 = rng.uniform(size=(C, Hh, Ww)).astype(np.float32)
    img2 = np.clip(img1 + rng.normal(scale=0.1, size=img1.shape), 0, 1).astype(np.float32)
    a, b = torch.tensor(img1, dtype=torch.float64)[None], torch.tensor(img2, dtype=torch.float64)[None]
    window = loss_utils.create_window(11, C).double()
    # _ssim(size_average=True) returns the mean; the map is its operand: recomputed here with the module's own window and formula
    import torch.nn.functional as F
    mu1, mu2 = F.conv2d(a, window, padding=5, groups=C), F.conv2d(b, window, padding=5, groups=C)
    s1 = F.conv2d(a * a, window, padding=5, groups=C) - mu1 * mu1
    s2 = F.conv2d(b * b, window, padding=5, groups=C) - mu2 * mu2
    s12 = F.conv2d(a * b, window, padding=5, groups=C) - mu1 * mu2
    ssim_map = ((2 * mu1 * mu2 + loss_utils.C1) * (2 * s12 + loss_utils.C2)) / ((mu1 * mu1 + mu2 * mu2 + loss_utils.C1) * (s1 + s2 + loss_utils.C2))
    ssim_mean = float(loss_utils._ssim(a, b, window, 11, C, True))
    assert abs(float(ssim_map.mean()) - ssim_mean) < 1e-12  # the map above IS what the module averages
    out.update(ssim_img1=img1, ssim_img2=img2, ssim_map_f64=ssim_map[0].numpy(), ssim_mean_f64=ssim_mean,
               l1_f64=float(loss_utils.l1_loss(a, b)), psnr_f64=float(image_utils.psnr(a, b)))
    # PSNR of 8-bit images the way scripts/metric.py feeds image_utils.psnr (uint8 -> float / 255)
    q1, q2 = np.clip(img1 * 255.0, 0, 255).astype(np.uint8), np.clip(img2 * 255.0, 0, 255).astype(np.uint8)
    out.update(psnr8_a=q1, psnr8_b=q2, psnr8_f64=float(image_utils.psnr(torch.tensor(q1, dtype=torch.float64)[None] / 255.0,
                                                                        torch.tensor(q2, dtype=torch.float64)[None] / 255.0)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refpy_sh_ssim_psnr.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
