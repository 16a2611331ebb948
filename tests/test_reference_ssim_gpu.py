"""The REFERENCE's own fused-SSIM kernels on the MI355X, next to the oracle and csrc/splat_ssim.hip.

oracle/_ref/_ref_ssim*.so (oracle/ref_ssim_build.py) is /root/reference/gsplat/rasterizer/ssim.cu compiled for gfx950 from the
reference's source (hipify of a temporary copy: host API names only) -- the one splat translation unit that builds without
headers the image lacks.  What is compared here is therefore reference output, not a restatement of it:
    reference kernel == oracle/splat_oracle.c orc_ssim_fwd / orc_ssim_bwd     (pins the SSIM oracle)
    reference kernel == gps_ssim_fwd / gps_ssim_bwd                            (pins the HIP kernels directly)
at toy, ragged and BASELINE image sizes.  Tolerances are float32 rounding of a 121-term window sum (the three evaluate the same
sums in different orders / with different fma contraction)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C1, C2 = float(np.float32(0.01 * 0.01)), float(np.float32(0.03 * 0.03))


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _ref():
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "_ref_ssim*.so")):
        pytest.skip("oracle/_ref/_ref_ssim*.so not built (needs /root/reference at build time)")
    from oracle import ref_ssim_build
    return ref_ssim_build.load()


def _images(B, CH, H, W, seed):
    rng = np.random.default_rng(seed)
    img2 = rng.uniform(0, 1, (B, CH, H, W)).astype(np.float32)
    img1 = np.clip(img2 + rng.normal(0, 0.15, img2.shape), 0, 1).astype(np.float32)
    dL = rng.normal(size=img1.shape).astype(np.float32)
    return img1, img2, dL


def _close(got, ref, name, rtol, atol_rel):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * max(1.0, float(np.abs(ref).max())), err_msg=name)


@pytest.mark.parametrize("B,CH,H,W", [(1, 3, 37, 50), (2, 3, 64, 96), (1, 1, 33, 31), (1, 3, 480, 640)])
def test_reference_ssim_kernel_equals_oracle_and_hip(B, CH, H, W):
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    ref = _ref()
    img1, img2, dL = _images(B, CH, H, W, seed=B * 1000 + H)
    r_m, r1, r2, r3 = [N_(t) for t in ref.fusedssim(C1, C2, T(img1), T(img2), True)]
    assert 0.05 < r_m.mean() < 0.95
    # forward: oracle and HIP against the reference kernel's four maps
    o = orc.ssim_fwd(img1, img2, C1, C2)
    h = ops.fusedssim(C1, C2, T(img1), T(img2), train=True)
    for k, name in enumerate(("ssim_map", "dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12")):
        want = (r_m, r1, r2, r3)[k]
        _close(o[k], want, "oracle " + name, 2e-4, 2e-5)
        _close(N_(h[k]), want, "hip " + name, 2e-4, 2e-5)
    # backward on the REFERENCE's partial maps (what FusedSSIMMap::backward saves): all three get identical inputs
    r_g = N_(ref.fusedssim_backward(C1, C2, T(img1), T(img2), T(dL), T(r1), T(r2), T(r3)))
    o_g = orc.ssim_bwd(img1, img2, dL, r1, r2, r3)
    h_g = N_(ops.fusedssim_backward(C1, C2, T(img1), T(img2), T(dL), T(r1), T(r2), T(r3)))
    _close(o_g, r_g, "oracle dL_dimg1", 1e-3, 1e-4)
    _close(h_g, r_g, "hip dL_dimg1", 1e-3, 1e-4)
    # not training: the same map
    m2 = ref.fusedssim(C1, C2, T(img1), T(img2), False)[0]
    assert np.array_equal(N_(m2), r_m)


def test_reference_ssim_golden_fixture_is_what_the_reference_kernel_gives():
    """tests/golden/ssim_ref_gfx950.npz (the CPU suite's pin of the oracle) was written by tests/golden/make_ssim_ref_golden.py from
    this kernel: regenerate and compare bit for bit."""
    ref = _ref()
    path = os.path.join(ROOT, "tests", "golden", "ssim_ref_gfx950.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated yet")
    z = np.load(path)
    for tag in ("a", "b"):
        img1, img2, dL = z[tag + "_img1"], z[tag + "_img2"], z[tag + "_dL"]
        outs = ref.fusedssim(C1, C2, T(img1), T(img2), True)
        for t, name in zip(outs, ("map", "dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12")):
            assert np.array_equal(N_(t), z[tag + "_" + name]), name
        g = ref.fusedssim_backward(C1, C2, T(img1), T(img2), T(dL), *outs[1:])
        assert np.array_equal(N_(g), z[tag + "_grad"])
