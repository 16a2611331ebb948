"""Generates tests/golden/refpy_quat_covar.npz from the REFERENCE'S OWN PYTHON (imported from /root/reference in the build
container; it cannot travel, so its outputs are committed as data):

  * scripts/utils/general_utils.py: build_rotation(r) -- quaternion (w, x, y, z), normalised inside, -> rotation matrix; and
    build_scaling_rotation(s, r) -- L = R diag(s); the 3-D covariance the Gaussian model defines from it is L L^T (the
    reference's Python model class does exactly `L @ L.transpose(1, 2)`).  These are the reference's own statements of the first
    two sub-steps of FullyFusedProjection (gsplat/rasterizer/utils.cuh:14-96: quat_to_rotmat, quat_scale_to_covar_preci).
    Also autograd through build_rotation: dL/dq for a fixed dL/dR (utils.cuh:38-62 quat_to_rotmat_vjp).
  * general_utils.inverse_sigmoid, sh_utils.RGB2SH / SH2RGB: what RawGaussianParams::init uses for the opacity logit and the
    SH DC term (src/raw_gs_param.cpp:44-58 through gsplat_wapper's rgb2sh).

The two functions hard-code device="cuda" for their outputs; the build container has no GPU, so torch.zeros is redirected to the
CPU for the duration of the calls -- the function bodies run unchanged.

Run from the repo root:  python tests/golden/make_refpy_quat_golden.py"""
import contextlib
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/scripts/utils"


def _load(name):
    spec = importlib.util.spec_from_file_location("refpy_" + name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@contextlib.contextmanager
def zeros_on_cpu(dtype):
    orig = torch.zeros

    def zeros(*a, **k):
        k.pop("device", None)
        if k.get("dtype") is torch.float:   # (the functions ask for float32 explicitly in places: follow the requested precision)
            k["dtype"] = dtype
        k.setdefault("dtype", dtype)
        return orig(*a, **k)
    torch.zeros = zeros
    try:
        yield
    finally:
        torch.zeros = orig


def main():
    gu, sh_utils = _load("general_utils"), _load("sh_utils")
    rng = np.random.default_rng(20260930)
    N = 64
    quats = (rng.normal(size=(N, 4)) * rng.uniform(0.1, 4.0, size=(N, 1))).astype(np.float32)   # un-normalised, as the model stores them
    quats[0] = [1, 0, 0, 0]; quats[1] = [0, 1, 0, 0]; quats[2] = [0.5, 0.5, 0.5, 0.5]; quats[3] = [-2.0, 0, 0, 2.0]
    scales = np.exp(rng.uniform(np.log(0.003), np.log(0.3), size=(N, 3))).astype(np.float32)
    v_R = rng.normal(size=(N, 3, 3)).astype(np.float32)
    out = {"quats": quats, "scales": scales, "v_R": v_R}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        with zeros_on_cpu(dt):
            q = torch.tensor(quats, dtype=dt, requires_grad=True)
            s = torch.tensor(scales, dtype=dt)
            R = gu.build_rotation(q)
            L = gu.build_scaling_rotation(s, q)
            cov = L @ L.transpose(1, 2)
            (R * torch.tensor(v_R, dtype=dt)).sum().backward()
        out["R_" + tag] = R.detach().numpy()
        out["covar_" + tag] = cov.detach().numpy()
        out["v_quats_" + tag] = q.grad.numpy()
    rgb = rng.uniform(size=(N, 3)).astype(np.float32)
    x = rng.uniform(0.01, 0.99, size=N).astype(np.float32)
    x[0] = 0.5
    out.update(rgb=rgb, rgb2sh_f64=sh_utils.RGB2SH(torch.tensor(rgb, dtype=torch.float64)).numpy(),
               sh2rgb_f64=sh_utils.SH2RGB(torch.tensor(rgb, dtype=torch.float64) - 0.5).numpy(),
               sig_in=x, inverse_sigmoid_f64=gu.inverse_sigmoid(torch.tensor(x, dtype=torch.float64)).numpy())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refpy_quat_covar.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
