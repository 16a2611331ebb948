"""The REFERENCE's own operator wrapper running on this repository's kernels.

oracle/_ref/_ref_wapper*.so (oracle/ref_wapper_build.py) is /root/reference/gsplat/gsplat_wapper.{hpp,cpp} -- the autograd
Functions and free functions raw_gs_model.cpp calls -- compiled from the reference's sources and linked against
gps_slam_amd/host/hip_bindings.cpp + libgpsslam_hip.so.  Here its SphericalHarmonicsNew / FullyFusedProjection /
isectTilesNoDepth / RasterizeToPixelsGes_NewParallel chain (the reference's save_for_backward lists, argument order and
returned gradient lists) is differentiated next to the repository's own mirror of that wrapper (gps_slam_amd._host): the
drop-in claim of SURVEY 8(b) at the launcher level, executed.  Also: RasterizeToPixelsGes (exact tile-parallel adjoint,
gsplat_wapper.hpp:355-487) against the oracle, RasterizeToPixels (raw) and FusedSSIMMap against the mirror.
"""
import glob
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _modules():
    import gps_slam_amd._lib as L
    L.load_library()
    import gps_slam_amd._host as h
    cands = glob.glob(os.path.join(ROOT, "oracle", "_ref", "_ref_wapper*.so"))
    if not cands:
        pytest.skip("oracle/_ref/_ref_wapper*.so not built (needs /root/reference at build time)")
    spec = importlib.util.spec_from_file_location("_ref_wapper", cands[0])
    r = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(r)
    return h, r


def _scene(N, W, H, seed):
    g = scenes.random_gaussians(N, seed=seed, scale_range=(0.004, 0.03))
    c2w, K = scenes.default_camera(W, H, seed=seed)
    gen = torch.Generator().manual_seed(seed)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    ref[ref < 0.4] = 0.0
    tensors = [T(g["means"]), T(g["log_scales"]), T(g["quats"]), T(g["sh"][:, 0].copy()), T(g["sh"][:, 1:].copy()),
               T(g["opac_logit"])]
    return tensors, c2w, K, base, ref


def _chain(mod, tensors, c2w, K, base, ref, W, H, exact_adjoint=False):
    """gesForward's operator sequence (raw_gs_model.cpp:207-316) written against a wrapper module's surface"""
    from gps_slam_amd.gs_model import pose_inv
    vm = pose_inv(torch.as_tensor(np.asarray(c2w, np.float32))).to(DEV)[None]
    Kt = T(np.asarray(K, np.float32))[None]
    tw, th = (W + 15) // 16, (H + 15) // 16
    leaves = [t.clone().requires_grad_(True) for t in tensors]
    m, s, q, d, r, o = leaves
    radii, m2, depths, conics = mod.FullyFusedProjection(m, q, torch.exp(s), vm, Kt, W, H, 0.3, 0.01, 1e10, 0.0)[:4]
    radii = torch.clamp_max(radii, 100)
    shs = torch.cat([d[:, None, :], r], 1)
    dirs = m - T(np.asarray(c2w, np.float32)[:3, 3])[None]
    cols = torch.clamp_min(mod.SphericalHarmonicsNew(3, dirs[None], shs[None], radii > 0) + 0.5, 0.0)
    cols = torch.cat([cols, depths.unsqueeze(-1)], 2)
    refc = torch.where(ref < 0.01, torch.full_like(ref, 1000.0), ref)[None]
    tpg, ids, flat, ggs, gst = mod.isectTilesNoDepth(m2, radii, depths, 16, tw, th)
    off = mod.isectOffsetEncodeNoDepth(ids, 1, tw, th)
    opac = torch.sigmoid(o)
    if exact_adjoint:
        rc, ws = mod.RasterizeToPixelsGes(m2, conics, cols, opac, refc, base, W, H, 16, off, flat, 0.1)
    else:
        rc, ws = mod.RasterizeToPixelsGes_NewParallel(m2, conics, cols, opac, radii, refc, base, W, H, 16, off, flat, ggs, gst, 0.1)
    (rc[..., :3].sum() * 0.7 + rc[..., 3].sum() * 0.1 + 0.5 * ws.sum()).backward()
    state = dict(radii=radii, m2=m2, conics=conics, cols=cols, opac=opac, refc=refc, flat=flat, ggs=ggs, gst=gst, off=off, ids=ids,
                 tpg=tpg)
    return rc.detach(), ws.detach(), [l.grad for l in leaves], state


@pytest.mark.parametrize("N,W,H", [(8000, 160, 112), (100000, 640, 480)])
def test_reference_wrapper_chain_equals_repository_wrapper(N, W, H):
    h, r = _modules()
    tensors, c2w, K, base, ref = _scene(N, W, H, seed=3)
    rc_r, ws_r, g_r, st_r = _chain(r, tensors, c2w, K, base, ref, W, H)
    rc_h, ws_h, g_h, st_h = _chain(h, tensors, c2w, K, base, ref, W, H)
    assert torch.equal(rc_r, rc_h) and torch.equal(ws_r, ws_h)
    for k in ("radii", "flat", "ggs", "gst", "off", "ids", "tpg"):
        assert torch.equal(st_r[k], st_h[k]), k
    assert ws_r.max().item() > 1.0
    for a, b in zip(g_r, g_h):
        assert a is not None and torch.isfinite(a).all()
        # same kernels underneath; the rasterizer backward completes shared Gaussians with float atomics
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())


def test_reference_exact_adjoint_function_matches_oracle_and_mirror():
    """RasterizeToPixelsGes::apply of the reference (exact tile-parallel adjoint) on gps_raster_ges_bwd_exact."""
    from oracle import splat_ref as orc
    h, r = _modules()
    N, W, H = 20000, 320, 240
    tensors, c2w, K, base, ref = _scene(N, W, H, seed=5)
    rc_r, ws_r, g_r, st = _chain(r, tensors, c2w, K, base, ref, W, H, exact_adjoint=True)
    rc_h, ws_h, g_h, _ = _chain(h, tensors, c2w, K, base, ref, W, H, exact_adjoint=True)
    assert torch.equal(rc_r, rc_h) and torch.equal(ws_r, ws_h)
    for a, b in zip(g_r, g_h):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())
    # the kernel itself against the oracle's restatement of rasterize_to_pixels_bwd_ges.cu:164-291 on the same state
    from gps_slam_amd import gsplat_ops as ops
    import ctypes as C
    from gps_slam_amd._lib import check, lib
    m2, conics, cols, opac, refc = (st[k].detach().contiguous() for k in ("m2", "conics", "cols", "opac", "refc"))
    v_rc = torch.zeros((1, H, W, 4), device=DEV)
    v_rc[..., :3] = 0.7
    v_rc[..., 3] = 0.1
    v_ra = torch.full((1, H, W, 1), 0.5, device=DEV)
    counts = torch.tensor([st["flat"].numel(), 0, 0, 0], dtype=torch.int64, device=DEV)
    out = [torch.empty_like(m2), torch.empty_like(conics), torch.empty_like(cols), torch.empty_like(opac)]
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.gps_raster_ges_bwd_exact(N, p(m2), p(conics), p(cols), p(opac), p(refc), W, H, 16, p(st["off"]), p(st["flat"]),
                                       p(counts), 0.1, p(v_rc), p(v_ra), p(out[0]), p(out[1]), p(out[2]), p(out[3]),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gps_raster_ges_bwd_exact")
    e = orc.raster_ges_bwd_exact(N_(m2)[0], N_(conics)[0], N_(cols)[0], N_(opac)[:, 0], N_(refc)[0, ..., 0], W, H, 16,
                                 N_(st["off"])[0], N_(st["flat"]), 0.1, N_(v_rc)[0], N_(v_ra)[0, ..., 0])
    for got, exp, name in zip(out, e, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        got = N_(got).reshape(exp.shape)
        bad = np.abs(got - exp) > (2e-3 * np.abs(exp) + 5e-4 * np.abs(exp).max())
        assert bad.mean() < 1e-4, (name, bad.mean())


def test_reference_raw_rasterizer_and_ssim_functions_equal_mirror():
    h, r = _modules()
    N, W, H = 8000, 160, 112
    tensors, c2w, K, base, ref = _scene(N, W, H, seed=7)
    _, _, _, st = _chain(h, tensors, c2w, K, base, ref, W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    res = []
    for mod in (r, h):
        m2 = st["m2"].detach().clone().requires_grad_(True)
        cols = st["cols"].detach().clone().requires_grad_(True)
        depths = st["cols"].detach()[..., 3].contiguous()
        tpg, ids, flat = mod.isectTiles(m2.detach(), st["radii"], depths, 16, tw, th)
        off = mod.isectOffsetEncode(ids, 1, tw, th)
        bg = torch.tensor([[0.1, 0.2, 0.3, 0.0]], device=DEV)
        rc, ra = mod.RasterizeToPixels(m2, st["conics"].detach(), cols, st["opac"].detach(), bg, W, H, 16, off, flat, False)
        (rc.sum() + ra.sum()).backward()
        res.append((rc.detach(), ra.detach(), m2.grad, cols.grad, flat))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][4], res[1][4])
    for a, b in ((res[0][2], res[1][2]), (res[0][3], res[1][3])):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())
    gen = torch.Generator().manual_seed(0)
    img2 = torch.rand((1, 3, 64, 96), generator=gen).to(DEV)
    outs = []
    for mod in (r, h):
        img1 = torch.rand((1, 3, 64, 96), generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
        m = mod.FusedSSIMMap(0.01 ** 2, 0.03 ** 2, img1, img2, "valid", True)
        m.mean().backward()
        outs.append((m.detach(), img1.grad))
    assert outs[0][0].shape == (1, 3, 54, 86)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
