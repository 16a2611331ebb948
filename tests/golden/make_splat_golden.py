"""Generates tests/golden/splat_96x64_n300.npz: a small fixed scene pushed through the splat CPU oracle
(oracle/splat_oracle.c) -- inputs and every intermediate / output of the ges chain.

The reference owns no fixtures for this path and is CUDA-only (cannot run here), so these vectors are produced by the build's
own restatement AFTER tests/test_oracle_splat.py has cross-checked it against the dense float64 formulation ("parity unpinned"
by the reference; the fixture freezes the restatement so that later edits of the oracle or the kernels cannot drift silently).
Run from the repo root:  python tests/golden/make_splat_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import splat_ref as orc  # noqa: E402
from tests import scenes  # noqa: E402

W, H, TS, N = 96, 64, 16, 300
tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
g = scenes.random_gaussians(N, seed=21, scale_range=(0.01, 0.1))
c2w, K = scenes.default_camera(W, H, seed=21)
vm = scenes.pose_inv(c2w)
scales = np.exp(g["log_scales"]).astype(np.float32)
radii, m2, depths, conics = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
radii = np.minimum(radii, 100).astype(np.int32)
dirs = (g["means"] - c2w[:3, 3][None]).astype(np.float32)
rgb = np.maximum(orc.sh_fwd(3, dirs, g["sh"], radii > 0) + 0.5, 0).astype(np.float32)
colors = np.concatenate([rgb, depths[:, None]], 1).astype(np.float32)
opac = (1.0 / (1.0 + np.exp(-g["opac_logit"].reshape(-1)))).astype(np.float32)
rng = np.random.default_rng(4)
ref_depth = rng.uniform(1.0, 5.0, (H, W)).astype(np.float32)
ref_depth[rng.uniform(size=(H, W)) < 0.15] = 1000.0
tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
rc, ra, last = orc.raster_ges_fwd(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, 0.1)
v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
v_ra = rng.normal(size=(H, W)).astype(np.float32)
v_m2, v_con, v_col, v_op = orc.raster_ges_bwd_gs(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, 0.1, v_rc, v_ra)
# rounding scale (sum |terms|: rel_band -1; the same weighted by sum |terms of sigma|: rel_band -2) and borderline-pair budgets
# (rel_band = 1e-5) of both rasterizer stages: what a consumer of this fixture may tolerate per element instead of a generic
# outlier budget
fb = {}
for tag, band in (("scale", -1.0), ("sig", -2.0), ("flip", 1e-5)):
    fb["fwd_" + tag], _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, 0.1, rel_band=band)
    fb["bwd_" + tag], _, _ = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, 0.1, v_rc, v_ra,
                                                            rel_band=band)
v_means, v_quats, v_scales = orc.proj_bwd(g["means"], g["quats"], scales, vm, K, W, H, radii, conics, v_m2,
                                          np.ascontiguousarray(v_col[:, 3]), v_con)
v_coeffs, v_dirs = orc.sh_bwd(3, dirs, g["sh"], radii > 0, np.ascontiguousarray(v_col[:, :3]))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "splat_96x64_n300.npz")
np.savez_compressed(out, W=W, H=H, TS=TS, means=g["means"], quats=g["quats"], log_scales=g["log_scales"], sh=g["sh"],
                    opac_logit=g["opac_logit"], c2w=c2w, K=K, viewmat=vm, ref_depth=ref_depth, v_rc=v_rc, v_ra=v_ra,
                    radii=radii, means2d=m2, depths=depths, conics=conics, colors=colors, opac=opac, tiles_per_gauss=tpg,
                    isect_ids=ids, flatten_ids=flat, group_gs_ids=ggs, group_starts=gst, offsets=offs, render_colors=rc,
                    weight_sum=ra, v_means2d=v_m2, v_conics=v_con, v_colors=v_col, v_opacities=v_op, v_means=v_means,
                    v_quats=v_quats, v_scales=v_scales, v_coeffs=v_coeffs, **fb)
print("wrote", out, os.path.getsize(out), "bytes; n_isects", len(flat), "n_groups", len(ggs), "visible", int((radii > 0).sum()))
