"""diagnostic: where does the op-level ges forward exceed rounding + flip budget (tests/test_splat_gpu.py scene)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_splat_gpu import _raster_state, T, N_
from gps_slam_amd import gsplat_ops as ops
from oracle import splat_ref as orc
N, W, H = 100000, 640, 480
TS, delta = 16, 0.1
tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
radii, m2, depths, conics, colors, opac, ref_depth = _raster_state(N, W, H, seed=N)
tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
e_rc, e_ra, e_last = orc.raster_ges_fwd(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta)
isect = ops.isect_tiles_no_depth(T(m2)[None], T(radii)[None], TS, tw, th)
rc, ra, last = ops.rasterize_to_pixels_fwd_ges(T(m2)[None], T(conics)[None], T(colors)[None], T(opac)[:, None], T(ref_depth)[None, ..., None], W, H, TS, isect, delta, want_last_ids=True)
scale_f, _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=-1.0)
sig_f, _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=-2.0)
for band in (1e-5,):
    flip_f, n_pairs, n_pix = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=band)
    got = np.concatenate([N_(rc)[0], N_(ra)[0]], -1)
    exp = np.concatenate([e_rc, e_ra[..., None]], -1)
    d = np.abs(got - exp)
    ex = d - (2e-5 * scale_f + 1e-7 + 1.001 * flip_f)
    bad = np.argwhere(ex > 0)
    print("band", band, "pairs", n_pairs, "pix", n_pix, "bad", bad.shape[0])
    for y, x, c in bad[:12]:
        print("  px", x, y, "ch", c, "d", d[y, x, c], "scale", scale_f[y, x, c], "flip", flip_f[y, x, c], "got", got[y, x, c], "exp", exp[y, x, c],
              "rel", d[y, x, c] / (scale_f[y, x, c] + 1e-30), "sig", sig_f[y, x, c], "d/sig", d[y, x, c] / (sig_f[y, x, c] + 1e-30))
print("max rel over all", (d / (scale_f + 1e-12)).max(), "opac range", opac.min(), opac.max(), "conic range", conics.min(), conics.max())

flip_f, _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=1e-5)
for eps in (1.2e-7, 2.4e-7, 4.8e-7):
    for rel in (2e-6, 5e-6, 2e-5):
        ex = d - (rel * scale_f + eps * sig_f + 1e-7 + 1.001 * flip_f)
        print("eps", eps, "rel", rel, "bad", int((ex > 0).sum()))
