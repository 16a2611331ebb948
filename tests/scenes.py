"""Seeded synthetic inputs shared by the oracle tests and the GPU parity tests (data only)."""
import numpy as np


def look_at_c2w(eye, target, up=(0.0, -1.0, 0.0)):
    """OpenCV-style camera (x right, y down, z forward) -> 4x4 camera-to-world."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, -np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, d, f, eye
    return c2w.astype(np.float32)


def intrinsics(W, H):
    fx = fy = 0.5 * W  # 90 deg HFOV as in SURVEY 8(d)
    return np.array([[fx, 0, (W - 1) / 2.0], [0, fy, (H - 1) / 2.0], [0, 0, 1]], np.float32)


def random_gaussians(N, seed=1234, depth_range=(1.0, 4.0), spread=2.5, scale_range=(0.003, 0.05), sh_k=16):
    """Gaussians scattered in front of a camera at the origin looking down +z."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(*depth_range, N)
    x = rng.uniform(-spread, spread, N) * z / depth_range[1]
    y = rng.uniform(-spread * 0.75, spread * 0.75, N) * z / depth_range[1]
    means = np.stack([x, y, z], 1).astype(np.float32)
    quats = rng.normal(size=(N, 4)).astype(np.float32)
    ls = rng.uniform(np.log(scale_range[0]), np.log(scale_range[1]), (N, 3))
    ls[:, 2] += np.log(0.3)
    log_scales = ls.astype(np.float32)
    opac_logit = rng.normal(0.0, 1.5, (N, 1)).astype(np.float32)
    sh = np.zeros((N, sh_k, 3), np.float32)
    sh[:, 0] = (rng.uniform(0.05, 0.95, (N, 3)) - 0.5) / 0.28209479177387814
    sh[:, 1:] = rng.normal(0, 0.05, (N, sh_k - 1, 3))
    return dict(means=means, quats=quats, log_scales=log_scales, opac_logit=opac_logit, sh=sh.astype(np.float32))


def default_camera(W, H, seed=0):
    rng = np.random.default_rng(seed)
    eye = rng.normal(0, 0.05, 3)
    c2w = look_at_c2w(eye, [0.05, -0.03, 3.0])
    return c2w, intrinsics(W, H)


def pose_inv(c2w):
    R, t = c2w[:3, :3], c2w[:3, 3]
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = R.T
    w2c[:3, 3] = -R.T @ t
    return w2c


def ulp_jitter(a, rng, ulps=1.0):
    """a float32 array with every element moved by a random relative amount of at most `ulps` * 2^-23 (integers untouched)"""
    a = np.asarray(a)
    if a.dtype.kind != "f":
        return a
    return (a.astype(np.float64) * (1.0 + rng.uniform(-1.0, 1.0, a.shape) * ulps * 2.0 ** -23)).astype(np.float32)


def condition_budget(fn, inputs, base, trials=3, seed=0, factor=32.0, floor=2e-5):
    """Per-ROW error budget of a float32 adjoint `fn(*inputs) -> tuple of [N, ...] arrays` from its own sensitivity: the outputs
    are re-evaluated `trials` times with every float input jittered by <= 1 ulp; a row's budget is
        factor * max_trials max_columns |fn(jittered) - base|  +  floor * max_columns |base|
    -- an ill-conditioned row (the projection adjoint inverts the 2x2 conic; gradients cancel) gets exactly the slack its own
    amplification of input rounding justifies, a well-conditioned one gets a few ulps.  -> list of [N] budgets, one per output."""
    rng = np.random.default_rng(seed)
    worst = [np.zeros(b.shape[0]) for b in base]
    for _ in range(trials):
        out = fn(*[ulp_jitter(a, rng) for a in inputs])
        for k, (o, b) in enumerate(zip(out, base)):
            d = np.abs(o.astype(np.float64) - b.astype(np.float64)).reshape(b.shape[0], -1).max(1)
            worst[k] = np.maximum(worst[k], d)
    return [factor * w + floor * np.abs(b).reshape(b.shape[0], -1).max(1) for w, b in zip(worst, base)]


def radius_is_borderline(conics, rel=2e-4):
    """[N] bool: the projection's radius = ceil(3 sqrt(lambda_max)) (fully_fused_projection_fwd.cu:165-167) is decided within
    `rel` of an integer -- the only Gaussians whose radius (or cull decision) float rounding may move.  lambda_max from the conic
    (= inverse of the blurred 2-D covariance), in float64."""
    a, b, c = (conics[:, k].astype(np.float64) for k in range(3))
    det_c = a * c - b * b
    with np.errstate(all="ignore"):
        ca, cb, cc = c / det_c, -b / det_c, a / det_c      # covariance
        det = ca * cc - cb * cb
        mid = 0.5 * (ca + cc)
        v = 3.0 * np.sqrt(mid + np.sqrt(np.maximum(0.01, mid * mid - det)))
    return ~np.isfinite(v) | (np.abs(v - np.round(v)) < rel * np.maximum(v, 1.0))


def bwd_class_lists(means2d, radii, tile_size, tw, th, bands=16):
    """The strip backward's work lists as the binning writes them (splat_bin.hpp: bwd_key): per class k (smallest 4 << k >= radius,
    the last class takes the rest) the visible Gaussians ordered by (band of the box's first tile row, id)."""
    r = np.asarray(radii)
    m = np.asarray(means2d, np.float32)
    ts = np.float32(tile_size)
    tr, tx, ty = r.astype(np.float32) / ts, m[:, 0] / ts, m[:, 1] / ts
    x0 = np.clip(np.floor(tx - tr), 0, tw); x1 = np.clip(np.ceil(tx + tr), 0, tw)
    y0 = np.clip(np.floor(ty - tr), 0, th); y1 = np.clip(np.ceil(ty + tr), 0, th)
    n_tiles = ((y1 - y0) * (x1 - x0)).astype(np.int64)
    band = np.where(n_tiles > 0, np.minimum(y0.astype(np.int64) * bands // th, bands - 1), 0)
    cls = np.where(r > 0, np.searchsorted(np.array([4, 8, 16, 32]), r, side="left"), -1)
    out = []
    for k in range(5):
        ids = np.nonzero(cls == k)[0]
        out.append(ids[np.argsort(band[ids], kind="stable")])
    return out
