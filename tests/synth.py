"""Procedural RGB-D sequences (data only): an axis-aligned room with spheres, analytic depth,
view-consistent procedural colour, smooth orbit of GT poses."""
import numpy as np

from tests.scenes import look_at_c2w


def _texture(p):
    """view-consistent colour as a function of the 3D point, in [0,1]"""
    r = 0.5 + 0.5 * np.sin(3.1 * p[..., 0] + 1.7 * p[..., 1])
    g = 0.5 + 0.5 * np.sin(2.3 * p[..., 1] - 2.9 * p[..., 2] + 1.0)
    b = 0.5 + 0.5 * np.sin(4.1 * p[..., 2] + 0.7 * p[..., 0] - 0.5)
    checker = ((np.floor(p[..., 0] * 2) + np.floor(p[..., 1] * 2) + np.floor(p[..., 2] * 2)) % 2) * 0.25
    return np.clip(np.stack([r, g, b], -1) * 0.75 + checker[..., None], 0, 1)


def render_rgbd(c2w, fx, fy, cx, cy, W, H, room=(3.0, 1.5, 2.5), spheres=((0.4, 0.2, 0.3, 0.45), (-0.8, 0.5, -0.4, 0.35))):
    """returns rgb uint8 [H,W,3], depth uint16 mm [H,W] (0 = invalid)"""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d_cam = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
    R, o = c2w[:3, :3].astype(np.float64), c2w[:3, 3].astype(np.float64)
    d = d_cam @ R.T  # world direction with unit camera-z component
    t_best = np.full((H, W), np.inf)
    half = np.asarray(room, np.float64)
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (sgn * half[ax] - o[ax]) / d[..., ax]
            p = o + t[..., None] * d
            ok = (t > 1e-6)
            for a2 in range(3):
                if a2 != ax:
                    ok &= np.abs(p[..., a2]) <= half[a2] + 1e-9
            t_best = np.where(ok & (t < t_best), t, t_best)
    for sx, sy, sz, sr in spheres:
        c = np.array([sx, sy, sz])
        oc = o - c
        a = (d * d).sum(-1)
        b = 2 * (d * oc).sum(-1)
        cc = (oc * oc).sum() - sr * sr
        disc = b * b - 4 * a * cc
        with np.errstate(invalid="ignore"):
            t = (-b - np.sqrt(disc)) / (2 * a)
        ok = (disc > 0) & (t > 1e-6)
        t_best = np.where(ok & (t < t_best), t, t_best)
    hit = np.isfinite(t_best)
    tb = np.where(hit, t_best, 0.0)
    p = o + tb[..., None] * d
    depth_m = tb  # camera-z component of the ray direction is 1
    depth = np.where(hit, np.clip(np.round(depth_m * 1000.0), 0, 65535), 0).astype(np.uint16)
    rgb = (np.where(hit[..., None], _texture(p), 0.0) * 255.0 + 0.5).astype(np.uint8)
    return rgb, depth


def orbit_poses(n, radius=0.6, height=0.0, step_deg=0.4, target=(0.3, 0.1, 0.9)):
    poses = []
    for k in range(n):
        a = np.deg2rad(step_deg * k)
        eye = np.array([radius * np.sin(a) - 1.2, height + 0.05 * np.sin(3 * a), -radius * np.cos(a) - 0.8])
        poses.append(look_at_c2w(eye, target))
    return poses


def make_sequence(W, H, n, fx=None, step_deg=0.4):
    fx = fx or 0.5 * W
    fy, cx, cy = fx, (W - 1) / 2.0, (H - 1) / 2.0
    poses = orbit_poses(n, step_deg=step_deg)
    frames = [render_rgbd(p, fx, fy, cx, cy, W, H) for p in poses]
    rgbs = np.stack([f[0] for f in frames])
    depths = np.stack([f[1] for f in frames])
    return dict(W=W, H=H, fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), rgb=rgbs, depth=depths,
                c2w=np.stack(poses).astype(np.float32))
