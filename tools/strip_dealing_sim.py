#!/usr/bin/env python
"""How evenly does raster_ges_bwd_strip_kernel's dealing spread a launch over its waves?  Host-side simulation from the radii of
the Gaussians one optimise view sees (tools/probe/radius_hist.py, SAVE_RADII; two dumps are committed:
profiles/r05_radii_{1200x680,640x480}_1000f.npz, the model after 1,000 frames of the whole-sequence run).

Model of the kernel (splat_raster_bwd.hip): class k = radius <= 4 << k (the last class takes everything wider); a task = 64 / (4 << k)
consecutive visible Gaussians of a class; its cost = rows of its widest member (2 r) x column passes (ceil(r / 64), wide class only)
= dependent row trips of one wave; an XCD slot takes a contiguous eighth of every class list and deals its tasks, widest class
first, round-robin to its 768 waves (256 x 6 workgroups x 4 waves / 8).  `--coop`: the wide class is a workgroup's task, a quarter
of the rows per wave (GPS_STRIP_COOP4, what ships).  Prints mean / max trips per wave: the kernel is as long as its busiest wave.

python tools/strip_dealing_sim.py profiles/r05_radii_1200x680_1000f.npz"""
import sys

import numpy as np

WAVES, WGS = 768, 192


def task_costs(radii):
    rv = radii[radii > 0].astype(np.int64)
    cls = np.zeros_like(rv)
    for k, hi in enumerate((4, 8, 16, 32)):
        cls[rv > hi] = k + 1
    out = []
    for k in range(5):
        g = rv[cls == k]
        per = 64 // (4 << k)
        nt = (len(g) + per - 1) // per
        pad = np.zeros(nt * per, np.int64)
        pad[:len(g)] = g
        mx = pad.reshape(nt, per).max(1) if nt else np.zeros(0, np.int64)
        out.append((2 * mx * (np.ceil(mx / 64.0) if k == 4 else 1)).astype(np.int64))
    return out, [int((cls == k).sum()) for k in range(5)]


def simulate(lists, coop):
    per_wave = np.zeros((8, WAVES))
    for x in range(8):
        parts = {}
        for k in range(5):
            t = lists[k]
            px = (len(t) + 7) // 8
            parts[k] = t[x * px:(x + 1) * px]
        order = [3, 2, 1, 0]
        if coop:
            for i, c in enumerate(parts[4]):
                wg = i % WGS
                per_wave[x, wg * 4:(wg + 1) * 4] += np.ceil(c / 4.0)
        else:
            order = [4] + order
        seq = np.concatenate([parts[k] for k in order])
        for f, c in enumerate(seq):
            per_wave[x, f % WAVES] += c
    return per_wave


def main():
    for path in sys.argv[1:] or ["profiles/r05_radii_1200x680_1000f.npz", "profiles/r05_radii_640x480_1000f.npz"]:
        radii = np.load(path)["radii"]
        lists, counts = task_costs(radii)
        print("%s: %d visible, per class %s; row trips per class %s" % (path, sum(counts), counts, [int(t.sum()) for t in lists]))
        for coop in (False, True):
            pw = simulate(lists, coop)
            print("   %-46s mean %4.0f  max %4.0f  (x %.2f)  p99 %4.0f" % ("wide class shared by a workgroup's four waves" if coop else "every task one wave's (rounds 3-4)",
                                                                       pw.mean(), pw.max(), pw.max() / pw.mean(), np.quantile(pw, 0.99)))


if __name__ == "__main__":
    main()
