#!/usr/bin/env python
"""Row trips of a raster_ges_bwd_strip_kernel launch under different ways of forming its tasks, from the radius and the
{alpha >= 1/255} pixel bounds (pack_record) of the Gaussians one optimise view sees (tools/probe/extent_dump.py; committed:
profiles/r05_extents_{640x480,1200x680}_1000f.npz = visible Gaussians in id order: radius, bounds' width and rows, image-clipped).

A task = 64 / (4 << k) consecutive members of a list; its trips = its tallest member's rows (x column passes in the 64-lane class).
Shipped: lists by radius class (lanes = radius).  Simulated: lanes by the bounds' half width, and lists additionally split into
row buckets relative to the class (rows * nb / (4 lanes)), so that a task's members are about equally tall.

python tools/strip_classes_sim.py [extents.npz ...]"""
import sys

import numpy as np


def lane_class(key):
    cls = np.zeros_like(key)
    for k, hi in enumerate((4, 8, 16, 32)):
        cls[key > hi] = k + 1
    return cls


def trips(lane_key, rows, n_buckets):
    lc = lane_class(lane_key)
    gw = 4 << lc
    rb = np.zeros_like(rows) if n_buckets <= 1 else np.minimum((rows * n_buckets) // (4 * gw), n_buckets - 1)
    total = 0
    for k in range(5):
        for b in np.unique(rb[lc == k]):
            m = (lc == k) & (rb == b)
            per = 64 // (4 << k)
            nt = (int(m.sum()) + per - 1) // per
            pad_r = np.zeros(nt * per, np.int64); pad_r[:m.sum()] = rows[m]
            pad_k = np.zeros(nt * per, np.int64); pad_k[:m.sum()] = lane_key[m]
            t = pad_r.reshape(nt, per).max(1) * (np.ceil(pad_k.reshape(nt, per).max(1) / 64.0) if k == 4 else 1)
            total += int(t.sum())
    return total


def main():
    for path in sys.argv[1:] or ["profiles/r05_extents_640x480_1000f.npz", "profiles/r05_extents_1200x680_1000f.npz"]:
        d = np.load(path)
        r, width, rows = d["radii"].astype(np.int64), d["width"].astype(np.int64), d["rows"].astype(np.int64)
        hw = np.maximum((width + 1) // 2, 1)
        base = trips(r, rows, 1)
        print("%s: %d visible; lanes x rows inside the bounds / 64 = %d trips' worth" % (path, len(r), int((rows * hw).sum() // 64)))
        print("   lists by radius class (shipped)                      %9d trips" % base)
        for name, key in (("radius", r), ("half width", hw)):
            for nb in (1, 4, 8, 16):
                if name == "radius" and nb == 1:
                    continue
                t = trips(key, rows, nb)
                print("   lanes by %-10s x %2d row buckets                  %9d trips  (%4.1f %% fewer)" % (name, nb, t, 100.0 * (1 - t / base)))


if __name__ == "__main__":
    main()
