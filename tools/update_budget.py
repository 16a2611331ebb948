"""Where one keyframe update spends its time on the map stream, from a rocprofv3 --kernel-trace rocpd database of bench.py.

An update = the map stream's kernels from one `upload_views_kernel` batch (the keyframe views going up) to the last kernel before
the next.  Split into: `pre` (up to the first backward kernel's iteration = raycasts, mask render, addGaussians),
`iterations` (first ... last preprocess_bwd_kernel), `post` (prune, keyframe bookkeeping).  For each part: wall time, busy time
(union of the update's kernels), the largest gaps and what stands on either side of them.
usage: update_budget.py <db> [schedule: 0 = first timed schedule, 1 = second] [windows per schedule, default 2]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    sched = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nw = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    marks = [s for n, s, e, st in rows if "spin_kernel" in n]
    lo, hi = marks[2 * nw * sched], marks[2 * nw * sched + 2 * nw - 1]
    sel = [r for r in rows if lo <= r[1] < hi and "spin_kernel" not in r[0]]
    # the frame chain's stream = the one the tracker evaluations run on; an update = everything on the OTHER streams (map stream +
    # the free views' raycast stream) from one batch of keyframe views going up (upload_views_kernel) to the next.  In the
    # sequential schedule frames and update share a stream: there the update's kernels are those up to the next frame's
    # first kernel (rgba8_to_rgbf_kernel / track_prepare).
    cnt = defaultdict(int)
    for n, s, e, st in sel:
        if "track_eval" in n:
            cnt[st] += 1
    frame_stream = max(cnt, key=cnt.get)
    shared = any("preprocess_bwd" in r[0] and r[3] == frame_stream for r in sel)
    if shared:
        ms, other, inside = [], [], False
        for r in sel:
            if "upload_views" in r[0]:
                inside = True
            elif r[3] == frame_stream and ("rgba8_to_rgbf" in r[0] or "track_prepare" in r[0]):
                inside = False
            (ms if inside else other).append(r)
    else:
        ms = [r for r in sel if r[3] != frame_stream]
        other = [r for r in sel if r[3] == frame_stream]
    starts = [i for i, r in enumerate(ms) if "upload_views" in r[0] and (i == 0 or "upload_views" not in ms[i - 1][0])]
    print("frame stream %s (%s): update kernels %d, %d updates in the window set; frame-chain kernels: %d" %
          (frame_stream, "shared with the update" if shared else "update on other streams", len(ms), len(starts), len(other)))
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    with_iters = [u for u in range(len(starts) - 1) if any("preprocess_bwd" in r[0] for r in ms[starts[u]:starts[u + 1]])]
    for u in range(len(starts) - 1):
        ks = ms[starts[u]:starts[u + 1]]
        bw = [i for i, r in enumerate(ks) if "preprocess_bwd" in r[0]]
        if not bw:
            continue
        # the first iteration starts with the first sb_scan / preprocess_fwd before the first backward kernel
        first = bw[0]
        while first > 0 and not ("raster_ges_fwd" in ks[first][0]):
            first -= 1
        while first > 0 and any(t in ks[first - 1][0] for t in ("sb_scan", "sb_scatter", "preprocess_fwd")):
            first -= 1
        parts = (("pre", ks[:first]), ("iterations", ks[first:bw[-1] + 1]), ("post", ks[bw[-1] + 1:]))
        t_update0, t_update1 = ks[0][1], ks[-1][2]
        line = "update %d: %.0f us from first to last kernel;" % (u, (t_update1 - t_update0) / 1e3)
        prev_end = None
        for name, part in parts:
            if not part:
                continue
            wall = (part[-1][2] - (prev_end if prev_end else part[0][1])) / 1e3
            busy = 0.0
            cur_s, cur_e = part[0][1], part[0][2]
            for n, s, e, st in part[1:]:
                if s > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s, e
                else:
                    cur_e = max(cur_e, e)
            busy += cur_e - cur_s
            busy /= 1e3
            line += "  %s %.0f us (busy %.0f, %d kernels)" % (name, wall, busy, len(part))
            a = agg[name]
            a[0] += wall; a[1] += busy; a[2] += 1
            prev_end = part[-1][2]
        print(line)
        if with_iters and u == with_iters[-1]:   # the last complete update in detail: every kernel outside the iterations, gaps > 8 us inside them
            print("  -- detail of this update (t from its first kernel; gap = idle time on the map stream before the kernel)")
            last_end = ks[0][1]
            for i, (n, s, e, st) in enumerate(ks):
                gap = (s - last_end) / 1e3
                inside = first <= i <= bw[-1]
                if not inside or gap > 8.0:
                    beside = [short(o[0]) for o in other if o[1] < e and o[2] > s]
                    print("  %8.1f us  gap %6.1f  %7.1f us  %-44s %s" % ((s - ks[0][1]) / 1e3, gap, (e - s) / 1e3, short(n),
                                                                         ("| beside: " + ", ".join(sorted(set(beside)))[:90]) if beside else ""))
                last_end = max(last_end, e)
    print()
    for name, (wall, busy, k) in agg.items():
        print("mean %-10s wall %.0f us, busy %.0f us, idle %.0f us" % (name, wall / k, busy / k, (wall - busy) / k))


main()
