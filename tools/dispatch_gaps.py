"""Overlap schedule: how long do the frame chain's kernels wait for dispatch, and behind what?  From a rocprofv3 --kernel-trace
rocpd database of `bench.py --schedule overlap`: for every kernel of the frame stream (the one the tracker's evaluations run on)
that follows another frame-stream kernel, gap = its start - the previous one's end.  Gaps > `thr` us are attributed to the map
stream's kernel that was running when the previous kernel ended, with the point of THAT kernel's life at which the waiting kernel
finally started (0 = its start, 1 = its end).  usage: dispatch_gaps.py <db> [thr_us, default 12] [windows, default 3]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
marks = [s for n, s, e, st in rows if "spin_kernel" in n]
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lo, hi = (marks[0], marks[2 * nw - 1]) if len(marks) >= 2 * nw else (rows[0][1], rows[-1][2])   # the timed windows of the (only) schedule
sel = [r for r in rows if lo <= r[1] < hi and "spin_kernel" not in r[0]]
cnt = defaultdict(int)
for n, s, e, st in sel:
    if "track_eval" in n:
        cnt[st] += 1
fs = max(cnt, key=cnt.get)
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:34]
frame = [r for r in sel if r[3] == fs]
other = [r for r in sel if r[3] != fs]
import bisect
ostart = [r[1] for r in other]
tot_gap = 0.0
by = defaultdict(list)
small = 0.0
for a, b in zip(frame, frame[1:]):
    if "track_eval" in b[0] and "track_eval" in a[0]:
        continue   # (between evaluations the host decides: not a dispatch wait)
    gap = (b[1] - a[2]) / 1e3
    if gap <= thr:
        small += max(gap, 0.0)
        continue
    # the other-stream kernel running at a.end with the latest start
    i = bisect.bisect_right(ostart, a[2]) - 1
    running = None
    for j in range(i, max(-1, i - 6), -1):
        if other[j][2] > a[2]:
            running = other[j]
            break
    tot_gap += gap
    if running:
        phase = (b[1] - running[1]) / max(1.0, running[2] - running[1])
        by[short(running[0])].append((gap, phase, short(b[0])))
    else:
        by["(nothing running)"].append((gap, 0.0, short(b[0])))
frames = sum(1 for r in frame if "integrate_kernel" in r[0])
print("frame stream %s: %d frames; gaps > %.0f us: %.1f us per frame, smaller gaps %.1f us per frame" % (fs, frames, thr, tot_gap / max(1, frames), small / max(1, frames)))
for k, v in sorted(by.items(), key=lambda kv: -sum(g for g, p, n in kv[1])):
    g = [x[0] for x in v]; p = sorted(x[1] for x in v)
    print("%-36s %4d waits, %7.1f us per frame, mean %.1f us; the waiting kernel started at phase (quartiles) %.2f %.2f %.2f of it" %
          (k, len(v), sum(g) / max(1, frames), sum(g) / len(g), p[len(p) // 4], p[len(p) // 2], p[(3 * len(p)) // 4]))
