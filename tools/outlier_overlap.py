"""Which launches of a kernel are outliers, and what ran beside them?  From a rocprofv3 --kernel-trace rocpd database of bench.py:
for every named kernel, the calls inside the timed windows that took more than `factor` x the kernel's median, each with the kernels
of OTHER streams whose execution overlapped it (name, grid z, overlap in us), and the position of the call inside its keyframe
update (index among the update's launches of that kernel).
usage: outlier_overlap.py <db> [factor, default 2.0] [kernel substrings, default: the three the round-3 review named]"""
import re
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
factor = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
names = sys.argv[3:] or ["raster_ges_fwd_pk_kernel", "sb_scatter_kernel", "preprocess_fwd_kernel"]
rows = list(db.execute("select name,start,end,stream_id,grid_z from kernels order by start"))
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:44]
marks = [r[1] for r in rows if "spin_kernel" in r[0]]
windows = [(marks[i], marks[i + 1]) for i in range(0, len(marks) - 1, 2)] if len(marks) >= 2 else [(rows[0][1], rows[-1][2])]
in_window = lambda t: any(lo <= t < hi for lo, hi in windows)
updates = [r[1] for r in rows if "upload_views" in r[0]]   # first kernel of every free-view batch: two batches open an update
for nm in names:
    calls = [r for r in rows if nm in r[0] and in_window(r[1])]
    if not calls:
        continue
    durs = sorted((r[2] - r[1]) / 1e3 for r in calls)
    med = durs[len(durs) // 2]
    slow = [r for r in calls if (r[2] - r[1]) / 1e3 > factor * med]
    print("## %s: %d calls in the timed windows, median %.1f us, %d above %.1f x median (max %.1f us)" % (nm, len(calls), med, len(slow), factor, durs[-1]))
    beside = Counter()
    for n, s, e, st, gz in slow:
        prev = [u for u in updates if u <= s]
        since = [c for c in calls if prev and prev[-1] <= c[1] <= s]
        others = []
        for n2, s2, e2, st2, gz2 in rows:
            if st2 == st or e2 <= s or s2 >= e or "spin_kernel" in n2:
                continue
            ov = (min(e, e2) - max(s, s2)) / 1e3
            others.append("%s z=%d %.0f of its %.0f us" % (short(n2), gz2, ov, (e2 - s2) / 1e3))
            beside[short(n2) + (" z>1" if gz2 > 1 else "")] += 1
        print("  %.1f us, call #%d of this kernel since the last free-view batch began (%.0f us earlier); beside it: %s" % (
            (e - s) / 1e3, len(since), (s - prev[-1]) / 1e3 if prev else -1, "; ".join(others[:6]) or "nothing on another stream"))
    if slow:
        print("  -> kernels running beside the slow calls:", dict(beside))
