"""How much do the streams of a schedule really run side by side?  From a rocprofv3 --kernel-trace rocpd database of
`bench.py --schedule <one>`: over the first timed region (between bench.py's first two marker kernels, see prof_summary.py),
wall time, per-stream kernel time, the union of busy intervals and the time two or more kernels were in flight, per frame.
usage: stream_overlap.py <db> <frames in the timed region (--steps)>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
q = "select name,start,end,%s from kernels order by start" % (key or "0")
rows = list(db.execute(q))
marks = [r[1] for r in rows if "spin_kernel" in r[0]]
t0, t1 = marks[0], marks[1]
per = collections.defaultdict(float)
ev = []
for n, s, e, st in rows:
    if s < t0 or s >= t1 or "spin_kernel" in n: continue
    per[st] += e - s
    ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, busy, multi = 0, t0, 0.0, 0.0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
n = frames
print("wall %.1f us/frame; busy (union) %.1f; >=2 kernels in flight %.1f; idle %.1f" % ((t1 - t0) / n / 1e3, busy / n / 1e3, multi / n / 1e3, ((t1 - t0) - busy) / n / 1e3))
for st, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print("  %s %s: %.1f us/frame of kernels" % (key, st, v / n / 1e3))
