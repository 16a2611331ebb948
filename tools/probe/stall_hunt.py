"""Find the rare multi-millisecond frame of the overlap schedule in a rocprofv3 --kernel-trace rocpd database and show what the
GPU did meanwhile: frames = intervals between consecutive track_prepare launches; for the longest ones (after the settle-in)
every kernel of every queue in the interval, with its queue / stream id, start offset, duration.
usage: stall_hunt.py <db> [threshold ms, default 3.0]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = "name,start,end" + ("," + qcol if qcol else ",0") + ("," + scol if scol else ",0")
rows = db.execute("select %s from kernels order by start" % sel).fetchall()
prep = [i for i, r in enumerate(rows) if "track_prepare" in r[0]]
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
frames = [(rows[prep[k + 1]][1] - rows[prep[k]][1], k) for k in range(60, len(prep) - 1)]
slow = [(d, k) for d, k in frames if d > thr * 1e6]
print("%d frames after settle-in, median %.3f ms, %d above %.1f ms:" % (len(frames), sorted(frames)[len(frames) // 2][0] / 1e6, len(slow), thr),
      [(k, round(d / 1e6, 2)) for d, k in slow][:40])
# frames directly behind a keyframe are long by construction (~1.5 ms); anything far above that is the hunted stall
for d, k in sorted(slow, reverse=True)[:2]:
    a, b = prep[k], prep[k + 1]
    t0 = rows[a][1]
    print("---- frame #%d: %.3f ms" % (k, d / 1e6))
    last = {}
    for n, s, e, q, st in rows[a:b + 1]:
        key = (q, st)
        gap = (s - last.get(key, s)) / 1e3
        last[key] = e
        name = short(n)
        if e - s > 30e3 or gap > 200 or "track_eval" in name or "track_prepare" in name:
            print("%9.1f us  q%-3s s%-3s %8.1f us  (+%7.1f us behind its queue's previous)  %s" % ((s - t0) / 1e3, q, st, (e - s) / 1e3, gap, name))
