"""Timeline of ONE tracked frame from a rocprofv3 --kernel-trace rocpd database: every kernel between two consecutive
track_prepare launches near the end of the run, with its start offset, duration and the idle gap in front of it.
usage: frame_timeline.py <db> [frames from the end, default 5]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(db.execute("select name,start,end,stream_id from kernels order by start")) if any(
    r[1] == "stream_id" for r in db.execute("pragma table_info(kernels)")) else [
    (n, s, e, 0) for n, s, e in db.execute("select name,start,end from kernels order by start")]
prep = [i for i, r in enumerate(rows) if "track_prepare" in r[0]]
a, b = prep[-back - 1], prep[-back]
t0, last_end = rows[a][1], rows[a][1]
busy = 0
for n, s, e, st in rows[a:b]:
    k = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:44]
    print("%8.1f us  +%6.1f gap  %7.1f us  s%-3s %s" % ((s - t0) / 1e3, (s - last_end) / 1e3, (e - s) / 1e3, st, k))
    last_end = max(last_end, e); busy += e - s
print("frame %.1f us, kernels %.1f us" % ((rows[b][1] - t0) / 1e3, busy / 1e3))
