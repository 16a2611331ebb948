"""Kernel timeline (all streams) around one keyframe update of the overlap schedule, from a rocprofv3 --kernel-trace rocpd
database of `bench.py --schedule overlap`: from the first free-view batch kernel of the last complete update in the first
timed region, `before` microseconds earlier to `after` microseconds later.  usage: keyframe_timeline.py <db> [before] [after] [which update, default -2]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
before = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
after = float(sys.argv[3]) if len(sys.argv) > 3 else 1600.0
rows = list(db.execute("select name,start,end,stream_id,grid_z from kernels order by start"))
marks = [r[1] for r in rows if "spin_kernel" in r[0]]
lo, hi = (marks[0], marks[1]) if len(marks) >= 2 else (rows[0][1], rows[-1][2])
ups = [r[1] for r in rows if "upload_views" in r[0] and lo <= r[1] < hi]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -2
t0 = ups[which] if len(ups) >= abs(which) else ups[-1]
for n, s, e, st, gz in rows:
    if s < t0 - before * 1e3 or s > t0 + after * 1e3: continue
    k = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
    print("%9.1f us  %7.1f us  stream %-3s z=%-2d %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, gz, k))
