"""Kernel timeline of ONE frame (convert_depth .. next convert_depth) inside the first timed region of a bench.py
kernel trace (rocprofv3 --kernel-trace rocpd database).  usage: frame_timeline.py <db> [frame index in the region]"""
import glob, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = db.execute("select name,start,end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
sel = rows[marks[0]:marks[1]]
idx = [i for i, r in enumerate(sel) if "convert_depth" in r[0]]
i0, i1 = idx[k], idx[k + 1]
t0 = sel[i0][1]
prev_end = t0
for r in sel[i0:i1]:
    n = r[0].replace("(anonymous namespace)::", "").split("(")[0][:44]
    print("%-46s start %8.1f us  dur %6.1f  gap %5.1f" % (n, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3))
    prev_end = r[2]
print("frame: %.1f us" % ((sel[i1][1] - t0) / 1e3))
