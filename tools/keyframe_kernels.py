"""Kernel sequence of ONE keyframe's map update (sequential schedule) from a bench.py kernel trace: everything between the
keyframe's icp_kernel and the next frame's convert_depth_kernel, with the 20 optimise iterations collapsed.
usage: keyframe_kernels.py <db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
sel = rows[marks[0]:marks[1]]
conv = [i for i, r in enumerate(sel) if "convert_depth" in r[0]]
seg = sel[conv[0]:conv[1]]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]
t0 = seg[0][1]
in_iter = 0
tot_torch = 0.0
for n, s, e in seg:
    k = short(n)
    if k.startswith("preprocess_fwd_kernel"):
        in_iter += 1
    mine = not (k.startswith("at::") or k.startswith("rocprim") or k.startswith("__amd"))
    if not mine:
        tot_torch += (e - s) / 1e3
    if in_iter > 2 and in_iter < 21 and not k.startswith("at::") and not k.startswith("rocprim"):
        continue
    print("%9.1f us  %6.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, "" if mine else "   * ", k))
print("torch / rocprim / runtime kernels in this keyframe period: %.1f us" % tot_torch)
