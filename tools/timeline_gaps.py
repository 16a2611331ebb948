"""Largest GPU-idle gaps inside the first timed region of a bench.py kernel trace (rocprofv3 --kernel-trace rocpd database):
which kernels precede / follow them.  usage: timeline_gaps.py <db>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i,(n,s,e) in enumerate(rows) if "spin_kernel" in n]
lo, hi = marks[0], marks[1]
sel = rows[lo:hi+1]
gaps = sorted(((sel[i+1][1]-sel[i][2], i) for i in range(len(sel)-1)), reverse=True)[:6]
for g,i in gaps:
    print("gap %.2f ms after #%d" % (g/1e6, i))
    for k in range(max(0,i-3), min(len(sel), i+5)):
        n = re.sub(r"\(.*","",sel[k][0].replace("(anonymous namespace)::",""))[:70]
        print("    %s%-70s %.1f us  t=%.3f ms" % ("*" if k==i else " ", n, (sel[k][2]-sel[k][1])/1e3, (sel[k][1]-sel[0][1])/1e6))
