"""Average duration per kernel name (full template signature kept) of a rocprofv3 --kernel-trace rocpd database.
usage: kernel_avgs.py <db> [substring filter]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for n, s, e in db.execute("select name,start,end from kernels"):
    if flt and flt not in n:
        continue
    k = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-72s n=%5d avg %8.2f us" % (k, a[0], a[1] / a[0] / 1e3))
