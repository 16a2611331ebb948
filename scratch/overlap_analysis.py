"""Timeline of the overlapped schedule from a rocprofv3 kernel trace: per HW queue busy time and their union over the timed frames."""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
print("columns:", cols)
rows = cur.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
cd = [r[1] for r in rows if "convert_depth" in r[0]]
off = int(sys.argv[2]) if len(sys.argv) > 2 else 120
t_lo, t_hi = cd[-100 - off], cd[-1 - off]
rows = [r for r in rows if t_lo <= r[1] < t_hi]
print("window %.3f ms per frame over 99 frames, %d dispatches" % ((t_hi - t_lo) / 99e6, len(rows)))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
byq = collections.defaultdict(list)
for n, s, e, q in rows: byq[q].append((s, e))
for q, iv in byq.items():
    print("queue %s: %d kernels, busy %.1f ms, sum %.1f ms" % (q, len(iv), union(iv) / 1e6, sum(e - s for s, e in iv) / 1e6))
allv = [(s, e) for _, s, e, _ in rows]
print("union busy %.1f ms of %.1f ms (%.0f%%); sum of durations %.1f ms" % (union(allv) / 1e6, (t_hi - t_lo) / 1e6, 100 * union(allv) / (t_hi - t_lo), sum(e - s for s, e in allv) / 1e6))
def short(n):
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return (n.split("<")[0] if n.startswith("at::") else n)[:40]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in rows: a = agg[short(n)]; a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-42s %5d calls  %7.2f ms  avg %6.1f us" % (k, c, t / 1e6, t / c / 1e3))
