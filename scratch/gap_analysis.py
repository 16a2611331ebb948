"""GPU idle time between consecutive kernel dispatches of a rocprofv3 kernel trace (rocpd sqlite), attributed to the kernel
that FOLLOWS the gap.  usage: gap_analysis.py trace.db [t0_frac] -- only dispatches after t0_frac of the run are counted."""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return (n.split("<")[0] if n.startswith("at::") else n)[:48]
# timed region of bench.py = its last 100 frames; one convert_depth_kernel per frame -> window of the last 99 frames
cd = [r[1] for r in rows if "convert_depth" in r[0]]
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # frames to skip at the end (bench.py: 120 of the fusion-only pass)
t_lo, t_hi = cd[-100 - off], cd[-1 - off]
rows = [r for r in rows if t_lo <= r[1] < t_hi]
print("window: %d frames, %.3f ms per frame" % (99, (t_hi - t_lo) / 99e6))
span = rows[-1][2] - rows[0][1]
busy = 0; gaps = collections.defaultdict(lambda: [0, 0]); prev_end = rows[0][1]
for n, s, e in rows:
    if s > prev_end:
        g = gaps[short(n)]; g[0] += 1; g[1] += s - prev_end
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
print("span %.1f ms, busy %.1f ms (%.0f%%), idle %.1f ms over %d dispatches" % (span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6, len(rows)))
print("idle time by the kernel that follows the gap:")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-50s gaps %5d  total %8.2f ms  avg %6.1f us" % (k, c, t / 1e6, t / c / 1e3))
