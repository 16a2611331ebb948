"""Summarise a rocprofv3 rocpd sqlite (kernel trace) into a per-kernel stats table (markdown/csv-ish)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, (end-start) as dur from kernels").fetchall()
agg = {}
for name, dur in rows:
    short = name.replace("(anonymous namespace)::", "")
    short = re.sub(r"^void ", "", short)
    short = re.sub(r"\(.*", "", short)
    short = short.split("<")[0] if short.startswith(("at::", "void at::")) else short
    a = agg.setdefault(short, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values())
print("| kernel | calls | total_ms | avg_us | min_us | max_us | pct |")
print("|---|---|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (k[:70], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))
# the roofline micro-benchmark of bench.py launches the dominant kernel 1 + 50 times at the end of the run: its average is
# the number bench.py's `roofline.avg_launch_us` must agree with (the whole-run average mixes all scene sizes)
for kname in ("raster_ges_bwd_gs_kernel", "raster_ges_fwd_pk_kernel"):
    d = [r[0] for r in cur.execute("select (end-start) from kernels where name like ? order by start", ("%" + kname + "%",))]
    if len(d) > 50:
        tail = d[-50:]
        print("%s: last 50 launches (bench.py roofline micro-benchmark) avg %.2f us, min %.2f, max %.2f" %
              (kname, sum(tail) / 50e3, min(tail) / 1e3, max(tail) / 1e3))
