"""Summarise a rocprofv3 --kernel-trace rocpd sqlite database into per-kernel tables, ONE PER PHASE of bench.py.

bench.py drops a marker kernel (torch.cuda._sleep(1) -> at::cuda::spin_kernel, which nothing else in the run launches) at
the start and end of every timed region and once more before its measurement scaffolding (fusion-only split pass, roofline
micro-benchmark).  Kernels are assigned to the phase whose markers bracket their start time:

    setup            scene construction, priming run, prologue + warm-up frames of the first schedule
    timed:<k>        the k-th timed region (sequential, then overlap when both schedules run) -- the SLAM frames `value` counts
    between          warm-up frames of the next schedule
    scaffolding      split pass + micro-benchmarks (not SLAM frames)

usage: prof_summary.py <db> [rows] [--frames K]   (K = --steps of the profiled run: adds a per-frame column)
"""
import re
import sqlite3
import sys


def short(name):
    s = name.replace("(anonymous namespace)::", "")
    s = re.sub(r"^void ", "", s)
    s = re.sub(r"\(.*", "", s)
    return s.split("<")[0] if s.startswith(("at::", "void at::")) else s


def table(rows, top, frames=None):
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    hdr = "| kernel | calls | total_ms | avg_us | min_us | max_us | pct |" + (" us/frame |" if frames else "")
    print(hdr)
    print("|---" * (hdr.count("|") - 1) + "|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        line = "| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (k[:70], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                                  100.0 * a[1] / tot)
        if frames:
            line += " %.1f |" % (a[1] / 1e3 / frames)
        print(line)
    print("total kernel time %.3f ms over %d dispatches%s" % (tot / 1e6, len(rows), (", %.1f us of kernels per frame" % (tot / 1e3 / frames)) if frames else ""))


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else None
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [s for n, s, e in rows if "spin_kernel" in n]
    body = [(n, s, e) for n, s, e in rows if "spin_kernel" not in n]
    if len(marks) < 2:
        print("(no phase markers in this trace: whole run)")
        table([(n, e - s) for n, s, e in body], top)
        return
    # markers come in (start, end) pairs per timed region, then one scaffolding marker
    n_timed = (len(marks) - 1) // 2 if len(marks) % 2 else len(marks) // 2
    phases = []
    for k in range(n_timed):
        phases.append(("timed:%d" % k, marks[2 * k], marks[2 * k + 1]))
    scaffold_from = marks[2 * n_timed] if len(marks) > 2 * n_timed else None
    for name, lo, hi in phases:
        sel = [(n, e - s) for n, s, e in body if lo <= s < hi]
        print("\n### phase %s  (%.3f ms between the markers)\n" % (name, (hi - lo) / 1e6))
        table(sel, top, frames)
    if scaffold_from is not None:
        sel = [(n, e - s) for n, s, e in body if s >= scaffold_from]
        print("\n### phase scaffolding (fusion-only split pass + roofline micro-benchmarks; NOT SLAM frames)\n")
        table(sel, 12)
        for kname in ("raster_ges_bwd_gs_kernel", "raster_ges_fwd_pk_kernel"):
            d = [e - s for n, s, e in body if s >= scaffold_from and kname in n]
            if len(d) >= 50:
                tail = d[1:51] if kname == "raster_ges_bwd_gs_kernel" else d[-50 - 21:-21] if len(d) >= 71 else d[-50:]
                print("%s: the micro-benchmark's 50 launches avg %.2f us (min %.2f, max %.2f) -- bench.py's roofline.avg_launch_us must agree"
                      % (kname, sum(tail) / len(tail) / 1e3, min(tail) / 1e3, max(tail) / 1e3))


if __name__ == "__main__":
    main()
