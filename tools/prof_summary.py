"""Summarise a rocprofv3 --kernel-trace rocpd sqlite database into per-kernel tables, ONE PER PHASE of bench.py.

bench.py drops a marker kernel (torch.cuda._sleep(1) -> at::cuda::spin_kernel, which nothing else in the run launches) at
the start and end of every timed region and once more before its measurement scaffolding (fusion-only split pass, roofline
micro-benchmark).  Kernels are assigned to the phase whose markers bracket their start time:

    setup            scene construction, priming run, prologue + warm-up frames of the first schedule
    timed            the timed windows -- the SLAM frames `value` counts.  bench.py times --windows consecutive windows per
                     schedule (sequential, then overlap when both run): with --windows NW the NW windows of a schedule are
                     summarised as ONE table (the per-frame column divides by NW x K frames)
    between          warm-up frames of the next schedule
    scaffolding      split pass + micro-benchmarks (not SLAM frames)

usage: prof_summary.py <db> [rows] [--frames K] [--windows NW]   (K = --steps of the profiled run: adds a per-frame column)
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
    nw = int(sys.argv[sys.argv.index("--windows") + 1]) if "--windows" in sys.argv else 1
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [s for n, s, e in rows if "spin_kernel" in n]
    body = [(n, s, e) for n, s, e in rows if "spin_kernel" not in n]
    if len(marks) < 2:
        print("(no phase markers in this trace: whole run)")
        table([(n, e - s) for n, s, e in body], top)
        return
    # markers come in (start, end) pairs per timed window: nw windows per schedule, 1 or 2 schedules (round 6: + one pair per
    # schedule around the in-loop launch-timing window, --in-loop); then one scaffolding marker followed by one marker in front of
    # every micro-benchmark loop
    il = 1 if "--in-loop" in sys.argv else 0
    per = nw + il
    n_sched = 2 if len(marks) >= 4 * per + 1 else 1
    n_timed = per * n_sched
    if len(marks) < 2 * n_timed:
        n_timed, n_sched, nw, per, il = len(marks) // 2, 1, len(marks) // 2, len(marks) // 2, 0
    names = ("sequential", "overlap") if n_sched == 2 else ("timed",)
    for sidx in range(n_sched):
        sel, span = [], 0
        for k in range(sidx * per, sidx * per + nw):
            lo, hi = marks[2 * k], marks[2 * k + 1]
            sel += [(n, e - s) for n, s, e in body if lo <= s < hi]
            span += hi - lo
        print("\n### schedule %s: %d timed window(s), %.3f ms between their markers\n" % (names[sidx], nw, span / 1e6))
        table(sel, top, frames * nw if frames else None)
        if il:
            k = sidx * per + nw
            lo, hi = marks[2 * k], marks[2 * k + 1]
            print("\n### schedule %s: the in-loop launch-timing window (%d extra frames after the timed windows, every launch of the instrumented "
                  "kernels stamps the device clock; %.3f ms between its markers) -- the bench line's roofline.avg_launch_us is the "
                  "stamp-measured average of THESE launches\n" % (names[sidx], frames or 0, (hi - lo) / 1e6))
            table([(n, e - s) for n, s, e in body if lo <= s < hi], 12, frames)
    scaffold_from = marks[2 * n_timed] if len(marks) > 2 * n_timed else None
    if scaffold_from is not None:
        sel = [(n, e - s) for n, s, e in body if s >= scaffold_from]
        print("\n### phase scaffolding (fusion-only split pass + roofline micro-benchmarks; NOT SLAM frames)\n")
        table(sel, 12)
        micro = marks[2 * n_timed + 1:]
        for j, lo in enumerate(micro):
            hi = micro[j + 1] if j + 1 < len(micro) else (1 << 62)
            loop = {}
            for n, s0, e in body:
                if lo <= s0 < hi:
                    loop.setdefault(short(n), []).append(e - s0)
            if loop:
                k, d = max(loop.items(), key=lambda kv: sum(kv[1]))
                d = d[1:] if len(d) > 1 else d
                print("micro-benchmark loop %d: %s x %d, avg %.2f us (min %.2f, max %.2f) -- the bench line's live average must agree"
                      % (j, k[:60], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))

if __name__ == "__main__":
    main()
