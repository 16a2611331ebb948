#!/bin/bash
# distribution of track_eval_poll_kernel durations in the sequential whole run (which evaluations are the long ones?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/prof_eh && TRACE_SCHEDULES=${1:-sequential} rocprofv3 --kernel-trace -d /tmp/prof_eh -o wr -- python tools/whole_run_trace.py ${2:-300} > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
import numpy as np
db = sqlite3.connect(glob.glob("/tmp/prof_eh/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if "spin_kernel" in n]
lo, hi = marks[-2], marks[-1]
ev = [(s, e) for n, s, e in rows if "track_eval_poll" in n and lo < s < hi]
d = np.array([e - s for s, e in ev]) / 1e3
gap = np.array([ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]) / 1e3
print("%d evaluations, duration us: mean %.2f median %.2f; percentiles 10/25/50/75/90/99: %s" % (len(d), d.mean(), np.median(d), np.percentile(d, [10, 25, 50, 75, 90, 99]).round(1)))
h, edges = np.histogram(d, bins=[0, 4, 6, 8, 10, 12, 14, 16, 18, 20, 25, 30, 50, 1000])
for c, a, b in zip(h, edges[:-1], edges[1:]):
    print("  %5.0f-%-5.0f us: %5d  (%.1f %% of the evaluations, %.1f %% of their time)" % (a, b, c, 100.0 * c / len(d), 100.0 * d[(d >= a) & (d < b)].sum() / d.sum()))
g = gap[gap < 200]
print("gap between consecutive evaluation kernels (same frame, < 200 us): mean %.2f median %.2f us" % (g.mean(), np.median(g)))
PY
