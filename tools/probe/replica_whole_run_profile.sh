#!/bin/bash
# kernel table of the Replica-shaped whole-sequence run (2,000 frames at 1200x680 from an empty model), one schedule:
#   gpurun -- 'bash tools/probe/replica_whole_run_profile.sh sequential'
SCHED=${1:-sequential}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf /tmp/prof_wrr && TRACE_SCHEDULES=$SCHED rocprofv3 --kernel-trace -d /tmp/prof_wrr -o wr -- python tools/whole_run_trace.py ${FRAMES:-2000} 1200 680 > gpurun_out/whole_run_replica_$SCHED.log 2>&1
python - <<'PY'
import glob, sqlite3, re
db = sqlite3.connect(glob.glob("/tmp/prof_wrr/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if "spin_kernel" in n]
lo, hi = marks[-2], marks[-1]
agg = {}
for n, s, e in rows:
    if "spin_kernel" in n or s < lo or s > hi:
        continue
    k = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    k = k.split("<")[0] if k.startswith("at::") else k
    a = agg.setdefault(k, [0, 0, 0])
    a[0] += 1; a[1] += e - s
    if s > lo + 0.9 * (hi - lo): a[2] += e - s          # the last tenth of the run
tot = sum(a[1] for a in agg.values())
print("| kernel | calls | total_ms | avg_us | pct | us/frame | pct in the last tenth |\n|---|---|---|---|---|---|---|")
t10 = sum(a[2] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("| %s | %d | %.2f | %.2f | %.1f | %.1f | %.1f |" % (k[:60], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot, a[1] / 1e3 / float(__import__("os").environ.get("FRAMES", "2000")), 100.0 * a[2] / max(1, t10)))
print("total kernel time %.1f ms between the run's markers (%.1f ms apart)" % (tot / 1e6, (hi - lo) / 1e6))
PY
grep "^$SCHED" gpurun_out/whole_run_replica_$SCHED.log | cut -c1-300
