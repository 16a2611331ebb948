#!/bin/bash
# kernel table of the whole-sequence run (1,000 frames from an empty model), one schedule: gpurun -- 'bash tools/probe/whole_run_profile.sh sequential r05'
SCHED=${1:-sequential}
TAG=${2:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf /tmp/prof_wr && TRACE_SCHEDULES=$SCHED rocprofv3 --kernel-trace --stats -d /tmp/prof_wr -o wr -- python tools/whole_run_trace.py 1000 > gpurun_out/whole_run_prof_$SCHED.log 2>&1
{
  echo "# $TAG -- whole-sequence run ($SCHED schedule): 1,000 frames from an empty model, rocprofv3 --kernel-trace (MI355X)"
  echo
  python - <<PY
import glob, sqlite3, re
db = sqlite3.connect(glob.glob("/tmp/prof_wr/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if "spin_kernel" in n]
lo, hi = marks[-2], marks[-1]
agg = {}
for n, s, e in rows:
    if "spin_kernel" in n or s < lo or s > hi:
        continue
    k = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    k = k.split("<")[0] if k.startswith("at::") else k
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print("| kernel | calls | total_ms | avg_us | pct | us/frame |\n|---|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("| %s | %d | %.2f | %.2f | %.1f | %.1f |" % (k[:60], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot, a[1] / 1e3 / 1000))
print("total kernel time %.1f ms between the run's markers (%.1f ms apart)" % (tot / 1e6, (hi - lo) / 1e6))
PY
  echo
  grep "^$SCHED" gpurun_out/whole_run_prof_$SCHED.log | cut -c1-500
} > gpurun_out/${TAG}_whole_run_${SCHED}_kernel_stats.md
cat gpurun_out/${TAG}_whole_run_${SCHED}_kernel_stats.md
