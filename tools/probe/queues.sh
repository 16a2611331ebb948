#!/bin/bash
# which hardware queue did every chain of the overlap whole-run get?  (kernels.queue_id of a rocprofv3 kernel trace of bench.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/prof_q && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_q -o q -- python bench.py --no-cpu-baseline --no-oracle-psnr $1 > /tmp/q.log 2>&1
grep '^{"metric"' /tmp/q.log | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); w=j['config']['whole_run']
print('value %.1f whole-run seq %.1f overlap %.1f' % (j['value'], w['sequential']['fps'], w['overlap']['fps']))"
python - <<'PY'
import glob, sqlite3, collections
db = sqlite3.connect(glob.glob("/tmp/prof_q/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("kernels columns:", cols)
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall() if "stream_id" in cols else [(a, b, c, d, None) for a, b, c, d in db.execute("select name, start, end, queue_id from kernels order by start").fetchall()]
marks = [s for n, s, e, q, st in rows if "spin_kernel" in n]
lo, hi = marks[-2], marks[-1]
per = collections.defaultdict(lambda: collections.Counter())
for n, s, e, q, st in rows:
    if lo < s < hi and "spin_kernel" not in n:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
        per[(q, st)][short] += 1
for (q, st), c in sorted(per.items(), key=lambda kv: str(kv[0])):
    print("queue", q, "stream", st, ":", ", ".join("%s x%d" % kv for kv in c.most_common(6)))
try:
    mc = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
    print("memory_copies columns:", mc)
    q = "select queue_id, stream_id, count(*), sum(end-start) from memory_copies where start > %d and start < %d group by 1, 2" % (lo, hi) if "queue_id" in mc else "select stream_id, count(*), sum(end-start) from memory_copies where start > %d and start < %d group by 1" % (lo, hi)
    for r in db.execute(q): print("copies:", r)
except Exception as e:
    print("copies:", e)
PY
