# ATen / runtime kernels inside the last timed window of a short sequential bench run: gpurun -- bash tools/probe/aten_names.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/an && rocprofv3 --kernel-trace -d /tmp/an -o a -- python bench.py --steps 20 --warmup 5 --windows 2 --schedule sequential --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3, collections
db = sqlite3.connect(glob.glob("/tmp/an/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if "spin_kernel" in n]
lo, hi = marks[-2], marks[-1]
c = collections.Counter(); t = collections.Counter()
for n, s, e in rows:
    if lo < s < hi and ("at::" in n or "rocclr" in n or "Memset" in n or "fill" in n.lower()):
        c[n[:230]] += 1; t[n[:230]] += e - s
for n, k in c.most_common():
    print(k, "%.1f us avg" % (t[n] / k / 1e3), n)
PY
