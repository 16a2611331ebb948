# per-kernel average durations of a probe program under rocprofv3 --kernel-trace: bash tools/probe/kavg.sh <python args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk && rocprofv3 --kernel-trace -d /tmp/pk -o a -- python $R/"$@" > /tmp/kavg_prog.log 2>&1
tail -3 /tmp/kavg_prog.log
python - <<'PY'
import glob, sqlite3, re, collections
db = sqlite3.connect(glob.glob("/tmp/pk/**/*.db", recursive=True)[0])
agg = collections.OrderedDict()
for n, s, e, gz in db.execute("select name,start,end,grid_z from kernels order by start"):
    k = (re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60], gz)
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for (k, gz), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s grid.z %3d n=%5d avg %8.2f us" % (k, gz, a[0], a[1] / a[0] / 1e3))
PY
