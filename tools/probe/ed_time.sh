#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/ed && rocprofv3 --kernel-trace --stats -d /tmp/ed -o ed -- python tools/probe/ed_time.py 2>&1 | grep -v "^\[gps\|rocprofv3\|amdgpu.ids" | tail -4
python - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/ed/**/*.db", recursive=True)[0])
for name in ("expected_depths_partial_kernel", "expected_depths_reduce_kernel"):
    r = db.execute("select end - start from kernels where name like ? order by start", ("%" + name + "%",)).fetchall()
    d = sorted(x[0] for x in r[-100:])
    print("%s: %d launches, median %.2f us, mean %.2f us, min %.2f" % (name, len(d), d[len(d) // 2] / 1e3, sum(d) / len(d) / 1e3, d[0] / 1e3))
PY
