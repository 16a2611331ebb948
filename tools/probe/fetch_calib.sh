#!/bin/bash
# gpurun -- 'bash tools/probe/fetch_calib.sh' -> gpurun_out/fetch_calib.txt: FETCH_SIZE of tools/probe/fetch_calib.hip's kernels against their known bytes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
[ -x tools/probe/fetch_calib ] || hipcc --offload-arch=gfx950 -O2 -o tools/probe/fetch_calib tools/probe/fetch_calib.hip || exit 1
rm -rf /tmp/fc && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o a -- tools/probe/fetch_calib > gpurun_out/fetch_calib.log 2>&1
python - <<'PY' | tee gpurun_out/fetch_calib.txt
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/fc/**/*.db", recursive=True)[0])
rows = db.execute("select c.kernel_name, c.value from counters_collection c where c.counter_name='FETCH_SIZE' order by c.dispatch_id").fetchall()
B, L = 2 << 30, (2 << 30) // 128
known = {"stream16": ("2 GiB streamed, 16 B / lane", B), "stream4": ("2 GiB streamed, 4 B / lane", B),
         "gather<4>": ("4 B from each of %d lines" % L, None), "gather<8>": ("8 B from each line", None),
         "gather<16>": ("16 B from each line", None), "gather<64>": ("64 B from each line", None)}
print("| kernel | pattern | FETCH_SIZE (KB -> bytes) | per 128-B line touched | x factor to the bytes the pattern must move |")
print("|---|---|---|---|---|")
seen = set()
for name, v in rows[len(rows) // 2:]:     # second repetition
    k = next((k for k in known if k in name), None)
    if not k or k in seen:
        continue
    seen.add(k)
    fb = v * 1024.0
    what, must = known[k]
    print("| %s | %s | %.0f KB = %.3f GiB | %.1f B | %s |" % (k, what, v, fb / 2**30, fb / L, "%.2f" % (must / fb) if must else
          "32-B sectors: %.2f, 64 B: %.2f, 128 B: %.2f" % (32.0 * L / fb, 64.0 * L / fb, 128.0 * L / fb)))
PY
# second pass: the request-size classes behind FETCH_SIZE (TCC_EA0_RDREQ = all read requests, _32B / _64B / _128B by size)
rm -rf /tmp/fc2 && rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/fc2 -o a -- tools/probe/fetch_calib > gpurun_out/fetch_calib2.log 2>&1
python - <<'PY' | tee -a gpurun_out/fetch_calib.txt
import glob, sqlite3, collections
f = glob.glob("/tmp/fc2/**/*.db", recursive=True)
if not f:
    raise SystemExit("no database from the request-size pass (see gpurun_out/fetch_calib2.log)")
db = sqlite3.connect(f[0])
rows = db.execute("select c.dispatch_id, c.kernel_name, c.counter_name, c.value from counters_collection c order by c.dispatch_id").fetchall()
per = collections.OrderedDict()
for d, k, c, v in rows:
    per.setdefault((d, k), {})[c] = v
L = (2 << 30) // 128
print()
print("| kernel | RDREQ | 32 B | 64 B | 128 B | bytes by size class | per line touched |")
print("|---|---|---|---|---|---|---|")
items = list(per.items())
for (d, k), c in items[len(items) // 2:]:
    n = c.get("TCC_EA0_RDREQ_sum", 0); a = c.get("TCC_EA0_RDREQ_32B_sum", 0); b = c.get("TCC_EA0_RDREQ_64B_sum", 0); e = c.get("TCC_EA0_RDREQ_128B_sum", 0)
    by = 32 * a + 64 * b + 128 * e
    print("| %s | %.0f | %.0f | %.0f | %.0f | %.3f GiB | %.1f B |" % (k.split("(")[0][-40:], n, a, b, e, by / 2**30, by / L))
PY
