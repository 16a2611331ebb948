"""Kernel-trace view of fwd_bench's last probe (first M tiles non-empty): per-dispatch durations of raster_ges_fwd_pk_kernel from
rocprofv3's CSV, grouped like the probe launched them (6 values of M x 3 repeats x (1 warm-up + 50) launches at the end of the run).
usage: rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o x -- python tools/probe/fwd_bench.py; python tools/probe/fwd_trace.py /tmp/p"""
import csv
import glob
import sys

rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "raster_ges_fwd_pk_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
tail = rows[-(6 * 3 * 51):]
for i, M in enumerate((64, 256, 512, 768, 1024, 1200)):
    grp = tail[i * 153:(i + 1) * 153]
    best = 1e9
    gaps = []
    for k in range(3):
        g = grp[k * 51 + 1:(k + 1) * 51]
        best = min(best, sum(e - s for s, e in g) / len(g))
        gaps.append((g[-1][1] - g[0][0]) / len(g))
    print("M %4d: kernel duration %.1f us (best of 3 averages), start-to-end per launch %.1f us" % (M, best * 1e-3, min(gaps) * 1e-3))
