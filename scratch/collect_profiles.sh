#!/bin/bash
# Runs on the GPU box (via gpurun): refreshes the artefacts kept under profiles/.
#   1. rocprofv3 --kernel-trace --stats of the default bench -> gpurun_out/r01_bench_kernel_stats.md
#   2. HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (one TCC counter group
#      each, MI355X guide "rocprofv3 PMC slots") over scratch/raster_bench.py -> gpurun_out/pmc_raster_bwd.json
#   3. the bench line itself -> gpurun_out/r01_bench_line.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps 100 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- $CMD > gpurun_out/bench_prof.log 2>&1
{
  echo "# Round 1 — rocprofv3 --kernel-trace --stats summary (MI355X, gfx950)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\` (640x480, C++ host, 120 SLAM frames incl. warm-up)."
  echo "Summarised from the rocpd database with scratch/prof_summary.py. knn_kernel's maximum is the untimed scene set-up"
  echo "(200k seed Gaussians)."
  echo
  python scratch/prof_summary.py "$(find /tmp/prof_stats -name '*.db' | head -1)" 48
  echo
  echo "bench line of the profiled run:"
  grep '^{"metric"' gpurun_out/bench_prof.log | tail -1 | cut -c1-400
} > gpurun_out/r01_bench_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o a -- python scratch/raster_bench.py > gpurun_out/pmc_$c.log 2>&1
done
# issue-side view of the same kernel (it is VALU bound): wave instructions by class, own pass
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/prof_SQ -o a -- python scratch/raster_bench.py > gpurun_out/pmc_SQ.log 2>&1
python - <<'PY'
import glob, json, sqlite3
def avg(counter):
    db = sqlite3.connect(glob.glob("/tmp/prof_%s/**/*.db" % counter, recursive=True)[0])
    rows = db.execute("select value from counters_collection where kernel_name like '%raster_ges_bwd_gs_kernel%' and counter_name=?", (counter,)).fetchall()
    v = [r[0] for r in rows]
    return sum(v) / len(v), len(v)
f, nf = avg("FETCH_SIZE")
w, nw = avg("WRITE_SIZE")
# rocprofv3 reports both in KiB.  FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B on gfx950 (MI355X guide,
# HBM section: "reports exactly 1/2 of the bytes of a wide coalesced read"); this kernel's reads are 4..16-B gathers and
# record loads, for which the guide gives no calibration, so both the raw and the doubled figure are kept and the doubled
# one (upper bound) is used as `traffic`.
sq = {}
for cname in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
    try:
        dbq = sqlite3.connect(glob.glob("/tmp/prof_SQ/**/*.db", recursive=True)[0])
        v = [r[0] for r in dbq.execute("select value from counters_collection where kernel_name like '%raster_ges_bwd_gs_kernel%' and counter_name=?", (cname,))]
        sq[cname] = sum(v) / len(v)
    except Exception:
        pass
out = {"kernel": "raster_ges_bwd_gs_kernel", "launches": nf, "wave_instructions_per_launch": sq, "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
       "fetch_bytes_raw": f * 1024, "fetch_bytes_x2": 2 * f * 1024, "write_bytes": w * 1024,
       "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
       "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over scratch/raster_bench.py; gfx950 FETCH_SIZE doubled per the guide's correction"}
json.dump(out, open("gpurun_out/pmc_raster_bwd.json", "w"), indent=1)
print(out)
PY
python bench.py > gpurun_out/bench_full.log 2>&1
tail -1 gpurun_out/bench_full.log > gpurun_out/r01_bench_line.json
cut -c1-300 gpurun_out/r01_bench_line.json
