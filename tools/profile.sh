#!/bin/bash
# Regenerates the artefacts kept under profiles/ (run ON the GPU box:  gpurun -- 'bash tools/profile.sh rNN').
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command, summarised PER PHASE (tools/prof_summary.py): the
#      table of the timed SLAM frames only -> gpurun_out/<tag>_bench_kernel_stats.md
#   2. HBM traffic + instruction mix of the dominant kernel on THE SAME scene: FETCH_SIZE, WRITE_SIZE and the SQ counters in
#      SEPARATE --pmc passes (one counter group each, MI355X guide "rocprofv3 PMC slots"; never combined with trace domains
#      other than --kernel-trace) over the same bench command; the counters of the roofline micro-benchmark's launches (fixed
#      N, G -- recorded from that run's own bench line) -> gpurun_out/pmc_raster_bwd.json
#   3. the bench line itself -> gpurun_out/<tag>_bench_line.json
# Copy the three files into profiles/ and commit them.
TAG=${1:-rXX}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CMD="python bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-oracle-psnr"
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- $CMD > gpurun_out/bench_prof.log 2>&1
{
  echo "# $TAG -- rocprofv3 --kernel-trace --stats summary (MI355X, gfx950)"
  echo
  echo "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`.  Summarised per phase from the rocpd database with"
  echo "tools/prof_summary.py (phase markers: bench.py's spin_kernel launches).  \`timed:0\` = sequential schedule,"
  echo "\`timed:1\` = overlap schedule (the one \`value\` reports); us/frame = total / $STEPS timed frames."
  python tools/prof_summary.py "$(find /tmp/prof_stats -name '*.db' | head -1)" 40 --frames $STEPS
  echo
  echo "bench line of the profiled run:"
  grep '^{"metric"' gpurun_out/bench_prof.log | tail -1 | cut -c1-600
} > gpurun_out/${TAG}_bench_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c && rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o a -- $CMD > gpurun_out/pmc_$c.log 2>&1
done
rm -rf /tmp/prof_SQ && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/prof_SQ -o a -- $CMD > gpurun_out/pmc_SQ.log 2>&1
rm -rf /tmp/prof_SQ2 && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/prof_SQ2 -o a -- $CMD > gpurun_out/pmc_SQ2.log 2>&1
python - <<'PY'
import glob, json, sqlite3
K = "raster_ges_bwd_gs_kernel"
def series(dirname, counter):
    """per-dispatch counter values of kernel K in dispatch order"""
    db = sqlite3.connect(glob.glob("/tmp/%s/**/*.db" % dirname, recursive=True)[0])
    return [r[0] for r in db.execute("select c.value from counters_collection c where c.kernel_name like ? and c.counter_name=? "
                                     "order by c.dispatch_id", ("%" + K + "%", counter))]
def micro(v):
    """the roofline micro-benchmark = the LAST 51 launches of the kernel that are not train steps: 1 + 50 back-to-back
    launches on one fixed scene, followed by 1 + 20 train steps; drop the trailing 21"""
    return v[-(51 + 21):-21][1:] if len(v) >= 72 else v[-50:]
def line(logname):
    for l in reversed(open(logname).read().splitlines()):
        if l.startswith('{"metric"'):
            return json.loads(l)
    return None
out = {"kernel": K, "command": "bench.py --steps/--warmup as tools/profile.sh; counters of the roofline micro-benchmark's 50 launches"}
try:
    f, w = micro(series("prof_FETCH_SIZE", "FETCH_SIZE")), micro(series("prof_WRITE_SIZE", "WRITE_SIZE"))
    fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024   # rocprofv3 reports both in KiB
    # FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B on gfx950 (MI355X guide, HBM section); doubled per the
    # guide's correction -> an upper bound for this kernel's mix of 16-B gathers and record loads
    out.update(launches=len(f), fetch_bytes_raw=fb, fetch_bytes_x2=2 * fb, write_bytes=wb, hbm_bytes_per_launch=2 * fb + wb)
    b = line("gpurun_out/pmc_FETCH_SIZE.log")
    out["units"] = b["roofline"]["units"]
    out["algorithmic_bytes"] = b["roofline"]["algorithmic_bytes"]
except Exception as e:
    out["error_traffic"] = repr(e)
try:
    sq = {c: micro(series("prof_SQ", c)) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")}
    sq = {c: sum(v) / len(v) for c, v in sq.items() if v}
    b = line("gpurun_out/pmc_SQ.log")
    t = b["roofline"]["avg_launch_us"] * 1e-6   # launch time OF THE SAME RUN the counters were collected in
    out["valu"] = {"wave_instructions_per_launch": sq, "launch_us_same_run": t * 1e6,
                   "issue_frac": sq["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * t),
                   "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time of the same profiled run)"}
    s2 = {c: micro(series("prof_SQ2", c)) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT")}
    s2 = {c: sum(v) / len(v) for c, v in s2.items() if v}
    out["stalls"] = dict(s2, wait_any_frac=s2["SQ_WAIT_ANY"] / s2["SQ_WAVE_CYCLES"],
                         lds_conflict_per_active_lds=s2["SQ_LDS_BANK_CONFLICT"] / s2["SQ_ACTIVE_INST_LDS"])
except Exception as e:
    out["error_sq"] = repr(e)
json.dump(out, open("gpurun_out/pmc_raster_bwd.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
python bench.py --steps $STEPS --warmup $WARMUP > gpurun_out/bench_full.log 2>&1
grep '^{"metric"' gpurun_out/bench_full.log | tail -1 > gpurun_out/${TAG}_bench_line.json
cut -c1-400 gpurun_out/${TAG}_bench_line.json
