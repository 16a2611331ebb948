# PMC of one rasterizer kernel alone (KERNEL=raster_ges_fwd_pk_kernel | raster_ges_bwd_gs_kernel; tools/raster_bench.py: 50 back-to-back launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/p1 -o a -- python $R/tools/raster_bench.py > /tmp/l1 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d /tmp/p2 -o a -- python $R/tools/raster_bench.py > /tmp/l2 2>&1
grep "raster bwd" /tmp/l1 /tmp/l2
python - <<'PY'
import glob, sqlite3
import os
K = os.environ.get("KERNEL", "raster_ges_bwd_gs_kernel")
for d, cs in (("p1", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")), ("p2", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"))):
    db = sqlite3.connect(glob.glob("/tmp/%s/**/*.db" % d, recursive=True)[0])
    for c in cs:
        v = [r[0] for r in db.execute("select c.value from counters_collection c where c.kernel_name like ? and c.counter_name=? order by c.dispatch_id", ("%" + K + "%", c))]
        v = v[-60:-10] if len(v) > 70 else v
        print(c, len(v), sum(v) / max(1, len(v)))
PY
