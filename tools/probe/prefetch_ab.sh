# Kernel-level A/B of the cross-iteration fusion (GPS_BENCH_PREFETCH=0/1): rocprofv3 kernel trace of the sequential schedule, the
# preprocessing / rasterizer rows of both runs (run ON the GPU box: bash tools/probe/prefetch_ab.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  rm -rf /tmp/pf_$v
  GPS_BENCH_PREFETCH=$v rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -o t -- python bench.py --steps 20 --warmup 5 --windows 2 --schedule sequential --no-cpu-baseline --no-oracle-psnr --no-other-configs > /tmp/pf_$v.log 2>&1
  echo "== prefetch $v"; grep '^{"metric"' /tmp/pf_$v.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])"
  python tools/prof_summary.py "$(find /tmp/pf_$v -name '*.db' | head -1)" 12 --frames 20 --windows 2 2>/dev/null | grep -i "preprocess\|raster_ges_fwd\|sb_sc\|total kernel" | head -12
done
