#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace -d /tmp/prof_ov -o bench -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > /tmp/ov.log 2>&1
tail -1 /tmp/ov.log | cut -c1-160
python scratch/overlap_analysis.py "$(find /tmp/prof_ov -name '*.db' | head -1)" 120
