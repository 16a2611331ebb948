#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace -d /tmp/prof_gap -o bench -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > /tmp/gap.log 2>&1
python scratch/gap_analysis.py "$(find /tmp/prof_gap -name '*.db' | head -1)" 0.6 120
