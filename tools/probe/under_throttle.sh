#!/bin/bash
# The whole-sequence run while ANOTHER process of the container burns the CPU quota (64 busy threads against a quota of 16): the
# kernel freezes every thread of the container for most of every 100 ms accounting period.  Expected: a slow run with holes, no
# error, no "neither answered nor retired" and no 2 s hole (the evaluation's time-outs outlast a frozen host; LABBOOK section 14).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
grep thrott /sys/fs/cgroup/cpu.stat
python - <<'PY' &
import time, torch
torch.set_num_threads(64)
x = torch.randn(2048, 2048)
t0 = time.time()
while time.time() - t0 < 45: x @ x
PY
BURN=$!
sleep 2
timeout 300 python tools/whole_run_trace.py 1000 2> gpurun_out/under_throttle.err | cut -c1-400
kill $BURN 2>/dev/null; wait $BURN 2>/dev/null
grep thrott /sys/fs/cgroup/cpu.stat
echo "tracker diagnostics:"; grep "gps_slam_hip" gpurun_out/under_throttle.err | sed 's/evaluation [0-9]*/evaluation N/' | cut -c1-160 | sort | uniq -c | sort -rn | head
