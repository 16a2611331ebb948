"""Are the frame chain's dispatch waits in the overlap schedule host-side or device-side?  From a rocprofv3 --kernel-trace
--hip-runtime-trace rocpd database: for every frame-stream kernel b that starts > thr us after the previous frame-stream
kernel a ended (tracker evaluations excluded as predecessors' successors: the host decides between them), the hipLaunchKernel
call that produced b (same correlation id): when it was entered and left, relative to a's end and b's start.
usage: launch_vs_start.py <db> [thr_us] [windows, default 2]"""
import sqlite3, sys, re
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
corr = "stack_id"   # a dispatch and the API call that made it share the event stack id
q = "select name, start, end, stream_id, queue_id, stack_id from kernels order by start"
rows = db.execute(q).fetchall()
marks = [r[1] for r in rows if "spin_kernel" in r[0]]
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lo, hi = (marks[0], marks[2 * nw - 1]) if len(marks) >= 2 * nw else (rows[0][1], rows[-1][2])   # the timed windows of the (only) schedule
sel = [r for r in rows if lo <= r[1] < hi and "spin_kernel" not in r[0]]
cnt = defaultdict(int)
for r in sel:
    if "track_eval" in r[0]: cnt[r[3]] += 1
fs = max(cnt, key=cnt.get)
queues = defaultdict(set)
for r in sel: queues[r[3]].add(r[4])
print("stream -> hardware queue ids:", {k: sorted(v) for k, v in queues.items()})
api = {}
if corr:
    for name, s, e, tid, c in db.execute("select name, start, end, tid, stack_id from regions where name like 'hipLaunchKernel%' or name like 'hipExtLaunch%' or name like 'hipModuleLaunch%'"):
        api[c] = (name, s, e, tid)
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:30]
frame = [r for r in sel if r[3] == fs]
n = 0; host_late = 0; dev_wait = 0; tot_host = 0.0; tot_dev = 0.0
for a, b in zip(frame, frame[1:]):
    if "track_eval" in b[0] and "track_eval" in a[0]: continue
    gap = (b[1] - a[2]) / 1e3
    if gap <= thr or gap > 400: continue
    n += 1
    l = api.get(b[5])
    if not l:
        if n <= 5: print("no launch record for", short(b[0]), b[5])
        continue
    enter, leave = (l[1] - a[2]) / 1e3, (l[2] - a[2]) / 1e3   # relative to a's end
    # host-side share: how long after a's end the launch call RETURNED (<= 0: it was queued before a ended); device-side: the rest
    h = max(0.0, min(gap, leave)); d = gap - h
    tot_host += h; tot_dev += d
    if h > d: host_late += 1
    else: dev_wait += 1
    if n <= 14:
        print("%-28s after %-28s gap %6.1f us: launch call entered %+7.1f, returned %+7.1f us after the predecessor ended (thread %s)" %
              (short(b[0]), short(a[0]), gap, enter, leave, l[3]))
print("%d waits: host-side share %.1f us, device-side share %.1f us in total; mostly host %d, mostly device %d" % (n, tot_host, tot_dev, host_late, dev_wait))

# ---- what do the OTHER threads do inside the runtime while a launch call of the frame thread takes long?
frame_tid = None
tids = defaultdict(int)
for b in frame:
    l = api.get(b[5])
    if l: tids[l[3]] += 1
frame_tid = max(tids, key=tids.get) if tids else None
allapi = db.execute("select name, start, end, tid from regions where start >= ? and start < ? order by start", (lo, hi)).fetchall()
long_calls = [r for r in allapi if r[3] == frame_tid and r[0].startswith("hipLaunchKernel") and r[2] - r[1] > 30000]
norm = [r[2] - r[1] for r in allapi if r[3] == frame_tid and r[0].startswith("hipLaunchKernel")]
norm.sort()
print("frame thread %s: %d launch calls, median %.1f us, %d longer than 30 us (%.1f us per frame in them)" %
      (frame_tid, len(norm), norm[len(norm) // 2] / 1e3, len(long_calls), sum(r[2] - r[1] for r in long_calls) / 1e3 / max(1, sum(1 for r in frame if "integrate_kernel" in r[0]))))
beside = defaultdict(lambda: [0, 0.0])
for r in long_calls:
    for o in allapi:
        if o[3] != frame_tid and o[1] < r[2] and o[2] > r[1]:
            ov = (min(o[2], r[2]) - max(o[1], r[1])) / 1e3
            beside[(o[0], o[3])][0] += 1; beside[(o[0], o[3])][1] += ov
for (name, tid), (n_, ov) in sorted(beside.items(), key=lambda kv: -kv[1][1])[:12]:
    print("   beside them on thread %s: %-34s %4d times, %8.1f us overlapping, its own durations: median %.1f us" %
          (tid, name, n_, ov, sorted(x[2] - x[1] for x in allapi if x[0] == name and x[3] == tid)[len([x for x in allapi if x[0] == name and x[3] == tid]) // 2] / 1e3))
