# Host-side duration of every processFrame call (GPS_BENCH_FRAME_TIMES) of the overlap schedule, for values of one environment
# variable: median of the frames of a keyframe period by position (0 = the keyframe).  bash tools/probe/frame_times.sh VAR "v1 v2"
VAR=$1; VALS=$2
mkdir -p gpurun_out
for v in $VALS; do
  for i in 1 2; do
  env $VAR=$v GPS_BENCH_FRAME_TIMES=1 python bench.py --steps 20 --warmup 5 --windows 5 --schedule ${SCHEDULE:-overlap} --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 2> gpurun_out/ft_${v}_$i.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$v run $i: value %.1f' % d['value'])"
  done
  python - <<PY
import re, statistics
pos = {}
for i in (1, 2):
    fr = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"frame (\d+): ([0-9.]+) ms", open("gpurun_out/ft_${v}_%d.err" % i).read())]
    fr = fr[-100:]   # the timed windows
    for f, t in fr:
        pos.setdefault(f % 10, []).append(t)
print("$VAR=$v  median ms by position in the period:", " ".join("%d:%.3f" % (k, statistics.median(v)) for k, v in sorted(pos.items())),
      " sum %.3f" % sum(statistics.median(v) for v in pos.values()))
PY
done
