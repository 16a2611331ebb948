# N repetitions of the DRIVER's bench command on the tree's build: value, both schedules, slow windows, anything on stderr
N=${1:-10}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs > gpurun_out/rep3_$i.log 2> gpurun_out/rep3_err_$i.log
  echo "run $i rc $?"
done
python - <<'PY'
import json, glob
vals = []
for f in sorted(glob.glob('gpurun_out/rep3_[0-9]*.log'), key=lambda s: int(s.split('_')[-1].split('.')[0])):
    l = [x for x in open(f) if x.startswith('{')]
    if not l: print(f, 'NO LINE'); continue
    j = json.loads(l[-1]); c = j['config']
    vals.append(j['value'])
    print("%-26s overlap %.1f sequential %.1f" % (f, j['value'], c['schedules']['sequential']['frames_per_s']), [round(x, 3) for x in c['windows_ms_per_step']])
import statistics
print("median %.1f min %.1f max %.1f" % (statistics.median(vals), min(vals), max(vals)))
PY
cat gpurun_out/rep3_err_*.log | grep -v "^$" | sort | uniq -c | sort -rn | head -10
