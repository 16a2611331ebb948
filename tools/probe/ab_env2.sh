#!/bin/bash
# A/B of environment switches of bench.py on one box: tools/probe/ab_env2.sh <runs> "<VAR=val ...>" "<VAR=val ...>" ...  (quote "" for the default)
# prints sequential / overlap frames per second of every run, alternating the variants
RUNS=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 $RUNS); do
  k=0
  for v in "$@"; do
    k=$((k+1))
    env $v python bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-oracle-psnr --no-other-configs --whole-run-frames 0 2> gpurun_out/abenv_err.log | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; j=json.loads(l); c=j['config']
print('variant $k [%s] run $i: sequential %.1f overlap %.1f' % ('$v', c['sequential_fps'], c['overlap_fps']))"
  done
done
