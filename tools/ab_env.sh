#!/bin/bash
# GPU box: A/B of environment switches on the default headline: tools/ab_env.sh "LABEL:VAR=VAL[,VAR=VAL]" ...   (LABEL alone = default)
mkdir -p gpurun_out; O=gpurun_out/ab_env.txt; : > $O
for rep in 1 2; do
for item in "$@"; do
  label=${item%%:*}; envs=${item#*:}; [ "$envs" = "$item" ] && envs="X=1"
  env $(echo $envs | tr ',' ' ') timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --steps 20 --warmup 5 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('%-40s %8.1f ambisonic-s/s (repeats %s) dom %s %.1f us' % ('$label', d['value'], r.get('values'), d['roofline']['kernel'], d['roofline']['avg_launch_us']))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
done; done
cat $O
