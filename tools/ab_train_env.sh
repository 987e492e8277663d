#!/bin/bash
# GPU box: environment switches on the training-step bench (alternating)
mkdir -p gpurun_out; O=gpurun_out/ab_train_env.txt; : > $O
for rep in 1 2; do
for item in "$@"; do
  label=${item%%:*}; envs=${item#*:}; [ "$envs" = "$item" ] && envs="X=1"
  env $(echo $envs | tr ',' ' ') timeout 400 python bench.py --config train --no-cpu-baseline --steps 14 --warmup 3 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-30s %8.1f ambisonic-s/s trained  %.3f ms/step' % ('$label', d['value'], d['ms_per_step']))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
done; done
cat $O
