#!/bin/bash
# GPU box: groups per tuning candidate (SAGEN_TUNE_GROUPS) - wall time of the bench process and the headline it reaches
mkdir -p gpurun_out; O=gpurun_out/ab_tune_groups.txt; : > $O
for rep in 1 2; do
for tg in 4 0 2; do
  t0=$(date +%s.%N)
  SAGEN_TUNE_GROUPS=$tg timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --steps 30 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('tune groups %-3s %8.1f ambisonic-s/s %s' % ('$tg', d['value'], d['headline_repeats']['values']), end=' ')" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
  t1=$(date +%s.%N); echo " wall $(python -c "print(round($t1-$t0,1))") s" >> $O
done; done
cat $O
