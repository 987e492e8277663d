#!/bin/bash
# GPU box: A/B of bench.py argument sets on the headline: tools/ab_args.sh "LABEL:args..." ...
mkdir -p gpurun_out; O=gpurun_out/ab_args.txt; : > $O
for rep in 1 2; do
for item in "$@"; do
  label=${item%%:*}; a=${item#*:}
  timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs $a 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('%-34s %8.1f ambisonic-s/s (repeats %s)' % ('$label', d['value'], r.get('values')))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
done; done
cat $O
