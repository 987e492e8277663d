#!/bin/bash
# GPU box: sweep of (batches per grouped launch, contexts in flight, steps) for bench.py's headline
mkdir -p gpurun_out; O=gpurun_out/ab_group2.txt; : > $O
run() {  # group in-flight steps
  timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --group $1 --in-flight $2 --steps $3 --warmup 5 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('group %2d x %d contexts, %d steps: %8.1f ambisonic-s/s (repeats %s)' % ($1, $2, $3, d['value'], r.get('values')))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
}
for K in 20 30; do
  run 1 3 $K; run 2 2 $K; run 4 2 $K; run 5 2 $K; run 5 3 $K; run 10 2 $K; run 10 1 $K; run 3 2 $K; run 6 2 $K
done
cat $O
