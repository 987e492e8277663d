#!/bin/bash
# GPU box: two grouped contexts in lockstep (default) against the second trailing the first by one kernel (sagen_follow)
mkdir -p gpurun_out; O=gpurun_out/ab_follow.txt; : > $O
run() {  # label env steps extra...
  label=$1; e=$2; K=$3; shift 3
  env $e timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --steps $K --warmup 5 "$@" 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('%-34s K=%d %8.1f ambisonic-s/s (repeats %s)' % ('$label', $K, d['value'], r.get('values')))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
}
for rep in 1 2; do
run "default 10x2" X=1 20
run "follow 10x2" BENCH_FOLLOW=1 20
run "default 10x2" X=1 30
run "follow 10x2" BENCH_FOLLOW=1 30
run "default 5x2" X=1 20 --group 5
run "follow 5x2" BENCH_FOLLOW=1 20 --group 5
run "follow 5x3" BENCH_FOLLOW=1 30 --group 5 --in-flight 3
done
cat $O
