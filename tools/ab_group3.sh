#!/bin/bash
# GPU box: knobs around the grouped default (10 batches per call): two-stream contexts, items per thread of the plane passes
mkdir -p gpurun_out; O=gpurun_out/ab_group3.txt; : > $O
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --steps 20 --warmup 5 $EXTRA 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('%-44s %8.1f ambisonic-s/s (repeats %s)' % ('$label', d['value'], r.get('values')))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
}
for i in 1 2; do
run "default (10 x 2, one stream per context)" X=1
run "two-stream contexts (10 x 2)" BENCH_TWO_STREAM_CONTEXTS=1
EXTRA="--in-flight 1" run "one two-stream context (10 x 1)" X=1
run "p3 amort 1" SAGEN_P3_AMORT=1
run "p3 amort 2" SAGEN_P3_AMORT=2
run "p3 amort 8" SAGEN_P3_AMORT=8
EXTRA="--group 20 --in-flight 1" run "20 x 1" X=1
EXTRA="--group 20 --in-flight 1" run "20 x 1 two-stream" BENCH_TWO_STREAM_CONTEXTS=1
done
cat $O
