#!/bin/bash
# GPU box: the audio-only leg (batch 10, grouped) with / without the decoder's planes at small batch
mkdir -p gpurun_out; O=gpurun_out/ab_leg_a.txt; : > $O
for rep in 1 2; do
for e in X=1 BENCH_NO_SMALL_BATCH_DECODER_PLANES=1; do
  env $e timeout 300 python bench.py --config a --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs --no-other-configs --no-pmc 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-44s %9.1f ambisonic-s/s %s' % ('$e', d['value'], d.get('headline_repeats',{}).get('values')))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
done; done
cat $O
