#!/bin/bash
# GPU box: what would G batches of 32 in ONE launch per layer cost?  Upper bound by running the forward at batch 64 / 96 / 128 as ONE
# batch (same kernel shapes and launch counts as a grouped launch; only the batch-norm statistics differ) with 1 / 2 / 3 contexts
# in flight, against the headline configuration (batch 32, three in flight).  Output: gpurun_out/probe_group.txt
mkdir -p gpurun_out
O=gpurun_out/probe_group.txt; : > $O
run() {  # batch in-flight
  BENCH_PROBE_BATCH=$1 timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --no-extra-legs --in-flight $2 --steps ${3:-20} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch %3d in-flight %d: %8.1f ambisonic-s/s  %.3f ms/step  kernel-sum %.0f us  dom %s %.1f us' % ($1, $2, d['value'], d['ms_per_step'], d['roofline']['whole_step']['kernel_time_us_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us']))" >> $O
}
run 32 3 30; run 32 1 30
run 64 1; run 64 2
run 96 1; run 96 2; run 96 3
run 128 1; run 128 2
cat $O
