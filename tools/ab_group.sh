#!/bin/bash
# GPU box: the grouped launch (bench.py --group G --in-flight NF) against three single-batch contexts in flight, same box, alternating.
mkdir -p gpurun_out; O=gpurun_out/ab_group.txt; : > $O
run() {  # group in-flight tag
  timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-pmc --group $1 --in-flight $2 --steps ${3:-30} 2>gpurun_out/ab_group_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('headline_repeats',{})
print('group %d x %d contexts: %8.1f ambisonic-s/s (repeats median %s min %s max %s)  one-in-flight %s  h2d %s  float %s  dom %s %.1f us frac %.3f' % ($1, $2, d['value'], r.get('median'), r.get('min'), r.get('max'), d['one_in_flight']['value'], d['h2d_inclusive']['value'], d['float_frames']['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" >> $O 2>&1 || tail -3 gpurun_out/ab_group_err.txt >> $O
}
run 1 3; run 3 2; run 3 3; run 1 3; run 2 3; run 4 2 32; run 3 2 20
cat $O
