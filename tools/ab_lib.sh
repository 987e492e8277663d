#!/bin/bash
# GPU box: same-box A/B of the current build against tools/build_ab/libsagen_prev.so (alternating runs of the headline bench)
for i in 1 2; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_new_$i.json 2>/dev/null
  SAGEN_LIB=$PWD/tools/build_ab/libsagen_prev.so timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_prev_$i.json 2>/dev/null
done
python - <<'PY'
import json
for t in ('new','prev'):
    for i in (1,2):
        d=json.load(open('gpurun_out/ab_%s_%d.json'%(t,i)))
        r=d['roofline']
        print(t,i,d['value'],d['ms_per_step'],d['one_in_flight']['value'],r['whole_step']['kernel_time_us_per_step'],r['kernel'],r['avg_launch_us'],r['frac'],r['contraction_time_split'])
PY
