#!/bin/bash
# GPU box: p3_pack grid amortisation A/B (SAGEN_P3_AMORT=1 = one item per thread, the round-4 grid) - headline + plane-pass time
for i in 1 2; do
  for a in 0 1 2 4; do
    SAGEN_P3_AMORT=$a timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_am${a}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json
for a in (0,1,2,4):
    for i in (1,2):
        d=json.load(open('gpurun_out/ab_am%d_%d.json'%(a,i)))
        r=d['roofline']
        print('amort',a,'(default by C)' if a==0 else '',i,d['value'],d['ms_per_step'],d['one_in_flight']['value'],r['whole_step']['kernel_time_us_per_step'],r['largest_hbm_bound_kernel']['us_per_step'])
PY
