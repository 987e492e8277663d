#!/bin/bash
# GPU box: grid amortisation of the backward's elementwise passes (SAGEN_BWD_AMORT items per thread) - training step time
for i in 1 2; do
  for a in "8 4" "16 4" "32 4" "8 8" "16 16"; do
    set -- $a
    SAGEN_BWD_AMORT=$1 SAGEN_P3_AMORT=$2 timeout 300 python bench.py --config train > gpurun_out/ab_bw$1_$2_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json
for a in ("8_4","16_4","32_4","8_8","16_16"):
    for i in (1,2):
        d=json.load(open('gpurun_out/ab_bw%s_%d.json'%(a,i)))
        print('bwd_p3 amort',a,i,d['value'],d['ms_per_step'])
PY
