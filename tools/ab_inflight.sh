#!/bin/bash
# GPU box: the headline with 2 / 3 / 4 / 5 batches in flight on one box (alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do for n in 3 4 2 5; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-extra-legs --in-flight $n > gpurun_out/abf_${n}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json
for n in (2,3,4,5):
    for i in (1,2):
        d=json.loads(open('gpurun_out/abf_%d_%d.json'%(n,i)).read().strip().splitlines()[-1])
        r=d['roofline']
        print('in flight',n,i,d['value'],d['ms_per_step'],r['kernel'],r['avg_launch_us'],r['frac'],d['step_latency_ms_event']['median'])
PY
