#!/bin/bash
# GPU box: the training leg under SAGEN_BWD_AMORT = 4 (default) / 2 / 1, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do for a in 4 1 2; do
  SAGEN_BWD_AMORT=$a timeout 300 python bench.py --config train --no-cpu-baseline --no-other-configs > gpurun_out/abm_${a}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json
for a in (4,2,1):
    for i in (1,2):
        d=json.loads(open('gpurun_out/abm_%d_%d.json'%(a,i)).read().strip().splitlines()[-1])
        r=d['roofline']
        print('amort',a,i,d['value'],d['ms_per_step'],r['kernel'],r['avg_launch_us'],r['whole_step']['kernel_time_us_per_step'])
PY
