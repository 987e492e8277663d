#!/bin/bash
# GPU box: same-box A/B of the training leg, current build against tools/build_ab/libsagen_base.so (alternating runs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  timeout 300 python bench.py --config train --no-cpu-baseline --no-other-configs > gpurun_out/abt_new_$i.json 2>/dev/null
  SAGEN_LIB=$PWD/tools/build_ab/libsagen_base.so timeout 300 python bench.py --config train --no-cpu-baseline --no-other-configs > gpurun_out/abt_base_$i.json 2>/dev/null
done
python - <<'PY'
import json
for t in ('new','base'):
    for i in (1,2):
        d=json.loads(open('gpurun_out/abt_%s_%d.json'%(t,i)).read().strip().splitlines()[-1])
        r=d['roofline']
        print(t,i,d['value'],d['ms_per_step'],r['kernel'],r['avg_launch_us'],r['whole_step']['kernel_time_us_per_step'])
PY
