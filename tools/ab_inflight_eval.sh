#!/bin/bash
# GPU box: the eval stand-in (configs[3], batches of 16) and the audio+video+flow configuration with 3 / 4 / 6 batches in flight
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in eval avf; do for n in 3 4 6; do
  timeout 300 python bench.py --config $cfg --no-other-configs --no-cpu-baseline --no-extra-legs --in-flight $n > gpurun_out/abe_${cfg}_${n}.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/abe_${cfg}_${n}.json').read().strip().splitlines()[-1])
print('${cfg}','in flight',${n},d['value'],d['ms_per_step'])
PY
done; done
