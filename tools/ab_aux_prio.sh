#!/bin/bash
# GPU box: priority of the context's second stream (SAGEN_AUX_PRIO=low|high|unset) - training step and headline
for i in 1 2; do
  for a in none low high; do
    if [ $a = none ]; then unset SAGEN_AUX_PRIO; else export SAGEN_AUX_PRIO=$a; fi
    timeout 300 python bench.py --config train > gpurun_out/ab_pr_train_${a}_$i.json 2>/dev/null
    timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_pr_av_${a}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json
for c in ('train','av'):
  for a in ('none','low','high'):
    for i in (1,2):
        d=json.load(open('gpurun_out/ab_pr_%s_%s_%d.json'%(c,a,i)))
        print(c,'aux prio',a,i,d['value'],d['ms_per_step'])
PY
