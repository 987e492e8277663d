#!/bin/bash
# 1 / 2 / 4 / 8-GPU sweep of bench.py on ONE node (one rank per GPU over RCCL) and a table of the lines.
#   tools/scale_sweep.sh [config=av] [steps=30] [warmup=5]          (configs: av a avf eval train)
# Needs as many GPUs as the largest N it runs (it stops at what `rocm-smi` / torch sees).  Output: gpurun_out/scale_<config>.jsonl (the
# bench lines) and gpurun_out/SCALE_<config>.json - ONE JSON object in the shape of the driver's SCALE record: metric / unit / config
# and per N {n_gpus, value, ms_per_step, ranks_seen, backend, efficiency_vs_n1} - also printed as the last line of stdout.
set -u
cd "$(dirname "$0")/.."
CFG=${1:-av}; STEPS=${2:-30}; WARM=${3:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
mkdir -p gpurun_out
OUT=gpurun_out/scale_${CFG}.jsonl
: > "$OUT"
PLAN=gpurun_out/scale_plan_${CFG}.json          # tuned once at N=1, replayed by every N: all runs execute the same kernels
for N in 1 2 4 8; do
    [ "$N" -gt "$NGPU" ] && { echo "skip N=$N (only $NGPU GPUs)"; continue; }
    if [ "$N" -eq 1 ]; then
        python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --config "$CFG" --plan-file "$PLAN" --no-cpu-baseline | tail -1 >> "$OUT"
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
            bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --config "$CFG" --plan-file "$PLAN" | tail -1 >> "$OUT"
    fi
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
if not rows:
    sys.exit('no bench lines')
base = rows[0]['value'] / rows[0]['n_gpus']
print('%4s %12s %10s %8s %10s %24s %s' % ('N', 'value', 'ms/step', 'eff', 'ranks', 'rank ms/step min..max', 'backend'))
for r in rows:
    rk = r.get('ranks') or {}
    mm = rk.get('rank_ms_per_step', {})
    print('%4d %12.1f %10.3f %8.3f %10s %24s %s' % (r['n_gpus'], r['value'], r['ms_per_step'], r['value'] / (base * r['n_gpus']),
          rk.get('ranks_seen'), '%s..%s' % (mm.get('min'), mm.get('max')), rk.get('backend')))
    x = r.get('gradient_exchange_timing')
    if x:
        print('       gradient exchange: step %.2f ms, comm stream waiting for gradients %.2f ms, in all-reduce %.2f ms, last all-reduce done at %.2f ms'
              % (x['step_ms'], x['waiting_for_gradients_ms'], x['in_all_reduce_ms'], x['last_all_reduce_done_ms']))
scale = {'metric': rows[0]['metric'], 'unit': rows[0]['unit'], 'scaling': rows[0].get('scaling'), 'config': rows[0].get('config'),
         'steps': rows[0]['steps'], 'warmup': rows[0]['warmup'],
         'points': [{'n_gpus': r['n_gpus'], 'value': r['value'], 'ms_per_step': r['ms_per_step'],
                     'ranks_seen': (r.get('ranks') or {}).get('ranks_seen'), 'backend': (r.get('ranks') or {}).get('backend'),
                     'efficiency_vs_n1': r['value'] / (base * r['n_gpus'])} for r in rows]}
out = sys.argv[1].replace('scale_', 'SCALE_').replace('.jsonl', '.json')
json.dump(scale, open(out, 'w'), indent=1)
print(json.dumps(scale))
PY
