#!/bin/bash
# Dev helper (GPU box): true kernel durations of the grid sweep from rocprofv3's kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_sweep_$1; rm -rf $O; mkdir -p $O
SAGEN_FORCE_TILE=${2:-2} rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/tools/bench_conv_sweep.py ${3:-64} > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob('$O/*/*_kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'igemm' in r['Kernel_Name']:
        agg[int(r['Grid_Size'])//256].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(agg.items()):
    v=sorted(v); print('blocks=%5d  n=%2d  median %.1f us  min %.1f' % (g, len(v), v[len(v)//2], v[0]))
PY
