#!/bin/bash
# Dev helper (GPU box): per-kernel times of the ResNet 3x3 layer shapes under the pre-split-planes kernel (conv3p.hip)
# and, for reference, the shared-tap kernel (SAGEN_NO_P3=1).  usage: tools/p3_sweep.sh <tag> [tile-name-substrings...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; shift
O=$R/gpurun_out/p3_$TAG; rm -rf $O; mkdir -p $O
run() {   # label, env...
    local label=$1; shift
    for c in s2 s3 s4 s5; do
        env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_${label}_$c -- python $R/tools/bench_conv.py $c 10 > $O/${label}_$c.log 2>&1
        f=$(find $O/t_${label}_$c -name "*kernel_stats.csv" | head -1)
        echo "== $label $c" >> $O/summary.txt
        [ -n "$f" ] && python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    n = r['Name'].replace('void sagen::', '').split('(')[0]
    print('   %-58s calls %3s avg %9.1f us  min %9.1f' % (n[:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
        rm -rf $O/t_${label}_$c
    done
}
run p3default SAGEN_X=1
run nop3 SAGEN_NO_P3=1
for t in "$@"; do
    id=$(python3 -c "
import sys; sys.path.insert(0, '$R')
from spatialaudiogen_amd.model import SptAudioGen
print([i for i, n in enumerate(SptAudioGen.tile_names()) if n == '$t'][0])")
    run "force_$id" SAGEN_FORCE_TILE=$id
    echo "   (force_$id = $t)" >> $O/summary.txt
done
cat $O/summary.txt
