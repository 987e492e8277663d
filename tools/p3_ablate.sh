#!/bin/bash
# Dev helper (GPU box): conv3p_kernel time per layer shape under the variant libraries of tools/ab_p3_variants.sh, plus
# SQ counters of the shipped kernel.  usage: tools/p3_ablate.sh <tag> <variant...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; shift
O=$R/gpurun_out/p3abl_$TAG; rm -rf $O; mkdir -p $O
CASES=${CASES:-"s2 s3"}
one() {   # label, env...
    local label=$1; shift
    for c in $CASES; do
        env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python $R/tools/bench_conv.py $c 10 > $O/${label}_$c.log 2>&1
        f=$(find $O/t -name "*kernel_stats.csv" | head -1)
        [ -n "$f" ] && python3 - "$f" "$label $c" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv3p' in r['Name'] or 'igemm3dw' in r['Name']:
        print('%-24s %-44s avg %8.1f us  min %8.1f' % (sys.argv[2], r['Name'].replace('void sagen::', '').split('(')[0][:44], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
        rm -rf $O/t
    done
}
one shipped SAGEN_X=1
for v in "$@"; do one $v SAGEN_LIB=$R/tools/build_ab/libsagen_$v.so; done
if [ -z "$NO_PMC" ]; then
for c in $CASES; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p1_$c -- python $R/tools/bench_conv.py $c 5 > $O/p1_$c.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/p2_$c -- python $R/tools/bench_conv.py $c 5 > $O/p2_$c.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p3_$c -- python $R/tools/bench_conv.py $c 5 > $O/p3_$c.log 2>&1
  python3 - $O $c >> $O/summary.txt <<'PY'
import csv, glob, sys, collections
root, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p?_%s/**/*counter_collection.csv' % c, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3p' not in k: continue
        acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print('PMC', c, k)
    for n, v in sorted(cs.items()):
        v = sorted(v); print('   %-30s n=%3d median %.5g' % (n, len(v), v[len(v) // 2]))
PY
  rm -rf $O/p1_$c $O/p2_$c $O/p3_$c
done
fi
cat $O/summary.txt
