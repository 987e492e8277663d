#!/bin/bash
# PMC of the ubench kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ub; rm -rf $O; mkdir -p $O
T="conv3p_kernel<128,64,64,32>"; S=32,56,112,64,64
for F in "-DP3_ABLATE_MFMA" "-DP3_ABLATE_DMA"; do timeout 300 python $R/tools/ubench_p3.py --flags "$F" "$T" $S 32,28,56,128,128 2>&1 | grep conv3p; timeout 200 python $R/tools/ubench_p3.py --trace --flags "$F" "$T" $S 2>&1 | grep persistent; done
python $R/tools/ubench_p3.py "$T" $S > /dev/null 2>&1   # build cache
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/ubench_p3.py "$T" $S > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- python $R/tools/ubench_p3.py "$T" $S > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p3 -- python $R/tools/ubench_p3.py "$T" $S > $O/p3.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p?/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3p' not in k: continue
        acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print('PMC', k)
    for n, v in sorted(cs.items()):
        v = sorted(v); print('   %-30s n=%3d median %.5g' % (n, len(v), v[len(v) // 2]))
PY
rm -rf $O
