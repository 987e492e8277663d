#!/bin/bash
# Dev helper (GPU box): SQ counters of one kernel family inside the training step. usage: tools/pmc_train_kernel.sh <name substring> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; PAT=${1:-wgrad3h}; TAG=${2:-a}
O=/tmp/pmc_${TAG}; rm -rf $O; mkdir -p $O
CMD="python $R/tools/train_profile.py audio+video --steps-only"
export SAGEN_BWD_ONE_STREAM=1 STEPS_ONLY=3
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p5 -- $CMD > $O/p5.log 2>&1
python3 - "$O" "$PAT" <<'PY' | tee $R/gpurun_out/pmc_train_${PAT}_${TAG}.txt
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if pat not in k: continue
        acc[k.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v); print('   %-34s n=%3d median %.5g  min %.5g max %.5g' % (c, len(v), v[len(v) // 2], v[0], v[-1]))
PY
tail -3 $O/p1.log
