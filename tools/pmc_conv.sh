#!/bin/bash
# Dev helper (GPU box): SQ counters of one conv layer. usage: tools/pmc_conv.sh <case> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CASE=${1:-s2}; TAG=${2:-a}
O=$R/gpurun_out/pmc_${CASE}_${TAG}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/bench_conv.py $CASE 10 > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/bench_conv.py $CASE 5 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/p2 -- python $R/tools/bench_conv.py $CASE 5 > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p3 -- python $R/tools/bench_conv.py $CASE 5 > $O/p3.log 2>&1
python3 $R/tools/pmc_digest.py $O
find $O -name "*.csv" | head -20
grep -h "median" $O/*.log
