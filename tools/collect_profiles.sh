#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py (kernel trace + separate PMC passes; never combined with
# sys/hip traces).  Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py digests them.
TAG=${1:-r01}
CFG=${2:-av}          # bench configuration: av (headline) | train | ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
# tune once OUTSIDE the profiler, then replay the saved plan so the traces hold steady-state launches only
python $R/bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --no-repeats --no-pmc --config $CFG --steps 20 --warmup 5 --plan-file $O/plan.json > $O/tune.log 2>&1      # (tuned on the launch shapes the passes below replay: 20 steps = ten batches per call)
BENCH="python $R/bench.py --no-cpu-baseline --no-other-configs --config $CFG --plan-file $O/plan.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH --steps 20 --warmup 5 > $O/trace.log 2>&1
if [ "$CFG" = train ]; then
  SAGEN_BWD_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- $BENCH --steps 20 --warmup 5 > $O/trace1.log 2>&1
  PMC="$BENCH --steps 2 --warmup 1"
else
  # round 6 (grouped launches): a second trace with ONE context in flight (a launch's duration is then its own), and the counter passes on the
  # SAME launch shapes - ten batches per call - so that bytes / duration and MFMA-busy / duration are statements about one launch
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- $BENCH --group 10 --in-flight 1 --steps 20 --warmup 5 > $O/trace1.log 2>&1
  PMC="$BENCH --group 10 --in-flight 1 --steps 10 --warmup 1"
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $PMC > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $PMC > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $PMC > $O/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $O/sq2 -- $PMC > $O/sq2.log 2>&1
grep -h '"metric"' $O/trace.log | tail -1 > $O/bench_under_trace.json
# digest on the box, ship only the summaries (raw traces exceed gpurun's copy-back limit)
python $R/tools/summarize_profiles.py $TAG $R/gpurun_out/profiles_$TAG > $O/summary.txt 2>&1
cp $O/summary.txt $R/gpurun_out/profiles_$TAG/${TAG}_summary.txt
cp $O/plan.json $R/gpurun_out/profiles_$TAG/${TAG}_plan.json
rm -rf $O
ls $R/gpurun_out/profiles_$TAG
