cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in trace trace_no_STORE trace_no_ATOMICS trace_no_BOTH; do
  echo "=== $v"
  TUNE=0 FORCE=58:video_encoder SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_$v.so timeout 300 python tools/trace_conv3h.py 3 6 2>&1 | grep -v amdgpu.ids | grep -E "^launch|cycles p10|first round|later"
done > gpurun_out/r4_trace.txt 2>&1
cat gpurun_out/r4_trace.txt
