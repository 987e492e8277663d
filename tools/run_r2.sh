cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_trace.so timeout 300 python tools/trace_conv3h.py 0 3 5 6 7 10 > gpurun_out/r2_trace.txt 2>&1
cat gpurun_out/r2_trace.txt | grep -v amdgpu.ids
