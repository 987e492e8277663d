cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "forced_plans and (55- or 56- or 57- or 58- or 63- or 64- or 65- or 70- or 71- or 72-) or conv5_2 or torch_ref or fp16x2" > gpurun_out/r6_parity.txt 2>&1; echo parity rc=$?
tail -3 gpurun_out/r6_parity.txt
TUNE=0 FORCE=58:video_encoder SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_trace.so timeout 300 python tools/trace_conv3h.py 3 6 2>&1 | grep -v amdgpu.ids | grep -E "^launch|cycles p10|first round|later" > gpurun_out/r6_trace.txt
cat gpurun_out/r6_trace.txt
SAGEN_ONE_STREAM=1 timeout 300 python tools/profile_layers.py > gpurun_out/r6_layers_new.txt 2>&1
SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_prev.so timeout 300 python tools/profile_layers.py > gpurun_out/r6_layers_prev.txt 2>&1
grep -E "^total|^conv3h" gpurun_out/r6_layers_new.txt
echo ---- prev
grep -E "^total|^conv3h" gpurun_out/r6_layers_prev.txt
bash tools/ab_lib.sh 2>&1 | tail -6
