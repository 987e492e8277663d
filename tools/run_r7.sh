cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r7_parity.txt 2>&1; echo parity rc=$?
tail -3 gpurun_out/r7_parity.txt
SAGEN_ONE_STREAM=1 timeout 300 python tools/profile_layers.py > gpurun_out/r7_layers_new.txt 2>&1
SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_prev.so timeout 300 python tools/profile_layers.py > gpurun_out/r7_layers_prev.txt 2>&1
grep -E "^total|^conv3g" gpurun_out/r7_layers_new.txt
echo ---- prev
grep -E "^total|^conv3g" gpurun_out/r7_layers_prev.txt
bash tools/ab_lib.sh 2>&1 | tail -6
