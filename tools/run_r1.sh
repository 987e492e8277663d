cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "forced_plans and (70-1 or 71-2 or 72-1 or 58-2)" > gpurun_out/r1_parity.txt 2>&1; echo parity rc=$? 
tail -3 gpurun_out/r1_parity.txt
SAGEN_ONE_STREAM=1 timeout 300 python tools/profile_layers.py > gpurun_out/r1_layers_new.txt 2>&1
SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_prev.so timeout 300 python tools/profile_layers.py > gpurun_out/r1_layers_prev.txt 2>&1
grep -E "^plan video_encoder/conv[2345]_[12]/conv_[12]|^total|^conv3h" gpurun_out/r1_layers_new.txt
echo ---- prev
grep -E "^plan video_encoder/conv[2345]_[12]/conv_[12]|^total|^conv3h" gpurun_out/r1_layers_prev.txt
bash tools/ab_lib.sh 2>&1 | tail -6
