#!/bin/bash
# Dev helper (GPU box): A/B a compile-time variant of wgrad.hip against the shipped library on the same box (one-stream train profile).
# usage: ab_wgrad.sh "<extra hipcc flags>"
R=$GRAFT_REPO_ROOT; B=$R/spatialaudiogen_amd/csrc/build; S=$R/spatialaudiogen_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DSAGEN_BUILD_FLAGS="\"$FLAGS\"" $1 -c $S/wgrad.hip -o /tmp/wgrad_var.o || exit 1
OBJS=$(ls $B/*.o | grep -v wgrad.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/wgrad_var.o $OBJS -o /tmp/libsagen_var.so || exit 1
cd $R
for i in 1 2; do
  echo "base    $(SAGEN_BWD_ONE_STREAM=1 python tools/train_profile.py audio+video 2>&1 | grep -E '^wgrad3_kernel|forward_backward' | tr '\n' ' ')"
  echo "variant $(SAGEN_LIB=/tmp/libsagen_var.so SAGEN_BWD_ONE_STREAM=1 python tools/train_profile.py audio+video 2>&1 | grep -E '^wgrad3_kernel|forward_backward' | tr '\n' ' ')"
done
