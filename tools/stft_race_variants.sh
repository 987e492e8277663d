#!/bin/bash
# Dev helper (GPU box): which part / codegen choice of igemm3_kernel disturbs a concurrently running stft_kernel?
R=$GRAFT_REPO_ROOT; B=$R/spatialaudiogen_amd/csrc/build; S=$R/spatialaudiogen_amd/csrc
OTHERS="$B/igemm3dw.o $B/igemm3s2.o $B/igemm.o $B/elementwise.o $B/fft.o $B/eval.o $B/model.o $B/api.o"
run() {
  tag=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $S/igemm3.hip -o /tmp/igemm3_$tag.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm3_$tag.o $OTHERS -o /tmp/libsagen_$tag.so || exit 1
  echo "== variant $tag ($*)"
  SAGEN_LIB=/tmp/libsagen_$tag.so SAGEN_FORCE_TILE=${TILE:-22} python $R/tools/stft_race.py 2>&1 | grep "vs conv3x3"
}
run full
run agpr_acc -mllvm -amdgpu-mfma-vgpr-form=0
run vgpr_acc -mllvm -amdgpu-mfma-vgpr-form=1
run no_mfma -DSAGEN_ABLATE_MFMA
