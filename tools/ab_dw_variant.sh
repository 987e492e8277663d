#!/bin/bash
# Dev helper (GPU box): A/B a compile-time variant of igemm3dw.hip against the shipped library, same box, same plan.
# usage: ab_dw_variant.sh "<extra hipcc flags>"
R=$GRAFT_REPO_ROOT; B=$R/spatialaudiogen_amd/csrc/build; S=$R/spatialaudiogen_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize"
/opt/rocm/bin/hipcc $FLAGS $1 -c $S/igemm3dw.hip -o /tmp/igemm3dw_var.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm3dw_var.o $B/igemm3s2.o $B/igemm.o $B/igemm3.o $B/elementwise.o $B/fft.o $B/eval.o $B/model.o $B/api.o -o /tmp/libsagen_var.so || exit 1
cd $R
python bench.py --no-cpu-baseline --in-flight 1 --steps 2 --warmup 1 --plan-file /tmp/plan_ab.json > /dev/null 2>&1
for i in 1 2 3; do
  echo "base    $(python bench.py --no-cpu-baseline --in-flight 1 --plan-file /tmp/plan_ab.json 2>&1 | tail -1 | cut -c80-110)"
  echo "variant $(SAGEN_LIB=/tmp/libsagen_var.so python bench.py --no-cpu-baseline --in-flight 1 --plan-file /tmp/plan_ab.json 2>&1 | tail -1 | cut -c80-110)"
done
