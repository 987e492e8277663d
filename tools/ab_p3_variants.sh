#!/bin/bash
# Dev helper: build compile-time variants of conv3p.hip (ablations / schedule experiments) as separate libraries next to
# the shipped one (run HERE, no GPU needed): tools/build_ab/libsagen_<name>.so ; on the GPU box select with SAGEN_LIB.
R=$(cd $(dirname $0)/.. && pwd); B=$R/spatialaudiogen_amd/csrc/build; S=$R/spatialaudiogen_amd/csrc; O=$R/tools/build_ab
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize"
OBJS=$(ls $B/*.o | grep -v conv3p.o)
build() {  # name, flags (variant libraries keep the production flag string so _lib accepts them)
    /opt/rocm/bin/hipcc $FLAGS $2 -c $S/conv3p.hip -o $O/conv3p_$1.o 2>/dev/null || { echo "compile $1 failed"; return 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/conv3p_$1.o $OBJS -o $O/libsagen_$1.so || { echo "link $1 failed"; return 1; }
    rm -f $O/conv3p_$1.o; echo "built $O/libsagen_$1.so"
}
while [ $# -gt 1 ]; do build "$1" "$2"; shift 2; done
