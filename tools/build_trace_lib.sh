#!/bin/bash
# Build container: tools/build_ab/libsagen_trace.so = the current objects with conv3h.hip recompiled under -DSAGEN_TRACE
# (per-workgroup life stamps, tools/trace_conv3h.py).  Run after `python -m spatialaudiogen_amd.build`.
set -e
cd "$(dirname "$0")/.."
C=spatialaudiogen_amd/csrc
mkdir -p tools/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fno-vectorize \
  -DSAGEN_BUILD_FLAGS='"trace"' -DSAGEN_TRACE $EXTRA_FLAGS -c $C/conv3h.hip -o tools/build_ab/conv3h_trace.o
OBJS=$(ls $C/build/*.o | grep -v '/conv3h.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/build_ab/conv3h_trace.o -o tools/build_ab/libsagen_trace.so
ls -la tools/build_ab/libsagen_trace.so
