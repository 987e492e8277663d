#!/bin/bash
# GPU box: the round's evidence in one call - profiles (inference + training), the default bench line with its legs, the one-stream
# per-layer tables, the workgroup-life trace of conv3h, the smoke entry.  Outputs under gpurun_out/; copy what is judged into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r05}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo smoke rc=$?
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo bench rc=$?
bash tools/collect_profiles.sh ${TAG} av > gpurun_out/${TAG}_collect.log 2>&1
bash tools/collect_profiles.sh ${TAG}_train train > gpurun_out/${TAG}_train_collect.log 2>&1
cd $GRAFT_REPO_ROOT
SAGEN_ONE_STREAM=1 python tools/profile_layers.py > gpurun_out/${TAG}_layers.txt 2>&1
python tools/train_profile.py > gpurun_out/${TAG}_train_layers.txt 2>&1
DUMP=gpurun_out SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_trace.so python tools/trace_conv3h.py 0 3 5 7 10 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_trace_conv3h.txt
tail -2 gpurun_out/${TAG}_smoke.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:d[k]['value'] for k in d if k.startswith('leg_')}, d['one_in_flight']['value'], d['float_frames']['value'], d['h2d_inclusive']['value'])
PY
ls gpurun_out/profiles_${TAG} gpurun_out/profiles_${TAG}_train
