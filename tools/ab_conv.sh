#!/bin/bash
# Dev helper (GPU box): A/B the conv microbench under env knobs
cd $GRAFT_REPO_ROOT
for c in s2 s3 s4 s5; do
  echo "== $c default"; python tools/bench_conv.py $c 20 2>&1 | grep median
  echo "== $c no_prio"; SAGEN_NO_PRIO=1 python tools/bench_conv.py $c 20 2>&1 | grep median
  for t in 0 1 2 3; do echo "== $c tile $t"; SAGEN_FORCE_TILE=$t python tools/bench_conv.py $c 20 2>&1 | grep median; done
done
