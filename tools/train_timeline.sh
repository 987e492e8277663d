#!/bin/bash
# Dev helper (GPU box): the kernel timeline of ONE steady-state training step (start offset, duration, queue, kernel), the time each
# queue is busy and the time only one of them is - what the second stream hides and what it does not.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train_timeline; rm -rf $O; mkdir -p $O
python $R/bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --no-repeats --no-pmc --config train --steps 20 --warmup 5 --plan-file $O/plan.json > $O/tune.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --no-repeats --no-pmc --config train --plan-file $O/plan.json --steps 12 --warmup 3 > $O/trace.log 2>&1
python3 - > $O/timeline.txt <<PY
import csv, glob, collections
f = glob.glob('$O/trace/*/*_kernel_trace.csv')[0]
rd = list(csv.DictReader(open(f)))
qk = 'Stream_Id' if 'Stream_Id' in rd[0] else 'Queue_Id'
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void sagen::', '').replace('sagen::', '').split('(')[0][:58], r[qk]) for r in rd]
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('stft_kernel')]
print('steps seen:', len(starts), 'queue key:', qk)
a, b = starts[-4], starts[-3]
seg = rows[a:b]
t0 = seg[0][0]; t1 = rows[b][0]
print('step wall %.1f us, %d launches' % ((t1 - t0) / 1e3, len(seg)))
# busy intervals per queue
qs = sorted(set(r[3] for r in seg))
ev = []
for s, e, n, q in seg: ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
act = collections.Counter(); last = t0; only = collections.Counter(); both = 0; idle = 0
for t, d, q in ev:
    live = [k for k in qs if act[k] > 0]
    if len(live) == 0: idle += t - last
    elif len(live) == 1: only[live[0]] += t - last
    else: both += t - last
    act[q] += d; last = t
print('idle %.1f us, both queues %.1f us, ' % (idle / 1e3, both / 1e3) + ', '.join('only q%s %.1f us' % (k, v / 1e3) for k, v in only.items()))
for s, e, n, q in seg:
    print('%8.1f %7.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
head -5 $O/timeline.txt
rm -rf $O/trace
