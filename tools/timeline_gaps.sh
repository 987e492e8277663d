#!/bin/bash
# Dev helper (GPU box): GPU busy fraction and the largest idle gaps of steady-state bench steps, from a kernel trace.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O
python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --plan-file $O/plan.json > $O/tune.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --plan-file $O/plan.json --steps 20 --warmup 5 > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$O/trace/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: the stft kernel marks the start of a forward; take forwards 10..18
starts = [i for i, r in enumerate(rows) if 'stft_kernel' in r[2] and 'istft' not in r[2]]
print('forwards seen:', len(starts))
tot_busy = tot_wall = 0
gaps = []
for a, b in zip(starts[10:18], starts[11:19]):
    seg = rows[a:b]
    t0 = seg[0][0]; t1 = rows[b][0]
    cur_s, cur_e = seg[0][0], seg[0][1]; busy = 0
    for s, e, n, q in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot_busy += busy; tot_wall += t1 - t0
print('per forward: wall %.1f us, union-busy %.1f us (%.1f%%), sum of kernel durations %.1f us' % (tot_wall / 8e3, tot_busy / 8e3, 100.0 * tot_busy / tot_wall, sum(e - s for s, e, n, q in rows[starts[10]:starts[18]]) / 8e3))
gaps.sort(reverse=True)
print('largest idle gaps (us, next kernel):', [(round(g / 1e3, 1), n) for g, n in gaps[:12]])
print('total idle per forward %.1f us in %d gaps' % (sum(g for g, n in gaps) / 8e3, len(gaps) // 8))
PY
