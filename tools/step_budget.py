"""Dev helper: per-kernel busy time and idle gaps of the LAST steps of a rocprofv3 kernel trace of bench.py
(steps delimited by stft_kernel launches)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if 'stft_kernel' in r[2]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lo = starts[-nsteps - 4]; hi = starts[-4]          # skip the 3 profiled forwards + tail
sel = rows[lo:hi]
busy = collections.defaultdict(float); cnt = collections.Counter()
gap = 0.0; prev_end = sel[0][0]
for s, e, n in sel:
    k = n.replace('void sagen::', '').replace('sagen::', '').split('(')[0]
    k = k if len(k) < 60 else k[:60]
    busy[k] += (e - s) / 1e3; cnt[k] += 1
    if s > prev_end: gap += (s - prev_end) / 1e3
    prev_end = max(prev_end, e)
span = (sel[-1][1] - sel[0][0]) / 1e3
tot = sum(busy.values())
print('steps %d  span/step %.1f us  busy/step %.1f us  idle gaps/step %.1f us  launches/step %.1f' % (nsteps, span / nsteps, tot / nsteps, gap / nsteps, len(sel) / nsteps))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print('%-62s n=%5.1f  %8.1f us/step  %5.1f%%  avg %.1f us' % (k, cnt[k] / nsteps, v / nsteps, 100 * v / tot, v / cnt[k]))
