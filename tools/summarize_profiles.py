"""Digest gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.{csv,json,md}."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace('void sagen::', '').replace('sagen::', '')
    return name.split('(')[0].replace(' ', '')


# 1. kernel stats (rocprofv3 --kernel-trace --stats)
stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, tag + '_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'])
    rows = [r for r in rows if not r['Name'].startswith('Cijk_')]      # the host's stream-concurrency probe (torch.mm), not the path
    for r in rows[:45]:
        w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], '%.1f' % float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']])

# 1b. (training) the same trace with the backward on ONE stream: the launch durations bench.py's roofline object is computed from
#     (with two streams a weight gradient shares the chip with the data-gradient chain and its duration is not its own)
one = glob.glob(os.path.join(src, 'trace1', '*', '*_kernel_stats.csv'))
if one:
    rows1 = [r for r in csv.DictReader(open(one[0])) if not r['Name'].startswith('Cijk_')]
    with open(os.path.join(dst, tag + ('_kernel_stats_one_stream.csv' if any(r['Name'].find('wgrad') >= 0 for r in rows1) else '_kernel_stats_one_context.csv')), 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'])
        for r in rows1[:45]:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], '%.1f' % float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']])

# 2. PMC passes: per-kernel mean of each counter
pmc = collections.defaultdict(dict)
for sub in ('fetch', 'write', 'sq', 'sq2'):
    fs = glob.glob(os.path.join(src, sub, '*', '*_counter_collection.csv'))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            pmc[k][c] = sum(v) / len(v)
            pmc[k]['launches_' + sub] = len(v)
keep = {k: v for k, v in pmc.items() if 'igemm' in k or 'conv3p' in k or 'kernel' in k and not k.startswith('at::')}
json.dump(keep, open(os.path.join(dst, tag + '_pmc_per_launch.json'), 'w'), indent=1, sort_keys=True)

# 3. HBM traffic per launch for bench.py's roofline.traffic: FETCH_SIZE/WRITE_SIZE are in KiB-ish units of
#    64 B requests * 64 / 1024 (rocprofv3); the guide's gfx950 correction doubles FETCH_SIZE for wide
#    coalesced reads (MI355X_MICROARCH.md "HBM").  Reported both raw and corrected.
traffic = {}
for k, v in keep.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        raw = (v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0
        cor = (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0
        traffic[k] = {'hbm_bytes_raw': raw, 'hbm_bytes_fetch_x2': cor, 'fetch_kb': v['FETCH_SIZE'], 'write_kb': v['WRITE_SIZE']}
json.dump(traffic, open(os.path.join(dst, tag + '_traffic.json'), 'w'), indent=1, sort_keys=True)
# bench.py reads profiles/pmc_traffic.json keyed by the runtime's kernel label
label = {}
wsum = collections.defaultdict(lambda: [0.0, 0.0])
for k, v in traffic.items():
    if k.startswith(('igemm_kernel<', 'igemm3s2_kernel<', 'conv3p_kernel<', 'conv3h_kernel<', 'conv3hr_kernel<', 'conv3g_kernel<')) or k.startswith(('stem8pool_kernel', 'p3_pack_kernel')):
        label[k.replace(' ', '')] = round(v['hbm_bytes_fetch_x2'])
    elif k.startswith('igemm3_kernel<') or k.startswith('igemm3dw_kernel<'):
        # the runtime labels the bf16x3 kernels without their last template argument (batch-norm prologue flag):
        # launch-weighted mean over the two instantiations
        base = k.replace(' ', '').rsplit(',', 1)[0] + '>'
        n = keep[k].get('launches_fetch', 1)
        wsum[base][0] += v['hbm_bytes_fetch_x2'] * n
        wsum[base][1] += n
for base, (tot, n) in wsum.items():
    label[base] = round(tot / n)
# the training step labels the weight-gradient instantiations by family ("wgrad3r_kernel" / "wgrad3_kernel" / "wgrad_kernel"): launch-weighted mean
for pfx in ('wgrad3r_kernel', 'wgrad3_kernel', 'wgrad_kernel'):
    tot = n = 0.0
    for k, v in traffic.items():
        if k.startswith(pfx + '<'):
            m = keep[k].get('launches_fetch', 1)
            tot += v['hbm_bytes_fetch_x2'] * m; n += m
    if n:
        label[pfx] = round(tot / n)
is_train = any(k.startswith('wgrad') for k in traffic)
json.dump(label, open(os.path.join(dst, 'pmc_traffic_train.json' if is_train else 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
b = os.path.join(src, 'bench_under_trace.json')
if os.path.exists(b):
    open(os.path.join(dst, tag + '_bench_under_rocprof.json'), 'w').write(open(b).read())
# 4. (round 6, inference) per kernel of the one-context trace: duration of its own, HBM bytes per launch by the counters, GB/s against the 8 TB/s
#    peak (6.3 achievable), matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) - the same launch shapes in all passes
if one and not is_train:
    dur = {short(r['Name']): (float(r['AverageNs']), int(r['Calls'])) for r in rows1}
    with open(os.path.join(dst, tag + '_roofline_table.txt'), 'w') as f:
        f.write('# one grouped context alone (10 batches per launch); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)\n')
        f.write('%-46s %6s %10s %10s %9s %10s\n' % ('kernel', 'calls', 'avg us', 'MB/launch', 'GB/s', 'MFMA busy'))
        for k, (ns, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:28]:
            v = keep.get(k) or {}
            byts = (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0 if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v else None
            busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_BUSY_CU_CYCLES']) if v.get('SQ_BUSY_CU_CYCLES') else None
            f.write('%-46s %6d %10.1f %10s %9s %10s\n' % (k[:46], calls, ns / 1e3, '%.1f' % (byts / 1e6) if byts else '-',
                                                        '%.0f' % (byts / ns) if byts else '-', '%.0f %%' % (100 * busy) if busy is not None else '-'))
    print(open(os.path.join(dst, tag + '_roofline_table.txt')).read())
print(open(os.path.join(dst, tag + '_kernel_stats.csv')).read())
for k, v in sorted(keep.items()):
    if 'igemm' in k or 'conv3' in k or 'p3_' in k or 'stem8' in k or 'wgrad' in k or 'bwd' in k:
        print(k, {c: ('%.4g' % x) for c, x in v.items() if not c.startswith('launches')})
