"""Dev helper: per-kernel mean of the counters collected by tools/pmc_conv.sh (igemm kernels only)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'igemm' not in k: continue
        acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v); print('   %-34s n=%3d median %.4g' % (c, len(v), v[len(v) // 2]))
