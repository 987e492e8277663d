"""Dev helper (GPU box): A/B one layer's kernel choice inside a fixed launch plan.
usage: ab_plan.py <layer> <kernel name B> [splitk B]  - tunes once (plan A), then alternates bench runs A / B."""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
layer, kernel_b = sys.argv[1], sys.argv[2]
sk_b = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = os.path.join(ROOT, 'gpurun_out'); os.makedirs(out, exist_ok=True)
pa, pb = os.path.join(out, 'plan_a.json'), os.path.join(out, 'plan_b.json')
if os.path.exists(pa): os.remove(pa)
def bench(plan):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--plan-file', plan], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)
a0 = bench(pa)
P = json.load(open(pa))
for row in P['plan']:
    if row[0] == layer:
        print('plan A:', row); row[1] = kernel_b; row[2] = sk_b; print('plan B:', row)
json.dump(P, open(pb, 'w'))
for i in range(3):
    ra, rb = bench(pa), bench(pb)
    print('A %.1f ambisonic-s/s (%.3f ms)   B %.1f (%.3f ms)' % (ra['value'], ra['ms_per_step'], rb['value'], rb['ms_per_step']), flush=True)
