"""Dev helper: per-launch table of one forward (HIP events inside the runtime)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
enc = sys.argv[1].split(',') if len(sys.argv) > 1 else ['audio', 'video']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = init_weights(variable_specs(enc), seed=0, mode='bench')
GROUPS = int(os.environ.get('NGROUPS', '1'))       # (NGROUPS: bash ignores assignments to GROUPS) grouped launch: GROUPS batches of B per forward call (times below are per CALL)
inp = synth_inputs(B * GROUPS, enc, seed=1234)
net = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=GROUPS)
net.load_variables(P)
a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda() if 'video' in inp else None
if v is not None and os.environ.get('U8', '1') == '1':        # frames as decoded (uint8): the default entry point of deploy / evaluate / bench
    v = torch.round((v.double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
f = torch.as_tensor(inp['flow']).cuda() if 'flow' in inp else None
for _ in range(3): net.inference_ops(a, v, f)
if os.environ.get('TUNE', '1') == '1':
    plan = net.autotune(a, v, f)
    for row in plan: print('plan %-44s %-30s sk=%-3d %8.1f us' % row)
# FORCE="<tile id>:<layer substring>[,...]": pin matching layers to a tile after tuning (A/B of kernel variants inside the forward)
for item in filter(None, os.environ.get('FORCE', '').split(',')):
    tid_, sub = item.split(':')
    for name in variable_specs(enc):
        if name.endswith('/weights') and sub in name:
            net.plan_set(B, name[:-len('/weights')], int(tid_), int(os.environ.get('FORCE_SK', '1')))
net.profile_enable(B, True)
acc = None
N = 5
for _ in range(N):
    net.inference_ops(a, v, f)
    rows = net.profile_report(B)
    if acc is None: acc = [[k, l, 0.0, fl] for k, l, us, fl in rows]
    for r, (k, l, us, fl) in zip(acc, rows): r[2] += us / N
tot = sum(r[2] for r in acc)
print('%-34s %-40s %9s %8s %7s' % ('kernel', 'layer', 'us', 'TFLOP/s', '%'))
for k, l, us, fl in acc:
    print('%-34s %-40s %9.1f %8.1f %6.1f%%' % (k, l, us, GROUPS * fl / us / 1e6 if fl else 0, 100 * us / tot))
print('total %.1f us' % tot)
by = {}
for k, l, us, fl in acc:
    b = by.setdefault(k, [0, 0.0, 0.0]); b[0] += 1; b[1] += us; b[2] += fl
for k, (n, us, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print('%-34s n=%3d %9.1f us %6.1f%% %8.1f TFLOP/s' % (k, n, us, 100 * us / tot, GROUPS * fl / us / 1e6 if fl else 0))
