"""Per-kernel / per-layer launch times of one training step (native hipEvent profiler), B=32 audio+video."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.train import Trainer
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs

enc = ['audio', 'video'] if len(sys.argv) < 2 else sys.argv[1].split('+')
B = 32
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1234)
tgt = (inp['audio'][:, 24000:28800, :] * np.array([0.5, 0.25, -0.5], np.float32)).astype(np.float32)
net = SptAudioGen(1, encoders=enc, separation='unet_mask'); net.load_variables(P)
tr = Trainer(net, batch=B)
dev = [torch.as_tensor(inp[k]).cuda() if k in inp else None for k in ('audio', 'video', 'flow')] + [torch.as_tensor(tgt).cuda()]
if dev[1] is not None and os.environ.get('U8', '1') == '1':       # frames as decoded (uint8): the default entry point of the training feeder / bench
    dev[1] = torch.round((dev[1].double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
if '--tune' in sys.argv:
    tr.autotune(*dev)
for _ in range(3):
    tr.step(*dev)
torch.cuda.synchronize()
if '--steps-only' in sys.argv:         # for `rocprofv3 --kernel-trace --stats`: the kernels as they run inside the step (two streams, no events)
    for _ in range(int(os.environ.get('STEPS_ONLY', '20'))):
        tr.step(*dev)
    torch.cuda.synchronize()
    sys.exit(0)
tr.profile_enable(True)
agg, layers = {}, []
N = 3
for it in range(N):
    tr.forward_backward(*dev)
    rows = tr.profile_report()
    for k, layer, us, fl in rows:
        a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += fl
    if it == N - 1:
        layers = rows
tr.profile_enable(False)
tot = sum(a[1] for a in agg.values()) / N
print('total kernel time per step: %.1f us' % tot)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-46s n=%3d  %9.1f us/step  %5.1f%%  %7.1f TF' % (k, a[0] // N, a[1] / N, 100 * a[1] / N / tot, a[2] / max(a[1], 1e-9) / 1e6))
print()
for k, layer, us, fl in layers:
    print('%-38s %-58s %9.1f us %7.1f TF' % (k[:38], layer[:58], us, fl / max(us, 1e-9) / 1e6))
# optimiser + repack cost (torch-side timing)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.forward_backward(*dev)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(5):
    tr.step(*dev)
torch.cuda.synchronize()
t2 = time.perf_counter()
print('\nforward_backward %.3f ms   full step %.3f ms' % ((t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3))
