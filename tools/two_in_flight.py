"""Dev helper (GPU box): throughput with one vs two batches in flight (two contexts on two streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
enc = ['audio', 'video']; B = 32
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda()
NN = int(os.environ.get('NNETS', '3'))
nets = [SptAudioGen(1, encoders=enc, separation='unet_mask') for _ in range(NN)]
streams = [torch.cuda.Stream() for _ in range(NN)]
outs = []
for n in nets:
    n.load_variables(P); outs.append(n.inference_ops(a, v))
plan = nets[0].autotune(a, v)
nets[0].save_plan(B, '/tmp/plan.json')
for n in nets[1:]: n.load_plan(B, '/tmp/plan.json')
torch.cuda.synchronize()
EVENTS = os.environ.get('WITH_EVENTS') == '1'
def run(k, N=60):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    for i in range(6):
        with torch.cuda.stream(streams[i % k]): nets[i % k].inference_ops(a, v, out=outs[i % k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        if EVENTS: evs[i][0].record(streams[i % k])
        with torch.cuda.stream(streams[i % k]): nets[i % k].inference_ops(a, v, out=outs[i % k])
        if EVENTS: evs[i][1].record(streams[i % k])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return 0.1 * B * N / dt, dt / N * 1e3
seq = []
for n in nets:
    seq.append(n.inference_ops(a, v).clone())
torch.cuda.synchronize()
print('sequential references agree:', all(bool(torch.equal(seq[0], x)) for x in seq))
for k in [kk for kk in (1, 2, 3, 1, 2) if kk <= NN]:
    v_, ms = run(k)
    d = [float((outs[j] - seq[j]).abs().max()) for j in range(k)]
    print('%d batch(es) in flight: %.1f ambisonic-s/s  (%.3f ms per batch)   max |out - sequential| per context: %s' % (k, v_, ms, d), flush=True)

