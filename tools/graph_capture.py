"""Dev helper (GPU box): capture one sagen_forward (audio only, deploy.py's batch of 10 - the launch-latency-bound configuration) in a
HIP graph through torch.cuda.CUDAGraph and compare replay time / output with eager launches."""
import os, sys, time
os.environ['SAGEN_ONE_STREAM'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
enc = sys.argv[1].split('+') if len(sys.argv) > 1 else ['audio']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
net = SptAudioGen(1, encoders=enc, separation='unet_mask'); net.load_variables(P)
a = [torch.as_tensor(inp[k]).cuda() for k in ['audio'] + [k for k in ('video', 'flow') if k in inp]]
out = torch.empty(B, 4800, 3, device='cuda')
net.autotune(*a)
for _ in range(5): net.inference_ops(*a, out=out)
torch.cuda.synchronize()
ref = out.clone()
def timeit(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
eager = timeit(lambda: net.inference_ops(*a, out=out))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.inference_ops(*a, out=out)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    net.inference_ops(*a, out=out)
out.zero_(); g.replay(); torch.cuda.synchronize()
print('graph output identical:', bool(torch.equal(out, ref)))
graph = timeit(g.replay)
print('%s B=%d: eager %.1f us / forward, graph replay %.1f us' % ('+'.join(enc), B, eager, graph))
