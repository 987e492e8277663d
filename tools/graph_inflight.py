"""Dev helper (GPU box): N contexts in flight, each replaying its forward from a HIP graph (torch.cuda.CUDAGraph around inference_ops),
against the same contexts launched eagerly - is a configuration bound by the launching thread?
    python tools/graph_inflight.py audio 10 | audio+video 16 | audio+video 32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
enc = sys.argv[1].split('+') if len(sys.argv) > 1 else ['audio']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
nets, outs, streams = [], [], []
a = [torch.as_tensor(inp['audio']).cuda()]
if 'video' in inp: a.append(torch.round((torch.as_tensor(inp['video']).cuda().double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8))
plan = None
for j in range(NF):
    net = SptAudioGen(1, encoders=enc, separation='unet_mask'); net.load_variables(P)
    if plan is None: plan = net.autotune(*a)
    else: net.load_plan_rows(plan) if hasattr(net, 'load_plan_rows') else net.autotune(*a)
    nets.append(net); outs.append(torch.empty(B, 4800, 3, device='cuda')); streams.append(torch.cuda.Stream())
def eager_round():
    for j in range(NF):
        with torch.cuda.stream(streams[j]): nets[j].inference_ops(*a, out=outs[j])
for _ in range(5): eager_round()
torch.cuda.synchronize()
ref = outs[0].clone()
def timeit(f, n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n * NF) * 1e3
t_e = timeit(eager_round)
graphs = []
for j in range(NF):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(streams[j]):
        nets[j].inference_ops(*a, out=outs[j])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=streams[j]):
        nets[j].inference_ops(*a, out=outs[j])
    graphs.append(g)
def graph_round():
    for j in range(NF):
        with torch.cuda.stream(streams[j]): graphs[j].replay()
outs[0].zero_(); graph_round(); torch.cuda.synchronize()
print('graph output identical:', bool(torch.equal(outs[0], ref)))
t_g = timeit(graph_round)
print('%s B=%d, %d in flight: eager %.3f ms per forward = %.0f ambisonic-s/s; graph replay %.3f ms = %.0f' % ('+'.join(enc), B, NF, t_e, B * 0.1 / t_e * 1e3, t_g, B * 0.1 / t_g * 1e3))
