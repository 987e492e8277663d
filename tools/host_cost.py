"""Host-side cost of enqueueing one forward / one training step (the GPU runs behind): wall time of the native call itself."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
enc = ['audio', 'video']
B = 32
torch.cuda.set_device(0)
net = SptAudioGen(1, encoders=enc, separation='unet_mask')
net.load_variables(init_weights(variable_specs(enc), seed=0, mode='bench'))
inp = synth_inputs(B, enc, seed=1)
a, v = torch.as_tensor(inp['audio']).cuda(), torch.as_tensor(inp['video']).cuda()
for _ in range(3):
    net.inference_ops(a, v)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    net.inference_ops(a, v)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('forward: host enqueue %.3f ms per call, GPU-complete %.3f ms per call' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
from spatialaudiogen_amd.train import Trainer
tgt = torch.as_tensor((inp['audio'][:, 24000:28800, :] * np.array([0.5, 0.25, -0.5], np.float32)).astype(np.float32)).cuda()
tr = Trainer(net, batch=B)
for _ in range(3):
    tr.step(a, v, None, tgt)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    tr.step(a, v, None, tgt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('train step: host enqueue %.3f ms per call, GPU-complete %.3f ms per call' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# from an idle queue: how far ahead of the GPU does the host get within ONE step?
hs, gs = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(a, v, None, tgt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); gs.append((t2 - t0) * 1e3)
print('train step from an idle queue: host returns after %.3f ms (median), GPU done after %.3f ms' % (sorted(hs)[4], sorted(gs)[4]))
