"""Does a HIP graph of the native training step (forward + loss + backward, two streams) beat eager launches?  B = 32, audio+video."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.train import Trainer, synthetic_batches
from spatialaudiogen_amd.weights import init_weights
enc, B = ['audio', 'video'], 32
net = SptAudioGen(1, encoders=enc, separation='unet_mask')
tr = Trainer(net, batch=B, variables=init_weights(net.variable_specs(), seed=0, mode='bench', fc3_std=0.05))
a, v, f, t, m = next(synthetic_batches(enc, B, seed=3, pool=1))
dev = [torch.as_tensor(x).cuda() if x is not None else None for x in (a, v, f, t, m)]
tr.autotune(*dev[:4])

def eager(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.forward_backward(*dev)
        tr.opt.apply()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager(5)
print('eager   : %.3f ms per step' % eager(30))
ref = [g.clone() for g in tr.opt.grads]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tr.forward_backward(*dev)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.forward_backward(*dev)
def graphed(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
        tr.opt.apply()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
graphed(5)
print('graphed : %.3f ms per step' % graphed(30))
