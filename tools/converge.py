"""Loss curve of the training step on the synthetic learnable task of train.synthetic_batches (target = a fixed per-channel
scaling of the mono crop): evidence that forward + backward + Adam + moving averages descend, beyond the 3-step parity test.
usage: python tools/converge.py [av|a] [steps] [lr]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.weights import variable_specs, init_weights
from spatialaudiogen_amd.train import Trainer, synthetic_batches

cfg = sys.argv[1] if len(sys.argv) > 1 else 'av'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
enc = ['audio', 'video'] if cfg == 'av' else ['audio']
B = 32
torch.cuda.set_device(0)
net = SptAudioGen(1, encoders=enc, separation='unet_mask')
net.load_variables(init_weights(variable_specs(enc), seed=0, mode='bench'))
tr = Trainer(net, batch=B, lr=lr)
it = synthetic_batches(enc, B, seed=5, pool=4)
losses = []
t0 = time.time()
for s in range(steps):
    a, v, f, t, m = next(it)
    losses.append(float(tr.step(a, v, f, t, m)[0]))
torch.cuda.synchronize()
dt = time.time() - t0
print('config %s, B=%d, lr=%g, %d steps in %.1f s (incl. host-side batch upload)' % (cfg, B, lr, steps, dt))
for s in range(0, steps, max(1, steps // 30)):
    w = losses[s:s + max(1, steps // 30)]
    print('step %4d  loss mean %.6g  min %.6g  max %.6g' % (s, np.mean(w), np.min(w), np.max(w)))
print('first-10 mean %.6g  last-10 mean %.6g  ratio %.4f  finite %s' % (np.mean(losses[:10]), np.mean(losses[-10:]),
      np.mean(losses[-10:]) / np.mean(losses[:10]), bool(np.isfinite(losses).all())))
