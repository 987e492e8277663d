"""When does each gradient bucket complete inside the training step (sagen_train_set_grad_events)?  B = 32, audio+video."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.train import Trainer, synthetic_batches
from spatialaudiogen_amd.weights import init_weights
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
enc, B = ['audio', 'video'], 32
net = SptAudioGen(1, encoders=enc, separation='unet_mask')
P = init_weights(net.variable_specs(), seed=0, mode='bench', fc3_std=0.05)
tr = Trainer(net, batch=B, variables=P, bucket_bytes=mb << 20, overlap=False)
tr._enable_overlap(timing=True)
a, v, f, t, m = next(synthetic_batches(enc, B, seed=3, pool=1))
dev = [torch.as_tensor(x).cuda() if x is not None else None for x in (a, v, f, t, m)]
tr.autotune(*dev[:4])
for _ in range(3):
    tr.forward_backward(*dev)
start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
start.record(); tr.forward_backward(*dev); end.record()
torch.cuda.synchronize()
print('%d MiB buckets: sizes (MB) %s' % (mb, [round(g.numel() * 4 / 1e6, 1) for g in tr.opt.grads]))
print('bucket complete at (ms after step start): %s; step (forward + backward) %.2f ms' %
      ([round(start.elapsed_time(e), 2) for e in tr.bucket_events], start.elapsed_time(end)))
