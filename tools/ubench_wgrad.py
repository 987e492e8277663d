"""Kernel time of one weight gradient on its own (run under rocprofv3 --kernel-trace --stats to split the MFMA kernel from the split-K
reducer).  usage: ubench_wgrad.py B H W Cin Cout [kh kw] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialaudiogen_amd import ops
a = [int(v) for v in sys.argv[1:]]
B, H, W, Cin, Cout = a[:5]
kh, kw = (a[5], a[6]) if len(a) > 6 else (3, 3)
reps = a[7] if len(a) > 7 else 20
torch.cuda.set_device(0)
x = torch.randn(B, H, W, Cin, device='cuda')
dy = torch.randn(B, H, W, Cout, device='cuda')
for _ in range(3):
    ops.wgrad(x, dy, kh, kw, (1, 1), (-(kh // 2), -(kw // 2)))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.wgrad(x, dy, kh, kw, (1, 1), (-(kh // 2), -(kw // 2)))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = 2.0 * B * H * W * kh * kw * Cin * Cout
print('B=%d %dx%d %d->%d %dx%d: %.1f us per call, %.1f TFLOP/s' % (B, H, W, Cin, Cout, kh, kw, us, fl / us / 1e6))
