"""Dev helper: time one conv layer through the op-level C ABI (weights packed once)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd import _lib, ops
CASES = {
    's2': (32, 56, 112, 64, 3, 64, 1),
    's3': (32, 28, 56, 128, 3, 128, 1),
    's4': (32, 14, 28, 256, 3, 256, 1),
    's5': (32, 7, 14, 512, 3, 512, 1),
}
name = sys.argv[1] if len(sys.argv) > 1 else 's2'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, H, W, Cin, k, Cout, s = CASES[name]
x = torch.randn(B, H, W, Cin, device='cuda')
w = torch.randn(k, k, Cin, Cout, device='cuda') / np.sqrt(k * k * Cin)
for _ in range(3):
    y, st = ops.conv_2d(x, w, s, 'SAME', return_bn_stats=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(iters):
    torch.cuda.synchronize()
    e0.record(); y, st = ops.conv_2d(x, w, s, 'SAME', return_bn_stats=True); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
fl = 2.0 * B * (H // s) * (W // s) * Cout * k * k * Cin
print(name, 'median %.1f us (incl. pack) -> %.1f TFLOP/s' % (np.median(ts) * 1e3, fl / np.median(ts) / 1e9))
