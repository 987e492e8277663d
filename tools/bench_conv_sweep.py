"""Dev helper: 3x3 conv 64->64 on [B,64,128,64] (M = 8192*B rows) so that the 256x64 tiling gives exactly
32*B blocks; sweep B to see 1,2,3,4.. blocks per CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd import ops
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for B in (8, 16, 24, 32, 48, 64):
    x = torch.randn(B, 64, 128, C, device='cuda')
    w = torch.randn(3, 3, C, C, device='cuda') / np.sqrt(9 * C)
    for _ in range(3):
        y, st = ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        e0.record(); y, st = ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    fl = 2.0 * B * 64 * 128 * C * 9 * C
    print('C=%d B=%d M=%d  median %.1f us -> %.1f TFLOP/s' % (C, B, B * 8192, np.median(ts) * 1e3, fl / np.median(ts) / 1e9), flush=True)
