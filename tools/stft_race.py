"""Dev helper (GPU box): does stft_kernel stay correct while other kernels run on other streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from spatialaudiogen_amd import ops, _lib
if os.environ.get('SAGEN_LIB'): _lib.LIB_PATH = os.environ['SAGEN_LIB']
torch.manual_seed(0)
B = 32
audio = torch.randn(B, 52799, device='cuda')
ref, _ = ops.stft_mag(audio, 46, 173, 89, 117); torch.cuda.synchronize()
x = torch.randn(B, 56, 112, 64, device='cuda'); w = torch.randn(3, 3, 64, 64, device='cuda') * 0.05
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
def trial(name, other, n=30):
    worst = 0.0
    for _ in range(n):
        with torch.cuda.stream(s_b):
            for _ in range(2): other()
        with torch.cuda.stream(s_a):
            m, _ = ops.stft_mag(audio, 46, 173, 89, 117)
        torch.cuda.synchronize()
        worst = max(worst, float((m - ref).abs().max()))
    print('%-40s max |mag - ref| = %.3g' % (name, worst), flush=True)
def stft_other(): ops.stft_mag(audio, 46, 173, 89, 117)
def conv_other(): ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True)
def pool_other(): ops.maxpool3x3s2(x)
def bn_other(): ops.bn_apply_relu(x)
def torch_other(): torch.mm(torch.randn(2048, 2048, device='cuda'), torch.randn(2048, 2048, device='cuda'))
trial('alone', lambda: None)
trial('vs conv3x3 tile %s' % os.environ.get('SAGEN_FORCE_TILE', 'default'), conv_other)
trial('vs conv3x3 WITHOUT bn statistics', lambda: ops.conv_2d(x, w, 1, 'SAME'))
xs = torch.randn(4, 56, 112, 64, device='cuda')
trial('vs small conv3x3 (B=4) with statistics', lambda: ops.conv_2d(xs, w, 1, 'SAME', return_bn_stats=True))
A16 = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); B16 = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
A32 = torch.randn(4096, 4096, device='cuda'); B32 = torch.randn(4096, 4096, device='cuda')
Ah = A16.to(torch.float16); Bh = B16.to(torch.float16)
trial('vs torch.mm bf16 8192^3 (hipBLASLt/rocBLAS)', lambda: torch.mm(A16, B16))
trial('vs torch.mm fp16 8192^3', lambda: torch.mm(Ah, Bh))
trial('vs torch.mm fp32 4096^3', lambda: torch.mm(A32, B32))
