"""Dev helper (GPU box): first FFT stage (and addresses) that goes wrong while a bf16x3 conv runs on another stream."""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
from spatialaudiogen_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'fftdbg.hip'), '-o', '/tmp/libfftdbg.so'])
D = C.CDLL('/tmp/libfftdbg.so')
x = torch.randn(32, 56, 112, 64, device='cuda'); w = torch.randn(3, 3, 64, 64, device='cuda') * 0.05
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
D.dbg_init(None); torch.cuda.synchronize()
blocks = 2048
inp = torch.randn(64, 1024, 2, device='cuda')
dump = torch.zeros(blocks, 6, 1024, 2, device='cuda')
D.dbg_run(C.c_void_p(inp.data_ptr()), C.c_void_p(dump.data_ptr()), blocks, None); torch.cuda.synchronize()
ref = dump.clone()
def conv(): ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True)
conv(); torch.cuda.synchronize()
for trial in range(12):
    dump.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        for _ in range(3): conv()
    with torch.cuda.stream(sa):
        D.dbg_run(C.c_void_p(inp.data_ptr()), C.c_void_p(dump.data_ptr()), blocks, C.c_void_p(sa.cuda_stream))
    torch.cuda.synchronize()
    bad = (dump != ref).any(dim=3)                  # [blocks, 6, 1024]
    if not bad.any(): continue
    per_stage = bad.any(dim=2).sum(dim=0).cpu().numpy()
    print('trial %d: workgroups with a wrong entry per slot (0=input, 1..5 = after stage 0..4): %s' % (trial, per_stage))
    b = int(bad.any(dim=2).any(dim=1).nonzero()[0])
    first = int(bad[b].any(dim=1).nonzero()[0])
    idx = bad[b, first].nonzero().flatten().cpu().numpy()
    print('  workgroup %d: first wrong slot %d, %d wrong entries, indices %s ...' % (b, first, idx.size, idx[:16]))
    i0 = int(idx[0])
    print('  entry %d: got %s want %s' % (i0, dump[b, first, i0].cpu().numpy(), ref[b, first, i0].cpu().numpy()))
    # does the wrong value equal another entry of the reference slot (misdirected access)?
    diff = (ref[b, first] - dump[b, first, i0]).abs().sum(dim=1); j = int(diff.argmin())
    print('  nearest reference entry in the same slot: index %d (dist %.3g);' % (j, float(diff[j])), 'in previous slot:', end=' ')
    diff = (ref[b, first - 1] - dump[b, first, i0]).abs().sum(dim=1); j = int(diff.argmin()); print('index %d (dist %.3g)' % (j, float(diff[j])))
    break
else:
    print('no corruption observed')
