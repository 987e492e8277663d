import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libtr16.so'))
out = torch.zeros(256, dtype=torch.int16, device='cuda')
rc = lib.tr16_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().numpy().reshape(64, 4)
print('rc', rc)
for l in range(64):
    print('lane %2d: ' % l + ' '.join('e%-3d(src lane %2d elem %d)' % (v, v // 4, v % 4) for v in o[l]))
