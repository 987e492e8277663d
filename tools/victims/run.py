"""Dev helper (GPU box): run each victim kernel concurrently with a bf16x3 conv and compare with its solo result."""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spatialaudiogen_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC'] + os.environ.get('VICTIM_FLAGS', '').split() + [os.path.join(here, 'victims.hip'), '-o', '/tmp/libvictims.so'])
V = C.CDLL('/tmp/libvictims.so')
x = torch.randn(32, 56, 112, 64, device='cuda'); w = torch.randn(3, 3, 64, 64, device='cuda') * 0.05
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
V.victims_init(None); torch.cuda.synchronize()
def conv(): ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True)
conv(); torch.cuda.synchronize()
for name, fn, nout, reps in (('table (global 8-B loads of a __device__ array)', V.run_table, 2, 400), ('valu fma chain', V.run_valu, 1, 4000),
                             ('lds ping-pong + barriers', V.run_lds, 4, 200), ('sqrt chain', V.run_sqrt, 1, 1500)):
    blocks = 2048
    out = torch.zeros(blocks * 256 * nout, device='cuda')
    fn(C.c_void_p(out.data_ptr()), blocks, reps, None); torch.cuda.synchronize()
    ref = out.clone(); worst = 0.0; nbad = 0
    for _ in range(20):
        out.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(3): conv()
        with torch.cuda.stream(sa):
            fn(C.c_void_p(out.data_ptr()), blocks, reps, C.c_void_p(sa.cuda_stream))
        torch.cuda.synchronize()
        d = (out - ref).abs(); worst = max(worst, float(d.max())); nbad += int((d > 0).sum())
    print('%-50s max diff vs solo %.3g, wrong elements over 20 trials %d' % (name, worst, nbad), flush=True)

# an exact copy of stft_kernel living in THIS library (own tables), vs the one inside libsagen_hip.so
B = 32
audio = torch.randn(B, 52799, device='cuda')
V.stft_init(None); torch.cuda.synchronize()
mag = torch.zeros(B, 127, 1024, device='cuda'); spec = torch.zeros(B, 28, 513, 2, device='cuda')
V.run_stft(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), None); torch.cuda.synchronize()
ref = mag.clone(); lib_ref, _ = ops.stft_mag(audio, 46, 173, 89, 117); torch.cuda.synchronize()
print('copy vs library stft (solo): max diff %.3g' % float((ref - lib_ref).abs().max()))
worst_copy = worst_lib = 0.0
for _ in range(20):
    with torch.cuda.stream(sb):
        for _ in range(3): conv()
    with torch.cuda.stream(sa):
        V.run_stft(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), C.c_void_p(sa.cuda_stream))
    torch.cuda.synchronize(); worst_copy = max(worst_copy, float((mag - ref).abs().max()))
    with torch.cuda.stream(sb):
        for _ in range(3): conv()
    with torch.cuda.stream(sa):
        m2, _ = ops.stft_mag(audio, 46, 173, 89, 117)
    torch.cuda.synchronize(); worst_lib = max(worst_lib, float((m2 - lib_ref).abs().max()))
print('stft COPY in the victims library, concurrent with conv: max diff %.3g' % worst_copy)
print('stft inside libsagen_hip.so,    concurrent with conv: max diff %.3g' % worst_lib)

V.stft_init2(None); torch.cuda.synchronize()
for name, fn in (('twiddles from an LDS copy', V.run_stft_lds), ('twiddles by 16-byte aligned loads', V.run_stft_a16), ('original (8-byte twiddle loads)', V.run_stft)):
    fn(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), None); torch.cuda.synchronize()
    solo = float((mag - ref).abs().max()); worst = 0.0
    for _ in range(30):
        with torch.cuda.stream(sb):
            for _ in range(3): conv()
        with torch.cuda.stream(sa):
            fn(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), C.c_void_p(sa.cuda_stream))
        torch.cuda.synchronize(); worst = max(worst, float((mag - ref).abs().max()))
    print('stft variant: %-36s solo diff %.3g   concurrent with conv: max diff %.3g' % (name, solo, worst), flush=True)

for dyn in (0, 32 << 10, 100 << 10, 128 << 10):
    V.run_stft_biglds(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), dyn, None); torch.cuda.synchronize()
    solo = float((mag - ref).abs().max()); worst = 0.0
    for _ in range(30):
        with torch.cuda.stream(sb):
            for _ in range(3): conv()
        with torch.cuda.stream(sa):
            V.run_stft_biglds(C.c_void_p(audio.data_ptr()), B, 52799, C.c_void_p(mag.data_ptr()), C.c_void_p(spec.data_ptr()), dyn, C.c_void_p(sa.cuda_stream))
        torch.cuda.synchronize(); worst = max(worst, float((mag - ref).abs().max()))
    print('stft with %3d KiB of extra dynamic LDS: solo diff %.3g, concurrent with conv: max diff %.3g' % (dyn >> 10, solo, worst), flush=True)
