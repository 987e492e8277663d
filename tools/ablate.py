"""Dev helper (GPU box): time one 3x3 conv problem with the full kernel, without the DMA loads, and without the
MFMAs, for a list of tile ids.  usage: ablate.py B H W C tile [tile...]"""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
csrc = os.path.join(ROOT, 'spatialaudiogen_amd', 'csrc')
srcs = [os.path.join(csrc, f) for f in ('igemm.hip', 'igemm3.hip', 'igemm3dw.hip', 'igemm3s2.hip', 'elementwise.hip', 'fft.hip', 'eval.hip', 'model.hip', 'api.hip')]
extra = '/tmp/ablate_entry.hip'
open(extra, 'w').write('''
#include "%s/kernels.h"
using namespace sagen;
extern "C" int sagen_dbg_conv(const float* x, const float* wp, float* y, int B, int H, int W, int C, int N, int tile, void* stream) {
    IgemmDesc d;
    d.x = x; d.w = wp; d.y = y;
    d.M = B * H * W; d.N = N; d.K = 9 * C; d.Kpad = d.K; d.Hg = H; d.Wg = W; d.Hin = H; d.Win = W; d.Cin = C; d.ldx = C;
    d.x_bstride = (long)H * W * C; d.ntaps = 9; d.TW = 3; d.tap_h0 = -1; d.tap_w0 = -1; d.log2Cin = ilog2_exact(C);
    d.Cout = N; d.Hlim = H; d.Wlim = W; d.ldy = N; d.y_rstride = (long)W * N; d.y_bstride = (long)H * W * N;
    if (igemm_tile_split((IgemmTile)tile)) { d.w_split = 1; static int packed = 0; if (!packed) { packed = 1; int rc = pack_split_launch((float*)wp, d.N, d.Kpad, (hipStream_t)stream); if (rc) return rc; } }
    return igemm_launch(d, (IgemmTile)tile, (hipStream_t)stream);
}
''' % csrc)
libs = {}
VARIANTS = (('full', []), ('no_dma', ['-DSAGEN_ABLATE_DMA', '-DSAGEN_ABLATE_A', '-DSAGEN_ABLATE_B']), ('no_mfma', ['-DSAGEN_ABLATE_MFMA']),
            ('A_only', ['-DSAGEN_ABLATE_MFMA', '-DSAGEN_ABLATE_B']), ('B_only', ['-DSAGEN_ABLATE_MFMA', '-DSAGEN_ABLATE_A']),
            ('mfma+A', ['-DSAGEN_ABLATE_B']), ('mfma+B', ['-DSAGEN_ABLATE_A']))
for name, flag in VARIANTS:
    out = '/tmp/libsagen_%s.so' % name
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + flag + srcs + [extra, '-o', out])
    libs[name] = C.CDLL(out)
B, H, W, Cc = [int(v) for v in sys.argv[1:5]]
N = Cc
tiles = [int(v) for v in sys.argv[5:]]
x = torch.randn(B, H, W, Cc, device='cuda'); wp = torch.randn(3 * N, 9 * Cc, device='cuda') * 0.05   # room for the bf16x3 planes
y = torch.empty(B, H, W, N, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())
fl = 2.0 * B * H * W * N * 9 * Cc
names = ['128x128', '128x64', '256x64', '64x64', '128x32', '32x128', '128x128s2', '128x64s2', '256x64s2', '64x64s2', '64x128', '64x128s2', '64x256', '64x256s2', '256x32']
for t in tiles:
    res = []
    for name, _ in VARIANTS:
        lib = libs[name]
        for _ in range(3):
            assert lib.sagen_dbg_conv(p(x), p(wp), p(y), B, H, W, Cc, N, t, None) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            e0.record(); lib.sagen_dbg_conv(p(x), p(wp), p(y), B, H, W, Cc, N, t, None); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(np.median(ts))
    print('tile %-10s ' % (names[t] if t < len(names) else t) + '  '.join('%s %6.1f' % (n, r) for (n, _), r in zip(VARIANTS, res)) + '  us  (full = %.1f TF)' % (fl / res[0] / 1e6), flush=True)
