"""Dev helper (GPU box): conv3p_kernel alone - planes packed once outside the timed region, per-launch HIP-event time for a
list of (B,H,W,Cin,N) shapes and tiles; with --trace the s_memtime phase stamps of every workgroup's wave 0.
Builds its own small library from conv3p.hip + p3.hip (+ EXTRA_FLAGS) so schedule variants can be compared on one box.
usage: ubench_p3.py [--trace] [--flags "-DX ..."] tile_name shape [shape...]   shape = B,H,W,Cin,N"""
import sys, os, subprocess, ctypes as C, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
csrc = os.path.join(ROOT, 'spatialaudiogen_amd', 'csrc')
args = sys.argv[1:]
trace = '--trace' in args
if trace: args.remove('--trace')
nostats = '--nostats' in args
if nostats: args.remove('--nostats')
flags = []
if '--flags' in args:
    i = args.index('--flags'); flags = args[i + 1].split(); del args[i:i + 2]
tile_name, shapes = args[0], [tuple(int(v) for v in a.split(',')) for a in args[1:]]
tag = hashlib.md5((' '.join(flags) + str(trace)).encode()).hexdigest()[:8]
out = '/tmp/libp3_%s.so' % tag
entry = '/tmp/p3_entry.hip'
open(entry, 'w').write('''
#include "%s/kernels.h"
namespace sagen { char* err_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap); fprintf(stderr, "%%s\\n", err_buf()); return code; } }
using namespace sagen;
extern "C" int p3dbg_pack(const float* x, void* p3, int B, int H, int W, int C, void* s) { return p3_pack_launch(x, nullptr, nullptr, BnRef(), nullptr, 0, nullptr, p3, B, H, W, C, (hipStream_t)s); }
extern "C" int p3dbg_conv(const void* p3, const float* w, float* y, double* stats, void* trace, int B, int H, int W, int C, int N, int tile, void* s) {
    IgemmDesc d;
    d.w = w; d.y = y; d.stats = stats; d.w_split = 1;
    d.M = B * H * W; d.N = N; d.K = 9 * C; d.Kpad = d.K; d.Hg = H; d.Wg = W; d.Hin = H; d.Win = W; d.Cin = C; d.ldx = C;
    d.x_bstride = (long)H * W * C; d.ntaps = 9; d.TW = 3; d.tap_h0 = -1; d.tap_w0 = -1; d.log2Cin = ilog2_exact(C);
    d.Cout = N; d.Hlim = H; d.Wlim = W; d.ldy = N; d.y_rstride = (long)W * N; d.y_bstride = (long)H * W * N;
    d.xp3 = p3; d.p3_np = B * H * (W + 1); d.xp3_cstride = (unsigned)((size_t)d.p3_np * 96); d.xp3_bytes = (unsigned)p3_bytes(B, H, W, C);
    d.w_bytes = (unsigned)((size_t)N * d.Kpad * 4);
    d.trace = trace;
    return conv3p_dispatch(d, (IgemmTile)tile, (hipStream_t)s);
}
''' % csrc)
if not os.path.exists(out):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-fno-vectorize'] + \
          (['-DSAGEN_TRACE'] if trace else []) + flags + [os.path.join(csrc, 'conv3p.hip'), os.path.join(csrc, 'p3.hip'), entry, '-o', out]
    subprocess.check_call(cmd)
lib = C.CDLL(out)
from spatialaudiogen_amd.model import SptAudioGen
tile = SptAudioGen.tile_names().index(tile_name)
bm = 128 if tile_name.startswith('conv3pp') else int(tile_name.split('<')[1].split(',')[0])
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (B, H, W, Cin, N) in shapes:
    x = torch.randn(B, H, W, Cin, device='cuda')
    w = torch.randn(3, 3, Cin, N, device='cuda') / np.sqrt(9 * Cin)
    K = 9 * Cin
    wp = w.reshape(K, N).t().contiguous()                       # [N][K], k = (tap, c)
    hi = wp.bfloat16(); r1 = wp - hi.float(); mid = r1.bfloat16(); lo = (r1 - mid.float()).bfloat16()
    planes = torch.stack([hi, mid, lo], 0).reshape(3, N, K // 16, 16).permute(2, 0, 1, 3).contiguous()   # [K/16][3][N][16]
    wbuf = torch.zeros(N * K + (3 * N * K + 1) // 2, device='cuda')
    wbuf[:N * K] = wp.reshape(-1)
    wbuf[N * K:].view(torch.bfloat16)[:3 * N * K] = planes.reshape(-1)
    np3 = B * H * (W + 1)
    p3 = torch.zeros((Cin // 16) * np3 * 96 // 4 + 64, device='cuda')
    assert lib.p3dbg_pack(p(x), p(p3), B, H, W, Cin, None) == 0
    y = torch.empty(B, H, W, N, device='cuda'); stats = None if nostats else torch.zeros(2 * N, dtype=torch.float64, device='cuda')
    nblk = 8192
    trc = torch.zeros(nblk * 16, dtype=torch.int64, device='cuda') if trace else None
    for _ in range(3):
        assert lib.p3dbg_conv(p(p3), p(wbuf), p(y), p(stats), p(trc), B, H, W, Cin, N, tile, None) == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    err = float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); lib.p3dbg_conv(p(p3), p(wbuf), p(y), p(stats), p(trc), B, H, W, Cin, N, tile, None); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * B * H * W * N * K
    us = float(np.median(ts))
    print('%s %s flags=%s: %.1f us  %.1f TF (fp32-equiv)  rel err vs torch fp32 conv %.2e  blocks %d' % (tile_name, (B, H, W, Cin, N), ' '.join(flags), us, fl / us / 1e6, err, nblk * -(-N // 64)), flush=True)
    if trace:
        trc.zero_(); lib.p3dbg_conv(p(p3), p(wbuf), p(y), p(stats), p(trc), B, H, W, Cin, N, tile, None); torch.cuda.synchronize()
        t = trc.cpu().numpy().reshape(nblk, 16).astype(np.int64)
        ok = t[:, 6] > 0
        t = t[ok]
        if tile_name.startswith('conv3pp'):
            g = np.maximum(t[:, 4], 1)[:, None]
            ph = np.median(t[:, 8:15] / g, 0)
            print('   workgroups %d, %d groups per team; life %d ticks; per group (median s_memtime ticks of team 0 / wave 0): tap-0 reads %d | wait partner %d | '
                  'tap0 %d tap1 %d tap2 %d | DMA landed %d | end barrier %d  = %d per group' % ((len(t), np.median(t[:, 4]), np.median(t[:, 6] - t[:, 0])) + tuple(ph) + (ph.sum(),)))
            continue
        print('   workgroups %d: tiles/WG min %d median %d max %d; per WG (median ticks): entry->first issue %d | K loops %d | epilogues %d | life %d; per tile: K %d, epilogue %d'
              % (len(t), t[:, 4].min(), np.median(t[:, 4]), t[:, 4].max(), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2]), np.median(t[:, 3]),
                 np.median(t[:, 6] - t[:, 0]), np.median(t[:, 2] / np.maximum(t[:, 4], 1)), np.median(t[:, 3] / np.maximum(t[:, 4], 1))))
        continue
        t0 = t[:, 0].min()
        # occupancy over time: how many workgroups are between their first and last stamp
        ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 6], -np.ones(len(t))], 1)])
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        act = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0])
        print('   stamped workgroups %d; mean active (between stamps) %.0f; life sum / span = %.1f' % (len(t), (act[:-1] * dt).sum() / max(dt.sum(), 1), (t[:, 6] - t[:, 0]).sum() / (t[:, 6].max() - t0)))
        d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 6] - t[:, 5], t[:, 6] - t[:, 0]], 1)
        span = t[:, 6].max() - t0
        print('   phases (median s_memtime ticks): setup %d | fill-issue %d | first group %d | remaining groups %d | drain+rowinfo %d | epilogue %d | total %d'
              % tuple(np.median(d, 0)))
        print('   kernel span %d ticks (= %.1f us measured: %.1f ticks/us); block start ticks percentiles 50/90/99/max: %s' % (span, us, span / us,
              np.percentile((t[:, 0] - t0), [50, 90, 99, 100]).round(0)))
