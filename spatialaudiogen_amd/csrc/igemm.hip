// Implicit-GEMM convolution / transposed convolution / FC on the gfx950 fp32 matrix cores.
//
// Replaces, for the inference path, tf.nn.convolution (core.py:206), tf.nn.conv2d_transpose
// (core.py:140) and tf.matmul (core.py:79) of the reference's TF1 runtime.
//
// Design (CDNA4):
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A wave owns a (WM x WN) patch of
//    32x32 accumulators; a 256-thread workgroup (one wave per SIMD) owns BM x BN.
//  * No im2col: the A tile is gathered straight from the NHWC activation (16 B per lane, a
//    (tap, 4-channel) group per load), optionally through the previous layer's batch-norm
//    (relu(v*scale+shift)), and staged in LDS rows of BK+4 floats (conflict-free ds_read_b128).
//  * k is consumed in a lane-permuted order: lane (i, kk) of the MFMA holds the float4
//    A[i][8u+4kk .. +3]; MFMA r of the group contracts k in {8u+r, 8u+4+r}.  The filter is packed
//    [n][k] (k contiguous) so B fragments are read the same way.  One ds_read_b128 per operand
//    per four MFMAs.
//  * register-staged double buffering: global loads of tile t+1 are in flight while tile t is
//    contracted; one barrier per K tile.
//  * epilogue: bias / ReLU, depth-to-space scatter (transposed convs), per-tile per-channel
//    (sum, sumsq) partials for training-mode batch-norm, or raw split-K partials.
#include "kernels.h"

namespace sagen {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmDesc d) {
    constexpr int LD = BK + 4;              // LDS row stride in floats (16-B aligned, odd multiple of 4)
    constexpr int KQ = BK / 4;              // float4 groups per row per K tile
    constexpr int RPP = 256 / KQ;           // rows loaded per pass
    constexpr int A_IT = (BM + RPP - 1) / RPP;
    constexpr int B_IT = (BN + RPP - 1) / RPP;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
    __shared__ long s_rowoff[BM];
    __shared__ int s_hrem[BM], s_wrem[BM];
    float* As = smem;                        // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;          // [2][BN][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware M-tile remap (block b runs on XCD b%8; give each XCD a contiguous run of tiles so
    // neighbouring tiles' halo rows hit the same L2). Bijective for any grid size.
    const int gm = gridDim.x;
    int tile_m;
    {
        const int bid = blockIdx.x;
        const int q = gm >> 3, r = gm & 7, xcd = bid & 7, j = bid >> 3;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = tile_m * BM;
    const int n0 = blockIdx.y * BN;
    const int z = blockIdx.z;

    const int HgWg = d.Hg * d.Wg;
    // per-row output addressing for the epilogue
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        long off = 0;
        int hr = 0, wr = 0;
        if (m < d.M) {
            const int b = m / HgWg;
            const int rem = m - b * HgWg;
            const int ia = rem / d.Wg;
            const int a = d.g_h0 + ia, bb = d.g_w0 + (rem - ia * d.Wg);
            off = (long)b * d.y_bstride + (long)(a * d.dsh) * d.y_rstride + (long)(bb * d.dsw) * d.ldy;
            hr = d.Hlim - a * d.dsh;
            wr = d.Wlim - bb * d.dsw;
        }
        s_rowoff[r] = off;
        s_hrem[r] = hr;
        s_wrem[r] = wr;
    }

    // loader state: this thread stages float4 group `kq` of rows lrow + it*RPP
    const int kq = tid % KQ;
    const int lrow = tid / KQ;
    long a_off[A_IT];
    int a_hi0[A_IT], a_wi0[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = lrow + it * RPP;
        const int m = m0 + row;
        a_off[it] = 0;
        a_hi0[it] = -(1 << 28);
        a_wi0[it] = 0;
        if (row < BM && m < d.M) {
            const int b = m / HgWg;
            const int rem = m - b * HgWg;
            const int ia = rem / d.Wg;
            const int hi0 = (d.g_h0 + ia) * d.in_sh, wi0 = (d.g_w0 + (rem - ia * d.Wg)) * d.in_sw;
            a_hi0[it] = hi0;
            a_wi0[it] = wi0;
            a_off[it] = (long)b * d.x_bstride + ((long)hi0 * d.Win + wi0) * d.ldx;
        }
    }
    const float* b_ptr[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = lrow + it * RPP;
        const int n = n0 + row;
        b_ptr[it] = (row < BN && n < d.N) ? d.w + (long)n * d.Kpad + 4 * kq : nullptr;
    }

    const int nk = d.Kpad / BK;
    const int nk_per = (nk + d.splitk - 1) / d.splitk;
    const int kc0 = z * nk_per;
    const int kc1 = min(nk, kc0 + nk_per);
    const bool prologue = d.in_scale != nullptr;

    float4 ra[A_IT], rb[B_IT];
    auto load_tile = [&](int kc) {
        const int k = kc * BK + 4 * kq;
        int tap = 0, c = k;
        if (d.ntaps > 1) {
            tap = k >> d.log2Cin;
            c = k & (d.Cin - 1);
        }
        const int th = tap / d.TW;
        const int dh = th * d.tap_sh + d.tap_h0;
        const int dw = (tap - th * d.TW) * d.tap_sw + d.tap_w0;
        const long toff = ((long)dh * d.Win + dw) * d.ldx + c;
        const bool kok = k < d.K;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prologue && kok) {
            sc = ldg4(d.in_scale + c);
            sh = ldg4(d.in_shift + c);
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int hi = a_hi0[it] + dh, wi = a_wi0[it] + dw;
            const bool ok = kok && (unsigned)hi < (unsigned)d.Hin && (unsigned)wi < (unsigned)d.Win;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                v = ldg4(d.x + a_off[it] + toff);
                if (prologue) {
                    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
                    v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
                    v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                }
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            rb[it] = b_ptr[it] ? ldg4(b_ptr[it] + kc * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = lrow + it * RPP;
            if (A_IT * RPP == BM || row < BM)
                *reinterpret_cast<float4*>(&As[(buf * BM + row) * LD + 4 * kq]) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int row = lrow + it * RPP;
            if (B_IT * RPP == BN || row < BN)
                *reinterpret_cast<float4*>(&Bs[(buf * BN + row) * LD + 4 * kq]) = rb[it];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, kk = lane >> 5;
    if (kc0 < kc1) {
        load_tile(kc0);
        store_tile(0);
    }
    __syncthreads();
    for (int kc = kc0; kc < kc1; ++kc) {
        const int buf = (kc - kc0) & 1;
        const bool more = kc + 1 < kc1;
        if (more) load_tile(kc + 1);
        const float* Ab = &As[(buf * BM + wm * WM + li) * LD + 4 * kk];
        const float* Bb = &Bs[(buf * BN + wn * WN + li) * LD + 4 * kk];
#pragma unroll
        for (int u = 0; u < BK / 8; ++u) {
            float4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LD + 8 * u);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LD + 8 * u);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int dswC = d.dsw * d.Cout;
    const bool to_ws = d.splitk_ws != nullptr;     // raw partials for the split-K / replicate reduce
    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        csum[j] = 0.f;
        csq[j] = 0.f;
        const int n = n0 + wn * WN + j * 32 + li;
        const bool nok = n < d.N;
        int ry = 0, rx = 0, o = n;
        if (d.dsh * d.dsw > 1) {
            ry = n / dswC;
            const int rem = n - ry * dswC;
            rx = rem / d.Cout;
            o = rem - rx * d.Cout;
        }
        const long coloff = (long)ry * d.y_rstride + (long)rx * d.ldy + o;
        const float bias = (d.bias && nok && !to_ws) ? d.bias[o] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int m = m0 + row;
                float v = acc[i][j][e];
                if (to_ws) {
                    if (nok && m < d.M) d.splitk_ws[((long)z * d.M + m) * d.N + n] = v;
                } else {
                    const bool ok = nok && ry < s_hrem[row] && rx < s_wrem[row];
                    if (ok) {
                        csum[j] += v;
                        csq[j] += v * v;
                        v += bias;
                        if (d.relu_out) v = fmaxf(v, 0.f);
                        d.y[s_rowoff[row] + coloff] = v;
                    }
                }
            }
        }
    }
    if (d.stats != nullptr) {
        // per-tile per-channel partial sums of the raw conv output (pre-bias; BN convs have none)
        __syncthreads();                       // all waves are done reading the A/B tiles
        float* red = smem;                     // [2][WAVES_M][BN]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32);
            float q = csq[j] + __shfl_xor(csq[j], 32);
            if (kk == 0) {
                const int col = wn * WN + j * 32 + li;
                red[(0 * WAVES_M + wm) * BN + col] = s;
                red[(1 * WAVES_M + wm) * BN + col] = q;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 256) {
            const int which = t / BN, col = t - which * BN;
            const int n = n0 + col;
            if (n < d.N) {
                float s = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WAVES_M; ++w2) s += red[(which * WAVES_M + w2) * BN + col];
                d.stats[((long)tile_m * 2 + which) * d.N + n] = s;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 16>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

static int tile_bm(IgemmTile t) {
    switch (t) {
        case TILE_128x128: case TILE_128x64: case TILE_128x32: return 128;
        case TILE_256x64: return 256;
        case TILE_64x64: return 64;
        case TILE_32x128: return 32;
        default: return 0;
    }
}
static int tile_bn(IgemmTile t) {
    switch (t) {
        case TILE_128x128: case TILE_32x128: return 128;
        case TILE_128x64: case TILE_256x64: case TILE_64x64: return 64;
        case TILE_128x32: return 32;
        default: return 0;
    }
}

const char* igemm_tile_name(IgemmTile t) {
    switch (t) {
        case TILE_128x128: return "igemm_kernel<128,128,64,64,16>";
        case TILE_128x64: return "igemm_kernel<128,64,64,32,16>";
        case TILE_256x64: return "igemm_kernel<256,64,64,64,16>";
        case TILE_64x64: return "igemm_kernel<64,64,32,32,16>";
        case TILE_128x32: return "igemm_kernel<128,32,32,32,16>";
        case TILE_32x128: return "igemm_kernel<32,128,32,32,16>";
        default: return "igemm_kernel<?>";
    }
}

IgemmTile igemm_pick_tile(const IgemmDesc& d) {
    if (d.M <= 32) return TILE_32x128;
    if (d.N <= 32) return TILE_128x32;
    auto blocks = [&](IgemmTile t) { return (long)cdiv(d.M, tile_bm(t)) * cdiv(d.N, tile_bn(t)) * d.splitk; };
    const long want = 2 * 256;                 // >= 2 workgroups per CU
    if (d.N <= 64) {
        if (blocks(TILE_256x64) >= want) return TILE_256x64;
        if (blocks(TILE_128x64) >= want) return TILE_128x64;
        return TILE_64x64;
    }
    if (blocks(TILE_128x128) >= want) return TILE_128x128;
    if (blocks(TILE_128x64) >= want) return TILE_128x64;
    return TILE_64x64;
}

int igemm_grid_m(const IgemmDesc& d, IgemmTile tile) {
    if (tile == TILE_AUTO) tile = igemm_pick_tile(d);
    return cdiv(d.M, tile_bm(tile));
}

int igemm_launch(const IgemmDesc& d, IgemmTile tile, hipStream_t s) {
    if (!d.x || !d.w || (!d.y && !d.splitk_ws)) return fail(SAGEN_ERR_NULL, "igemm: null operand");
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return fail(SAGEN_ERR_SHAPE, "igemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    if (d.Kpad % 16 || d.Kpad < d.K) return fail(SAGEN_ERR_SHAPE, "igemm: Kpad=%d must be a multiple of 16 >= K=%d", d.Kpad, d.K);
    if (d.K % 4 || d.Cin % 4) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: Cin=%d / K=%d must be multiples of 4", d.Cin, d.K);
    if (d.ntaps > 1 && (d.log2Cin < 2 || (1 << d.log2Cin) != d.Cin))
        return fail(SAGEN_ERR_UNSUPPORTED, "igemm: multi-tap conv needs power-of-two Cin (got %d)", d.Cin);
    if (d.K != d.ntaps * d.Cin) return fail(SAGEN_ERR_SHAPE, "igemm: K=%d != ntaps*Cin=%d", d.K, d.ntaps * d.Cin);
    if (d.splitk > 1 && !d.splitk_ws) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: split-K needs a workspace");
    if (d.splitk_ws && (d.stats || d.dsh * d.dsw != 1))
        return fail(SAGEN_ERR_UNSUPPORTED, "igemm: partial-sum output needs a plain epilogue");
    if (d.N != d.dsh * d.dsw * d.Cout) return fail(SAGEN_ERR_SHAPE, "igemm: N=%d != dsh*dsw*Cout", d.N);
    if (tile == TILE_AUTO) tile = igemm_pick_tile(d);
    switch (tile) {
        case TILE_128x128: return launch_cfg<128, 128, 64, 64>(d, s);
        case TILE_128x64: return launch_cfg<128, 64, 64, 32>(d, s);
        case TILE_256x64: return launch_cfg<256, 64, 64, 64>(d, s);
        case TILE_64x64: return launch_cfg<64, 64, 32, 32>(d, s);
        case TILE_128x32: return launch_cfg<128, 32, 32, 32>(d, s);
        case TILE_32x128: return launch_cfg<32, 128, 32, 32>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "igemm: bad tile id %d", (int)tile);
    }
}

// -----------------------------------------------------------------------------------------
// split-K reduction + bias + activation + (optional) row replication
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N,
                                                            const float* __restrict__ bias, int relu,
                                                            float* __restrict__ y, int ldy, int rep) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
    float v = 0.f;
    for (int zz = 0; zz < splitk; ++zz) v += ws[((long)zz * M + m) * N + n];
    if (bias) v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    for (int r = 0; r < rep; ++r) y[((long)m * rep + r) * ldy + n] = v;
}

int splitk_reduce_launch(const float* ws, int splitk, int M, int N, const float* bias, int relu,
                         float* y, int ldy, int rep, hipStream_t s) {
    const long total = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, ws, splitk, M, N, bias, relu,
                       y, ldy, rep);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// filter repacking
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int ntaps, int cin_src, int cin_pad,
                                                        int cout, float* __restrict__ wp, int Npad, int Kpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Npad * Kpad) return;
    const int n = (int)(idx / Kpad), k = (int)(idx - (long)n * Kpad);
    float v = 0.f;
    if (n < cout && k < ntaps * cin_pad) {
        const int tap = k / cin_pad, c = k - tap * cin_pad;
        if (c < cin_src) v = w[((long)tap * cin_src + c) * cout + n];
    }
    wp[idx] = v;
}

int pack_conv_launch(const float* w_hwio, int ntaps, int cin_src, int cin_pad, int cout, float* wp, int Npad,
                     int Kpad, hipStream_t s) {
    const long total = (long)Npad * Kpad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w_hwio, ntaps, cin_src, cin_pad,
                       cout, wp, Npad, Kpad);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

__global__ __launch_bounds__(256) void pack_deconv_kernel(const float* __restrict__ w, int kh, int kw, int cout, int cin,
                                                          int sh, int sw, int nth, int ntw, float* __restrict__ wp,
                                                          int Npad, int Kpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Npad * Kpad) return;
    const int n = (int)(idx / Kpad), k = (int)(idx - (long)n * Kpad);
    float v = 0.f;
    if (n < sh * sw * cout && k < nth * ntw * cin) {
        const int ry = n / (sw * cout), rx = (n / cout) % sw, o = n % cout;
        const int tap = k / cin, c = k - tap * cin;
        const int dp = tap / ntw, dq = tap - dp * ntw;
        const int p = ry + sh * dp, q = rx + sw * dq;
        if (p < kh && q < kw) v = w[(((long)p * kw + q) * cout + o) * cin + c];
    }
    wp[idx] = v;
}

int pack_deconv_launch(const float* w_hwoi, int kh, int kw, int cout, int cin, int sh, int sw, float* wp, int Npad,
                       int Kpad, hipStream_t s) {
    const int nth = cdiv(kh, sh), ntw = cdiv(kw, sw);
    const long total = (long)Npad * Kpad;
    hipLaunchKernelGGL(pack_deconv_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w_hwoi, kh, kw, cout, cin, sh, sw,
                       nth, ntw, wp, Npad, Kpad);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
