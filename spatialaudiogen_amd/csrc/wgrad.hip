// Weight gradients of every contraction of the path (conv / conv2d_transpose / fully_connected) as ONE kernel.
//
// Backward of tf.nn.convolution (core.py:206), tf.nn.conv2d_transpose (core.py:140) and tf.matmul (core.py:79) with respect
// to their filters, i.e. what tf.gradients builds for train.py:147-149 (opt.minimize, myutils.py:220-221).  All three are
//
//     dW[t][g][d] = sum over p = (b, i, j) of  G[b, i*sh + th(t) + h0, j*sw + tw(t) + w0, g] * D[b, i, j, d]
//
// with D the tensor on the coarse grid and G the tensor that is gathered with the tap displacement:
//   conv  (HWIO weights [kh,kw,Cin,Cout]) : G = layer input x,   D = dL/dy          -> dW[tap][cin][cout]
//   deconv ([kh,kw,Cout,Cin], core.py:118): G = dL/dy (fine),    D = layer input x  -> dW[tap][cout][cin]
//   FC    ([in,out])                      : one tap, 1x1 grid    G = x, D = dL/dy   -> dW[in][out]
// so the result lands directly in the TF variable layout (= the gradient bucket of train.py: AdamBuckets).
//
// Design (CDNA4): the contraction index is the PIXEL, which is the slow dimension of both NHWC operands - [k][m] and [k][n]
// tiles.  That is the layout the fp32 matrix instruction wants (v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5],
// B[k=l>>5][j=l&31] - 32 consecutive channels of one pixel per half-wave), so both tiles go global -> LDS by LDS-DMA exactly
// as they lie in memory (no transpose, no register staging, no ds_write) and fragments are contiguous ds_read_b32/b64;
// the bf16x3 route of the forward kernels would need both operands transposed AND split in the loop.
//  * workgroup = (tap, 128|64 channels of G, 128|64 channels of D, pixel range z); 4 waves as 2x2; K tile = 16 pixels;
//    3-stage LDS ring with counted vmcnt, DMA issue spread between the MFMAs (same skeleton as igemm.hip);
//  * pixel -> (b, i, j) by two mul-hi divisions per DMA instruction; padding / out-of-range pixels / channel tails set
//    bit 31 of the buffer offset and the hardware range check writes zeros;
//  * with two 32-row sub-tiles per wave the sub-tile index is the LOW bit of the channel (row r of sub-tile ti is channel
//    2r + ti), so one ds_read_b64 feeds both sub-tiles; the epilogue undoes the permutation;
//  * pixel ranges (split-K) write raw partials, reduced by splitk_reduce_kernel in a fixed order (deterministic gradients);
//    the block index is remapped so that the tiles of one pixel range share an XCD (L2).
#include "igemm3_common.h"
#include <algorithm>
#include <type_traits>
#ifdef SAGEN_WGRAD_MUL32          // dev builds: 32-bit index multiplies (A/B of the 24-bit ones)
#define WG_MUL(a, b) ((a) * (b))
#else
#define WG_MUL(a, b) __umul24((a), (b))
#endif
#include <cstdlib>

namespace sagen {

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradDesc d) {
    constexpr int BK = 16;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_CPR = BM / 4, B_CPR = BN / 4;              // 16-byte chunks per pixel row of the tile
    constexpr int A_RPI = 64 / A_CPR, B_RPI = 64 / B_CPR;      // pixel rows per DMA instruction (64 lanes x 16 B)
    constexpr int A_DMA = BK / A_RPI, B_DMA = BK / B_RPI;
    static_assert(A_DMA % 4 == 0 && B_DMA % 4 == 0, "every wave issues the same number of DMA instructions");
    constexpr int A_PW = A_DMA / 4, B_PW = B_DMA / 4;
    constexpr int PER = A_PW + B_PW;
    constexpr int STAGES = 3;
    constexpr int TILE_F = (BM + BN) * BK;
    constexpr int NMFMA = (BK / 2) * MT * NT;

    __shared__ __attribute__((aligned(16))) float smem[STAGES * TILE_F];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kk = lane >> 5;

    // block -> (pixel range z, tap, channel tiles); blocks of one XCD (bid % 8) get a contiguous run, so one z stays on one L2
    const int ntaps = d.TH * d.TW;
    const int tiles_g = (d.Cg + BM - 1) / BM, tiles_d = (d.Cd + BN - 1) / BN;
    const int ntile = ntaps * tiles_g * tiles_d;
    int n;
    {
        const int gm = gridDim.x, bid = blockIdx.x;
        const int q = gm >> 3, r = gm & 7, xcd = bid & 7, j = bid >> 3;
        n = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int z = n / ntile;
    int rem = n - z * ntile;
    const int tap = rem / (tiles_g * tiles_d);
    rem -= tap * (tiles_g * tiles_d);
    const int tg = rem / tiles_d, td = rem - tg * tiles_d;
    const int th = tap / d.TW, tw = tap - th * d.TW;
    const int g0 = tg * BM, d0 = td * BN;
    const int roff = th * d.tsh + d.h0, coff = tw * d.tsw + d.w0;

    const int nchunks = (d.P + BK - 1) / BK;
    const int per_z = (nchunks + d.splitk - 1) / d.splitk;
    const int kc0 = z * per_z;
    const int kc1 = min(nchunks, kc0 + per_z);

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.g, 0, d.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.d, 0, d.d_bytes, 0x00020000);

    // per-lane constants of the loaders
    const unsigned a_cb = (unsigned)(g0 + 4 * (lane % A_CPR)), b_cb = (unsigned)(d0 + 4 * (lane % B_CPR));
    const unsigned a_cbad = a_cb < (unsigned)d.Cg ? 0u : OOB, b_cbad = b_cb < (unsigned)d.Cd ? 0u : OOB;
    unsigned a_voff[A_PW], b_voff[B_PW];
    int i_stage = 0;

    // pixel p -> byte offsets of its D row and of its (tap-displaced) G row; OOB when outside
    auto pixel = [&](unsigned p, unsigned& goff, unsigned& doff) {
        // every factor is < 2^24 (wgrad_launch checks): 24-bit multiplies are full rate, 32-bit ones quarter rate - 20 of them per
        // K step were a third of this kernel's VALU time
        const unsigned r = d.Wd == 1 ? p : __umulhi(p, d.magic_w);
        const unsigned jj = p - WG_MUL(r, (unsigned)d.Wd);
        const unsigned b = d.Hd == 1 ? r : __umulhi(r, d.magic_h);
        const unsigned ii = r - WG_MUL(b, (unsigned)d.Hd);
        const bool ok = p < (unsigned)d.P;
        doff = ok ? (WG_MUL(b, d.d_bstride) + WG_MUL(ii, d.d_rstride) + WG_MUL(jj, (unsigned)d.ldd)) * 4u : OOB;
        const int gi = (int)WG_MUL(ii, (unsigned)d.sh) + roff, gj = (int)WG_MUL(jj, (unsigned)d.sw) + coff;
        const bool gok = ok && (unsigned)gi < (unsigned)d.HG && (unsigned)gj < (unsigned)d.WG;
        goff = gok ? (WG_MUL(b, d.g_bstride) + WG_MUL((unsigned)gi, d.g_rstride) + WG_MUL((unsigned)gj, (unsigned)d.ldg)) * 4u : OOB;
    };
    auto begin_issue = [&](int kc, int stage) {
        i_stage = stage;
        const unsigned pbase = (unsigned)kc * BK;
        if constexpr (BM == BN) {
#pragma unroll
            for (int j = 0; j < A_PW; ++j) {
                unsigned go, dof;
                pixel(pbase + (unsigned)((wave + 4 * j) * A_RPI + lane / A_CPR), go, dof);
                a_voff[j] = (go + a_cb * 4u) | a_cbad;
                b_voff[j] = (dof + b_cb * 4u) | b_cbad;
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_PW; ++j) {
                unsigned go, dof;
                pixel(pbase + (unsigned)((wave + 4 * j) * A_RPI + lane / A_CPR), go, dof);
                a_voff[j] = (go + a_cb * 4u) | a_cbad;
            }
#pragma unroll
            for (int j = 0; j < B_PW; ++j) {
                unsigned go, dof;
                pixel(pbase + (unsigned)((wave + 4 * j) * B_RPI + lane / B_CPR), go, dof);
                b_voff[j] = (dof + b_cb * 4u) | b_cbad;
            }
        }
    };
    auto issue_one = [&](int g) {
        float* st = smem + i_stage * TILE_F;
        if (g < A_PW) dma16(g_rsrc, st + (wave + 4 * g) * 256, a_voff[g], 0);
        else dma16(d_rsrc, st + BK * BM + (wave + 4 * (g - A_PW)) * 256, b_voff[g - A_PW], 0);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ntiles = kc1 - kc0;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < ntiles) {
            begin_issue(kc0 + t, t);
#pragma unroll
            for (int g = 0; g < PER; ++g) issue_one(g);
        }
    if (ntiles > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    lds_barrier();

    int stage = 0;
    for (int kc = kc0; kc < kc1; ++kc) {
        const bool more = kc + (STAGES - 1) < kc1;
        int istage = stage + (STAGES - 1);
        if (istage >= STAGES) istage -= STAGES;
        if (more) begin_issue(kc + (STAGES - 1), istage);
        const float* As = smem + stage * TILE_F + wm * WM + MT * li;
        const float* Bs = smem + stage * TILE_F + BK * BM + wn * WN + NT * li;
        float af[BK / 2][MT], bf[BK / 2][NT];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int krow = 2 * s + kk;
            if constexpr (MT == 2) {
                const float2 v = *reinterpret_cast<const float2*>(As + krow * BM);
                af[s][0] = v.x; af[s][1] = v.y;
            } else {
                af[s][0] = As[krow * BM];
            }
            if constexpr (NT == 2) {
                const float2 v = *reinterpret_cast<const float2*>(Bs + krow * BN);
                bf[s][0] = v.x; bf[s][1] = v.y;
            } else {
                bf[s][0] = Bs[krow * BN];
            }
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
                    const int idx = (s * MT + i) * NT + j;
#pragma unroll
                    for (int g = 0; g < PER; ++g)
                        if (idx == (g + 1) * NMFMA / (PER + 1) - 1 && more) issue_one(g);
                }
        if (more) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        lds_barrier();
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }

    // epilogue.  C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5); row r of sub-tile i is channel MT*r + i
    float* out = d.splitk > 1 ? d.ws + (size_t)z * ntaps * d.Cg * d.Cd : d.out;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = (e & 3) + 8 * (e >> 2) + 4 * kk;
            const int g = g0 + wm * WM + MT * rr + i;
            if (g >= d.Cg) continue;
            float* orow = out + ((size_t)tap * d.Cg + g) * d.Cd;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int dd = d0 + wn * WN + NT * li + j;
                if (dd < d.Cd) orow[dd] = acc[i][j][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// wgrad3_kernel: the same contraction on the bf16 matrix cores, fp32-equivalent by the 3-way operand split of igemm3.hip
// (six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulate): 16x the matrix rate of the exact kernel above.
// The obstacle is the operand layout: the bf16 MFMA wants, per lane, 8 CONSECUTIVE k (= pixels) of one channel, and NHWC has
// the channels contiguous.  gfx950's LDS transpose read resolves it without any shuffling: the planes are kept in LDS in the
// natural [pixel][channel] order (each thread splits the 4 channels of one pixel it loaded - one 16-byte load, one 8-byte
// ds_write per plane) and ds_read_b64_tr_b16 delivers the column-major fragment.  Its lane mapping, measured on MI355X
// (tools/probe/tr16.py): in every 16-lane group, source lane q hands in 4 contiguous elements, row q/4, columns 4(q%4)..+3 of a
// 4 x 16 block; destination lane c receives column c, rows 0..3.  So the group (lane>>4) = (k-group g, channel half h) fetches
// pixels 8g+4r .. +3 (r = 0, 1: two reads make the 8-deep fragment) x channels 16h .. 16h+15 of the 32-row MFMA tile.
// Plane rows are padded by 64 bytes so that the four pixel rows of a read fall into four different 64-byte bank ranges.
// Two LDS stages; the global loads of chunk t+2 are in flight while chunk t is contracted and chunk t+1 is split / stored.
// ------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const char* p, int row_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * row_bytes));
    // (a plain concatenation: element-wise construction of the 8 x 16-bit vector cost 120 v_mov per K step)
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#else
    return bf16x8{};
#endif
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad3_kernel(const WgradDesc d) {
    constexpr int BK = 16;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int RSA = BM * 2 + 64, RSB = BN * 2 + 64;        // bytes per pixel row of one plane (64 B of padding: bank spread)
    constexpr int PLA = BK * RSA, PLB = BK * RSB;
    constexpr int ST = 3 * (PLA + PLB);
    constexpr int A_CPR = BM / 4, B_CPR = BN / 4;              // float4 chunks per pixel row
    constexpr int A_RPP = 256 / A_CPR, B_RPP = 256 / B_CPR;    // pixel rows per pass of the 256 threads
    constexpr int A_PT = BK / A_RPP, B_PT = BK / B_RPP;        // loads per thread and chunk
    __shared__ __attribute__((aligned(16))) char smem[2 * ST];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // d.fold > 1 (the stem: 32 gathered channels, 7 filter rows): a tile's rows are `fold` consecutive filter rows x Cg channels,
    // so the D operand (205 MB for the stem) is read once per `fold` filter rows and no MFMA rows are padding
    const int fold = d.fold;
    const int ntaps = fold > 1 ? (d.TH + fold - 1) / fold : d.TH * d.TW;
    const int tiles_g = fold > 1 ? 1 : (d.Cg + BM - 1) / BM, tiles_d = (d.Cd + BN - 1) / BN;
    const int ntile = ntaps * tiles_g * tiles_d;
    int n;
    {
        const int gm = gridDim.x, bid = blockIdx.x;
        const int q = gm >> 3, r = gm & 7, xcd = bid & 7, j = bid >> 3;
        n = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int z = n / ntile;
    int rem = n - z * ntile;
    const int tap = rem / (tiles_g * tiles_d);
    rem -= tap * (tiles_g * tiles_d);
    const int tg = rem / tiles_d, td = rem - tg * tiles_d;
    const int g0 = tg * BM, d0 = td * BN;
    int th = tap / d.TW;
    const int tw = tap - th * d.TW;
    unsigned a_cb = (unsigned)(g0 + 4 * (tid % (BM / 4)));
    unsigned a_cbad = a_cb < (unsigned)d.Cg ? 0u : OOB;
    if (fold > 1) {                                            // this thread's filter row and channel inside it
        const int gl = 4 * (tid % (BM / 4));
        const int tl = gl / d.Cg;
        th = tap * fold + tl;
        a_cb = (unsigned)(gl - tl * d.Cg);
        a_cbad = (tl < fold && th < d.TH) ? 0u : OOB;
    }
    const int roff = th * d.tsh + d.h0, coff = tw * d.tsw + d.w0;

    const int nchunks = (d.P + BK - 1) / BK;
    const int per_z = (nchunks + d.splitk - 1) / d.splitk;
    const int kc0 = z * per_z;
    const int kc1 = min(nchunks, kc0 + per_z);

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.g, 0, d.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.d, 0, d.d_bytes, 0x00020000);

    const unsigned b_cb = (unsigned)(d0 + 4 * (tid % B_CPR));
    const unsigned b_cbad = b_cb < (unsigned)d.Cd ? 0u : OOB;
    const int a_kp = tid / A_CPR, b_kp = tid / B_CPR;          // pixel row of pass 0
    const int a_wofs = a_kp * RSA + (tid % A_CPR) * 8, b_wofs = 3 * PLA + b_kp * RSB + (tid % B_CPR) * 8;

    auto pixel = [&](unsigned p, unsigned& goff, unsigned& doff) {
        // every factor is < 2^24 (wgrad_launch checks): 24-bit multiplies are full rate, 32-bit ones quarter rate - 20 of them per
        // K step were a third of this kernel's VALU time
        const unsigned r = d.Wd == 1 ? p : __umulhi(p, d.magic_w);
        const unsigned jj = p - WG_MUL(r, (unsigned)d.Wd);
        const unsigned b = d.Hd == 1 ? r : __umulhi(r, d.magic_h);
        const unsigned ii = r - WG_MUL(b, (unsigned)d.Hd);
        const bool ok = p < (unsigned)d.P;
        doff = ok ? (WG_MUL(b, d.d_bstride) + WG_MUL(ii, d.d_rstride) + WG_MUL(jj, (unsigned)d.ldd)) * 4u : OOB;
        const int gi = (int)WG_MUL(ii, (unsigned)d.sh) + roff, gj = (int)WG_MUL(jj, (unsigned)d.sw) + coff;
        const bool gok = ok && (unsigned)gi < (unsigned)d.HG && (unsigned)gj < (unsigned)d.WG;
        goff = gok ? (WG_MUL(b, d.g_bstride) + WG_MUL((unsigned)gi, d.g_rstride) + WG_MUL((unsigned)gj, (unsigned)d.ldg)) * 4u : OOB;
    };
    f32x4 ra[A_PT], rb[B_PT];
    auto load_chunk = [&](int kc) {
        const unsigned pbase = (unsigned)kc * BK;
        if constexpr (BM == BN) {
#pragma unroll
            for (int t = 0; t < A_PT; ++t) {
                unsigned go, dof;
                pixel(pbase + (unsigned)(a_kp + t * A_RPP), go, dof);
                ra[t] = bload16(g_rsrc, (go + a_cb * 4u) | a_cbad);
                rb[t] = bload16(d_rsrc, (dof + b_cb * 4u) | b_cbad);
            }
        } else {
#pragma unroll
            for (int t = 0; t < A_PT; ++t) {
                unsigned go, dof;
                pixel(pbase + (unsigned)(a_kp + t * A_RPP), go, dof);
                ra[t] = bload16(g_rsrc, (go + a_cb * 4u) | a_cbad);
            }
#pragma unroll
            for (int t = 0; t < B_PT; ++t) {
                unsigned go, dof;
                pixel(pbase + (unsigned)(b_kp + t * B_RPP), go, dof);
                rb[t] = bload16(d_rsrc, (dof + b_cb * 4u) | b_cbad);
            }
        }
    };
    auto split_store = [&](f32x4 v, char* dst, int plane_bytes) {
        float a = v[0], b = v[1], c = v[2], e = v[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            u32x2 w;
            w[0] = split_pair(a, b);
            w[1] = split_pair(c, e);
            *reinterpret_cast<u32x2*>(dst + pl * plane_bytes) = w;
        }
    };
    auto store_chunk = [&](int stage) {
        char* st = smem + stage * ST;
#pragma unroll
        for (int t = 0; t < A_PT; ++t) split_store(ra[t], st + a_wofs + t * A_RPP * RSA, PLA);
#pragma unroll
        for (int t = 0; t < B_PT; ++t) split_store(rb[t], st + b_wofs + t * B_RPP * RSB, PLB);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: lane group (lane>>4) = 2*g + h; source lane q = lane & 15 hands in row q/4, columns 4*(q%4)..+3
    const int q = lane & 15, gk = lane >> 5, hh = (lane >> 4) & 1;
    const int a_foff = (8 * gk + (q >> 2)) * RSA + (wm * WM + 16 * hh + 4 * (q & 3)) * 2;
    const int b_foff = 3 * PLA + (8 * gk + (q >> 2)) * RSB + (wn * WN + 16 * hh + 4 * (q & 3)) * 2;
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm

    const int ntiles = kc1 - kc0;
    if (ntiles > 0) {
        load_chunk(kc0);
        store_chunk(0);
        if (ntiles > 1) load_chunk(kc0 + 1);
    }
    lds_barrier();
    int stage = 0;
    for (int kc = kc0; kc < kc1; ++kc) {
        const char* st = smem + stage * ST;
        bf16x8 fa[3][MT], fb[3][NT];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[pl][i] = tr_frag(st + a_foff + pl * PLA + i * 64, RSA);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[pl][j] = tr_frag(st + b_foff + pl * PLB + j * 64, RSB);
        }
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[tt]][i], fb[TB[tt]][j], acc[i][j], 0, 0, 0);
        if (kc + 1 < kc1) {
            store_chunk(stage ^ 1);                    // chunk kc+1: its loads were issued one iteration ago
            if (kc + 2 < kc1) load_chunk(kc + 2);
        }
        lds_barrier();
        stage ^= 1;
    }

    // epilogue.  C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int li = lane & 31, kk = lane >> 5;
    float* out = d.splitk > 1 ? d.ws + (size_t)z * d.TH * d.TW * d.Cg * d.Cd : d.out;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int g = g0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            int otap = tap;
            if (fold > 1) {
                const int tl = g / d.Cg;
                g -= tl * d.Cg;
                otap = tap * fold + tl;
                if (tl >= fold || otap >= d.TH) continue;
            } else if (g >= d.Cg) continue;
            float* orow = out + ((size_t)otap * d.Cg + g) * d.Cd;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int dd = d0 + wn * WN + j * 32 + li;
                if (dd < d.Cd) orow[dd] = acc[i][j][e];
            }
        }
}

// wgrad3r_kernel: wgrad3_kernel for the dense 3x3 stride-1 SAME convolutions (13 of the 20 ResNet convs, half of all weight-gradient
// time), one workgroup per FILTER ROW instead of per tap.  wgrad3_kernel is bound by its VALU work, not by the matrix pipe: every
// workgroup splits its own copy of both operand tiles (about 8 VALU per element) and derives every pixel's address by two divisions,
// for each of the nine taps.  Here the three horizontal taps of a filter row share ONE staged G tile (18 pixel rows instead of
// 3 x 16: tap tw reads its fragments tw rows further down) and ONE D tile, so a K step splits (18*BM + 16*BN) elements for
// 3*6*MT*NT MFMAs - 2.8x fewer per MFMA - and the chunk's base pixel is decomposed once on the scalar unit, the rows of a thread
// following by compare-and-wrap.
// The contraction index runs over the PADDED pixel grid [B][Hd][Wd + 1] (one zero pixel closing every image row, the idea of
// conv3p.hip): the horizontal neighbours of a pixel are then its neighbours in the flat index - the pad pixel is the right-hand
// padding of its own row and the left-hand padding of the next - and a chunk may straddle image rows freely.  Pad pixels and
// rows outside the image are buffer-range-check zeros on both operands (1/Wd more K steps).
// ------------------------------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad3r_kernel(const WgradDesc d) {
    constexpr int BK = 16, NTW = 3, AR = BK + NTW - 1;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int RSA = BM * 2 + 64, RSB = BN * 2 + 64;
    constexpr int A_CPR = BM / 4, B_CPR = BN / 4;
    constexpr int A_RPP = 256 / A_CPR, B_RPP = 256 / B_CPR;
    constexpr int A_PT = (AR + A_RPP - 1) / A_RPP, B_PT = BK / B_RPP;
    static_assert(B_PT >= 1 && BK % B_RPP == 0, "D tile: whole passes");
    // the last G pass only has the two extra rows left: wave 0 alone runs it (a scalar branch); its 64 lanes cover 64 / A_CPR rows
    static_assert((A_PT - 1) * A_RPP == BK && 64 / A_CPR >= NTW - 1, "G tile: whole passes + one wave for the tap overlap");
    constexpr int AROWS = BK + 64 / A_CPR;
    constexpr int PLA = AROWS * RSA, PLB = BK * RSB;
    constexpr int ST = 3 * (PLA + PLB);
    __shared__ __attribute__((aligned(16))) char smem[2 * ST];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_g = (d.Cg + BM - 1) / BM, tiles_d = (d.Cd + BN - 1) / BN;
    const int ntile = d.TH * tiles_g * tiles_d;
    int n;
    {
        const int gm = gridDim.x, bid = blockIdx.x;
        const int q = gm >> 3, r = gm & 7, xcd = bid & 7, j = bid >> 3;
        n = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int z = n / ntile;
    int rem = n - z * ntile;
    const int th = rem / (tiles_g * tiles_d);
    rem -= th * (tiles_g * tiles_d);
    const int tg = rem / tiles_d, td = rem - tg * tiles_d;
    const int g0 = tg * BM, d0 = td * BN;
    const int roff = th * d.tsh + d.h0;
    const int Wp = d.Wd + 1;

    const int nchunks = (d.P + BK - 1) / BK;                   // d.P: PADDED pixels (wgrad_launch)
    const int per_z = (nchunks + d.splitk - 1) / d.splitk;
    const int kc0 = z * per_z;
    const int kc1 = min(nchunks, kc0 + per_z);

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.g, 0, d.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.d, 0, d.d_bytes, 0x00020000);

    const unsigned a_cb = (unsigned)(g0 + 4 * (tid % A_CPR)), b_cb = (unsigned)(d0 + 4 * (tid % B_CPR));
    const unsigned a_cbad = a_cb < (unsigned)d.Cg ? 0u : OOB, b_cbad = b_cb < (unsigned)d.Cd ? 0u : OOB;
    const int a_kp = tid / A_CPR, b_kp = tid / B_CPR;
    const int a_wofs = a_kp * RSA + (tid % A_CPR) * 8, b_wofs = 3 * PLA + b_kp * RSB + (tid % B_CPR) * 8;

    // pixel `delta` places after (b, i, j) of the padded grid; delta < 2 * Wp (wgrad_launch: Wd >= 12, delta <= 23)
    auto advance = [&](int& b, int& i, int& j, int delta) {
        j += delta;
        if (j >= Wp) { j -= Wp; ++i; }
        if (j >= Wp) { j -= Wp; ++i; }
        if (i >= d.Hd) { i -= d.Hd; ++b; }
        if (i >= d.Hd) { i -= d.Hd; ++b; }
    };
    f32x4 ra[2][A_PT], rb[2][B_PT];                          // two register sets: the loads run TWO K steps ahead of their split
    int cb0 = 0, ci0 = 0, cj0 = 0;                             // (b, i, j) of the first pixel of the chunk being loaded: wave-uniform
    auto chunk_base = [&](int kc) {
        const unsigned q0 = (unsigned)kc * BK;
        const unsigned r0 = __umulhi(q0, d.magic_w);           // / Wp: two scalar divisions per chunk
        cj0 = (int)(q0 - r0 * (unsigned)Wp);
        const unsigned bb = d.Hd == 1 ? r0 : __umulhi(r0, d.magic_h);
        ci0 = (int)(r0 - bb * (unsigned)d.Hd);
        cb0 = (int)bb;
    };
    auto load_a = [&](int set, int t) {
        const int row = a_kp + t * A_RPP;                      // slot `row` of the G tile holds padded pixel q0 - 1 + row
        int b = cb0, i = ci0, j = cj0 - 1;
        advance(b, i, j, row);                                 // (-1 only at a row start: that neighbour is the previous row's pad pixel - zero)
        const int gi = (int)WG_MUL((unsigned)i, (unsigned)d.sh) + roff;
        const bool ok = j >= 0 && j < d.Wd && b < d.B && (unsigned)gi < (unsigned)d.HG;
        const unsigned off = ok ? ((WG_MUL((unsigned)b, d.g_bstride) + WG_MUL((unsigned)gi, d.g_rstride) + WG_MUL((unsigned)j, (unsigned)d.ldg) + a_cb) * 4u) | a_cbad : OOB;
        ra[set][t] = bload16(g_rsrc, off);
    };
    auto load_b = [&](int set, int t) {
        int b = cb0, i = ci0, j = cj0;
        advance(b, i, j, b_kp + t * B_RPP);
        const bool ok = j < d.Wd && b < d.B;
        const unsigned off = ok ? ((WG_MUL((unsigned)b, d.d_bstride) + WG_MUL((unsigned)i, d.d_rstride) + WG_MUL((unsigned)j, (unsigned)d.ldd) + b_cb) * 4u) | b_cbad : OOB;
        rb[set][t] = bload16(d_rsrc, off);
    };
    // one plane of one staged row: the bf16 pair planes are peeled off in place (v keeps the residual for the next plane)
    auto split_plane = [&](f32x4& v, char* dst) {
        float a = v[0], b = v[1], c = v[2], e = v[3];
        u32x2 w;
        w[0] = split_pair(a, b);
        w[1] = split_pair(c, e);
        v = f32x4{a, b, c, e};
        *reinterpret_cast<u32x2*>(dst) = w;
    };
    // The staging work of one K step as UNITS that the K loop places between its MFMAs (left in one block after them, the wave
    // runs matrix phase and VALU phase back to back - 9 VALU per MFMA, measured - and the matrix pipe idles through the second):
    // per staged row: plane 0, 1, 2 of chunk kc+1 into the other stage, then the load of the same row of chunk kc+2.  No
    // per-lane predicates (rows past the range load range-check zeros; the last K steps stage chunks nobody reads).
    constexpr int NUNIT = 4 * (A_PT + B_PT);
    auto unit = [&](int set, int u, char* st_next) {
        const int r = u >> 2, k = u & 3;                       // row slot (A rows first), piece
        if (r < A_PT) {
            if (r == A_PT - 1 && wave != 0) return;
            if (k < 3) split_plane(ra[set][r], st_next + a_wofs + r * A_RPP * RSA + k * PLA);
            else load_a(set, r);
        } else {
            const int t = r - A_PT;
            if (k < 3) split_plane(rb[set][t], st_next + b_wofs + t * B_RPP * RSB + k * PLB);
            else load_b(set, t);
        }
    };

    f32x16 acc[NTW][MT][NT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

    const int q = lane & 15, gk = lane >> 5, hh = (lane >> 4) & 1;
    const int a_foff = (8 * gk + (q >> 2)) * RSA + (wm * WM + 16 * hh + 4 * (q & 3)) * 2;
    const int b_foff = 3 * PLA + (8 * gk + (q >> 2)) * RSB + (wn * WN + 16 * hh + 4 * (q & 3)) * 2;
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm

    // chunk c lives in register set (c - kc0) & 1: loaded two K steps before it is contracted, split one step before
    if (kc1 > kc0) {
        chunk_base(kc0);
#pragma unroll
        for (int u = 3; u < NUNIT; u += 4) unit(0, u, smem);
#pragma unroll
        for (int u = 0; u < NUNIT; ++u) if ((u & 3) != 3) unit(0, u, smem);
        chunk_base(kc0 + 1);
#pragma unroll
        for (int u = 3; u < NUNIT; u += 4) unit(1, u, smem);
        chunk_base(kc0 + 2);
#pragma unroll
        for (int u = 3; u < NUNIT; u += 4) unit(0, u, smem);
    }
    lds_barrier();
    int stage = 0;
    constexpr int NM1 = 6 * MT * NT, NMG = NTW * NM1;          // MFMAs per tap / per K step
    constexpr int USTRIDE = NMG / NUNIT > 0 ? NMG / NUNIT : 1;
    constexpr int NFA = 3 * MT;
    static_assert(NFA <= NM1, "a tap has enough MFMA slots for the next tap's fragment reads");
    auto step = [&](int kc, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;          // register set of chunk kc + 1 (split now) = of chunk kc + 3 (loaded now)
        const char* st = smem + stage * ST;
        char* st_next = smem + (stage ^ 1) * ST;
        chunk_base(kc + 3);
        bf16x8 fb[3][NT];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[pl][j] = tr_frag(st + b_foff + pl * PLB + j * 64, RSB);
        {
            bf16x8 fa[2][3][MT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[0][pl][i] = tr_frag(st + a_foff + pl * PLA + i * 64, RSA);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int cb = t & 1;
#pragma unroll
                for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int k = (tt * MT + i) * NT + j, idx = t * NM1 + k;
                            __builtin_amdgcn_sched_barrier(0);
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][TA[tt]][i], fb[TB[tt]][j], acc[t][i][j], 0, 0, 0);
                            // side jobs of this slot: the next tap's G fragments, one staging unit
                            if (t + 1 < NTW && k < NFA) {
                                const int pl = k / MT, fi = k - pl * MT;
                                fa[cb ^ 1][pl][fi] = tr_frag(st + a_foff + (t + 1) * RSA + pl * PLA + fi * 64, RSA);
                            }
                            if (idx % USTRIDE == 0 && idx / USTRIDE < NUNIT) unit(SET, idx / USTRIDE, st_next);
                        }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = (NMG + USTRIDE - 1) / USTRIDE; u < NUNIT; ++u) unit(SET, u, st_next);      // (units that found no slot)
        lds_barrier();
        stage ^= 1;
    };
    int kc = kc0;
    for (; kc + 1 < kc1; kc += 2) {
        step(kc, std::integral_constant<int, 1>{});
        step(kc + 1, std::integral_constant<int, 0>{});
    }
    if (kc < kc1) step(kc, std::integral_constant<int, 1>{});

    const int li = lane & 31, kk = lane >> 5;
    float* out = d.splitk > 1 ? d.ws + (size_t)z * d.TH * NTW * d.Cg * d.Cd : d.out;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = g0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (g >= d.Cg) continue;
                float* orow = out + ((size_t)(th * NTW + t) * d.Cg + g) * d.Cd;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int dd = d0 + wn * WN + j * 32 + li;
                    if (dd < d.Cd) orow[dd] = acc[t][i][j][e];
                }
            }
}

// plain reference of the same contraction (one thread per output element, fp64 accumulation): SAGEN_WGRAD_REF=1 routes every
// weight gradient through it - a debugging aid that isolates the MFMA kernel from the rest of the backward pass
__global__ __launch_bounds__(256) void wgrad_ref_kernel(const WgradDesc d) {
    const long total = (long)d.TH * d.TW * d.Cg * d.Cd;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int dd = (int)(idx % d.Cd);
    long r = idx / d.Cd;
    const int g = (int)(r % d.Cg);
    const int tap = (int)(r / d.Cg);
    const int th = tap / d.TW, tw = tap - th * d.TW;
    double acc = 0.0;
    for (int b = 0; b < d.B; ++b)
        for (int i = 0; i < d.Hd; ++i) {
            const int gi = i * d.sh + th * d.tsh + d.h0;
            if ((unsigned)gi >= (unsigned)d.HG) continue;
            for (int j = 0; j < d.Wd; ++j) {
                const int gj = j * d.sw + tw * d.tsw + d.w0;
                if ((unsigned)gj >= (unsigned)d.WG) continue;
                acc += (double)d.g[(size_t)b * d.g_bstride + (size_t)gi * d.g_rstride + (size_t)gj * d.ldg + g] *
                       (double)d.d[(size_t)b * d.d_bstride + (size_t)i * d.d_rstride + (size_t)j * d.ldd + dd];
            }
        }
    d.out[idx] = (float)acc;
}

// SAGEN_FP32_ONLY / SAGEN_WGRAD_F32: the exact fp32 MFMA kernel; default: the bf16x3 kernel (fp32-equivalent, finite inputs)
static bool wgrad_exact() {
    static const bool exact = getenv("SAGEN_FP32_ONLY") != nullptr || getenv("SAGEN_WGRAD_F32") != nullptr;
    return exact;
}
static bool wgrad_rowtap(const WgradDesc& d);
static bool wgrad_planes(const WgradDesc& d);
int wgrad3h_dispatch(const WgradDesc& d, int bm, unsigned blocks, hipStream_t s);      // (wgrad3h.hip)
const char* wgrad_kernel_name(const WgradDesc& d) {
    static const bool ref = getenv("SAGEN_WGRAD_REF") != nullptr;
    return ref ? "wgrad_ref_kernel" : (wgrad_exact() ? "wgrad_kernel" : (wgrad_planes(d) ? "wgrad3h_kernel" : (wgrad_rowtap(d) ? "wgrad3r_kernel" : "wgrad3_kernel")));
}

static unsigned magic_of(int dv) { return dv <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1ull); }

size_t wgrad_ws_floats(const WgradDesc& d, int splitk) { return splitk > 1 ? (size_t)splitk * d.TH * d.TW * d.Cg * d.Cd : 0; }

// the filter-row kernel runs dense 3-wide stride-1 rows with SAME padding (one pad pixel either side), rows of >= 12 pixels
static bool wgrad_rowtap(const WgradDesc& d) {
    static const bool off = getenv("SAGEN_WGRAD_NOROW") != nullptr;
    return !off && !wgrad_exact() && d.TW == 3 && d.tsw == 1 && d.sw == 1 && d.w0 == -1 && d.WG == d.Wd && d.Wd >= 12;
}
// ... and on fp16x2 planes when the caller has both operands in that form: 3x3, SAME in both directions, channels in whole 32s
bool wgrad_planes_enabled() {
    static const bool off = getenv("SAGEN_WGRAD_NO_H2") != nullptr || getenv("SAGEN_WGRAD_NOROW") != nullptr;
    return !off && !wgrad_exact();
}
bool wgrad_reads_fp32_operands() {          // the reference kernel reads the fp32 tensors whatever planes the descriptor carries
    static const bool ref = getenv("SAGEN_WGRAD_REF") != nullptr;
    return ref;
}
static bool wgrad_planes(const WgradDesc& d) {
    return wgrad_planes_enabled() && wgrad_rowtap(d) && d.gp && d.dp && d.gp_a_inv && d.dp_a_inv && d.TH == 3 && d.tsh == 1 && d.sh == 1 && d.h0 == -1 &&
           d.HG == d.Hd && d.Cg % 32 == 0 && d.Cd % 32 == 0;
}
struct WgradShape { int bm, bn, fold; long ntile, nchunks; bool row; };
static WgradShape wgrad_shape(const WgradDesc& d) {
    WgradShape w;
    w.row = wgrad_rowtap(d);
    w.bm = d.Cg > 64 ? 128 : 64;
    w.bn = (d.Cd > 64 && !w.row) ? 128 : 64;                      // (three accumulator sets: the row kernel keeps BN = 64)
    const long P = w.row ? (long)d.B * d.Hd * (d.Wd + 1) : (long)d.B * d.Hd * d.Wd;
    // 32 / 16 gathered channels, a column of filter rows (the stem / the audio conv1 after their horizontal taps were folded into the
    // channels): four / eight rows per tile
    static const bool nofold = getenv("SAGEN_WGRAD_NOFOLD") != nullptr;
    w.fold = (!nofold && !wgrad_exact() && !w.row && d.TW == 1 && d.TH >= 2 && (d.Cg == 32 || d.Cg == 16)) ? 128 / d.Cg : 1;
    if (w.fold > 1) w.bm = 128;
    w.ntile = w.fold > 1 ? (long)cdiv(d.TH, w.fold) * cdiv(d.Cd, w.bn)
                         : (long)d.TH * (w.row ? 1 : d.TW) * cdiv(d.Cg, w.bm) * cdiv(d.Cd, w.bn);
    w.nchunks = (P + 15) / 16;
    return w;
}

int wgrad_pick_splitk(const WgradDesc& d, size_t ws_capacity_floats) {
    const WgradShape w = wgrad_shape(d);
    // ~1024 workgroups (two rounds of the 2-per-CU slots for the 128x128 tile, never a thin third one), >= 8 chunks per range
    // Workgroups: two rounds of the 2-per-CU slots (1024) hide prologues and tails best - unless the partial sums become the cost:
    // every workgroup writes its whole tile and the reducer reads it back, 2 x (workgroups x tile bytes) of traffic.  Measured
    // on the 3x3 layers of stages 3-5 (2.4-9.4 MB of gradient, 64-98 KB per tile): 110 us with 1024 workgroups, of which 60 us is
    // partial-sum traffic, 94 us with 512; the stem (57 KB of gradient) 344 us with 1024, 467 us with 512.
    static const long forced = getenv("SAGEN_WGRAD_WGS") ? std::max(64, atoi(getenv("SAGEN_WGRAD_WGS"))) : 0;
    const size_t per = (size_t)d.TH * d.TW * d.Cg * d.Cd;
    const long target = forced ? forced : ((1024 / w.ntile) * per * 4 > (32u << 20) ? 512 : 1024);
    long sk = std::max<long>(1, std::min<long>(target / w.ntile, w.nchunks / 8));
    while (sk > 1 && sk * per > ws_capacity_floats) --sk;
    return (int)std::min<long>(sk, 256);
}

int wgrad_launch(const WgradDesc& d_in, hipStream_t s) {
    WgradDesc d = d_in;
    if (!d.g || !d.d || !d.out) return fail(SAGEN_ERR_NULL, "wgrad: null operand");
    if (d.B <= 0 || d.Hd <= 0 || d.Wd <= 0 || d.Cg <= 0 || d.Cd <= 0 || d.TH <= 0 || d.TW <= 0 || d.sh <= 0 || d.sw <= 0)
        return fail(SAGEN_ERR_SHAPE, "wgrad: bad dimensions");
    if (d.Cg % 4 || d.ldg <= 0 || d.ldd % 4 || (d.Cd % 4 && d.ldd < (d.Cd + 3) / 4 * 4))
        return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: Cg=%d must be a multiple of 4, D rows must be padded to a multiple of 4 floats (Cd=%d ldd=%d)", d.Cg, d.Cd, d.ldd);
    if (((uintptr_t)d.g | (uintptr_t)d.d) % 16) return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: operands must be 16-byte aligned");
    const WgradShape shp = wgrad_shape(d);
    const long P = shp.row ? (long)d.B * d.Hd * (d.Wd + 1) : (long)d.B * d.Hd * d.Wd;      // (the row kernel contracts over the padded grid)
    const long gb = ((long)(d.B - 1) * d.g_bstride + (long)d.HG * d.g_rstride) * 4 + 64, db = ((long)(d.B - 1) * d.d_bstride + (long)d.Hd * d.d_rstride) * 4 + 64;
    if (gb >= (1L << 31) || db >= (1L << 31) || P >= (1L << 24) || (long)P * (d.Wd + 1) >= (1L << 32))
        return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: operand exceeds 2 GiB buffer addressing / 2^24 pixels (use a smaller batch)");
    const unsigned lim = 1u << 24;
    if (d.g_bstride >= lim || d.g_rstride >= lim || d.d_bstride >= lim || d.d_rstride >= lim || (unsigned)d.ldg >= lim || (unsigned)d.ldd >= lim ||
        (unsigned)d.HG >= lim || (unsigned)d.WG >= lim)
        return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: a stride / extent exceeds 2^24 (24-bit index multiplies)");
    d.P = (int)P;
    d.g_bytes = (unsigned)gb; d.d_bytes = (unsigned)db;
    d.magic_w = magic_of(shp.row ? d.Wd + 1 : d.Wd); d.magic_h = magic_of(d.Hd);
    d.fold = shp.fold;
    static const bool use_ref = getenv("SAGEN_WGRAD_REF") != nullptr;
    const long total = (long)d.TH * d.TW * d.Cg * d.Cd;
    if (use_ref) {
        hipLaunchKernelGGL(wgrad_ref_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, d);
        SAGEN_LAUNCH_CHECK();
        return SAGEN_OK;
    }
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > 1 && !d.ws) return fail(SAGEN_ERR_WORKSPACE, "wgrad: split-K needs a workspace");
    const int bm = shp.bm, bn = shp.bn;
    const long blocks = shp.ntile * d.splitk;
    if (blocks >= (1L << 30)) return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: grid too large");
    const dim3 grid((unsigned)blocks);
    if (wgrad_exact()) {
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((wgrad_kernel<128, 128>), grid, dim3(256), 0, s, d);
        else if (bm == 128) hipLaunchKernelGGL((wgrad_kernel<128, 64>), grid, dim3(256), 0, s, d);
        else if (bn == 128) hipLaunchKernelGGL((wgrad_kernel<64, 128>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, dim3(256), 0, s, d);
    } else if (wgrad_planes(d)) {
        const size_t pb = (size_t)P * 64;
        if ((size_t)(d.Cg / 16) * pb >= (1ull << 31) || (size_t)(d.Cd / 16) * pb >= (1ull << 31))
            return fail(SAGEN_ERR_UNSUPPORTED, "wgrad: plane operand exceeds 2 GiB buffer addressing");
        d.gp_bytes = (unsigned)((d.Cg / 16) * pb); d.dp_bytes = (unsigned)((d.Cd / 16) * pb);
        const int rc = wgrad3h_dispatch(d, bm, (unsigned)blocks, s);
        if (rc) return rc;
    } else if (shp.row) {
        if (bm == 128) hipLaunchKernelGGL((wgrad3r_kernel<128, 64>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((wgrad3r_kernel<64, 64>), grid, dim3(256), 0, s, d);
    } else {
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((wgrad3_kernel<128, 128>), grid, dim3(256), 0, s, d);
        else if (bm == 128) hipLaunchKernelGGL((wgrad3_kernel<128, 64>), grid, dim3(256), 0, s, d);
        else if (bn == 128) hipLaunchKernelGGL((wgrad3_kernel<64, 128>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((wgrad3_kernel<64, 64>), grid, dim3(256), 0, s, d);
    }
    SAGEN_LAUNCH_CHECK();
    if (d.splitk > 1 && !d.defer_reduce) return wgrad_reduce_launch(d, s);
    return SAGEN_OK;
}

// the fixed-order sum of the pixel-range partials (wgrad_launch runs it itself unless d.defer_reduce)
int wgrad_reduce_launch(const WgradDesc& d, hipStream_t s) {
    static const bool use_ref = getenv("SAGEN_WGRAD_REF") != nullptr;
    if (d.splitk <= 1 || use_ref) return SAGEN_OK;
    return splitk_reduce_launch(d.ws, d.splitk, d.TH * d.TW * d.Cg, d.Cd, nullptr, 0, d.out, d.Cd, 1, nullptr, s);
}

}  // namespace sagen
