// wgrad3h_kernel: the weight gradient of the dense 3x3 stride-1 SAME convolutions of the ResNet trunks (what tf.gradients builds for
// tf.nn.convolution, core.py:206, under train.py:147-149) on fp16x2 PLANES of both operands - three v_mfma_f32_32x32x16_f16 products
// per fp32 product (hi*hi + hi*lo + lo*hi, conv3h.hip) instead of the six bf16 products of wgrad3r_kernel, and no operand split in
// the kernel at all: both operands already exist as planes when the backward pass gets here -
//   G = the layer's input activation: the training forward wrote its planes for conv3h_kernel (p3.hip) and now RETAINS them;
//   D = dy: the batch-norm backward writes it as planes for the data gradient (backward.hip: bn_bwd_apply_h2_kernel).
// Both are [C/16][NP][2][16] fp16 over the padded pixel grid NP = B*H*(W+1) (one zero pixel closing every image row) - the very grid
// wgrad3r_kernel contracts over: dW[th][tw][g][d] = sum_q G[q + (th-1)(W+1) + (tw-1)][g] * D[q][d], the pad pixel being the
// right-hand padding of its row and the left-hand padding of the next; rows outside the image are range-check zeros of the DMA.
//
// Workgroup = (filter row th, BM channels of G, BN of D, a range of 16-pixel K steps).  Per step both tiles go global -> LDS by
// LDS-DMA (1 KiB per wave-instruction: 16 pixels x 32 channels of one plane), into a RING of five steps per (32-channel pair,
// plane): 80 pixel rows x 64 B, contiguous in the pixel index; the chunk of step c + 4 is issued during step c and the step waits
// with vmcnt(2 chunks' worth) - a round trip to memory is several 16-pixel steps long.  G is staged ONCE for the three horizontal
// taps, as the stream u -> G[u + (th-1)(W+1) - 1]: tap tw of step c reads stream rows 16c + tw .. + 15, i.e. two rows into the NEXT step's chunk, which
// the ring keeps adjacent (the wrap at the last row is a per-lane constant).  The matrix cores want 8 consecutive pixels of one channel
// per lane: ds_read_b64_tr_b16 delivers them from the [pixel][channel] image (lane mapping: wgrad.hip, tools/probe/tr16.py); a
// lane group reads THREE 4-pixel blocks (12 rows) per operand tile and step, and the fragments of taps 1 and 2 are cut out of
// those registers (one v_perm per dword for the odd shift) instead of being read again.  The product is scaled by 2^-(ka + kd) in the
// epilogue (exact).  Split-K partials and their fixed-order reduction as in wgrad.hip.
#include "igemm3_common.h"

namespace sagen {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
typedef unsigned u32x4h __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x2h tr4(const char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    return __builtin_bit_cast(u32x2h, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p)));
#else
    return u32x2h{};
#endif
}

// LDS-DMA as inline assembly: hipcc drains vmcnt to ZERO in front of every LDS read that follows a builtin LDS-DMA it cannot prove
// disjoint (SIInsertWaitcnts) - with the DMA of chunk c + 4 issued at the top of step c that turned the ring into a one-step round
// trip per step (60 us per layer whatever the lead).  Hidden in asm, the loads are ordered by this kernel's own s_waitcnt vmcnt(N)
// + barrier alone.  rsrc = {base lo, base hi, bytes, 0x00020000} (raw buffer, range-checked), lds = byte address of lane 0's 16 B.
__device__ __forceinline__ void dma16_raw(u32x4h rsrc, unsigned lds, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
#endif
}
__device__ __forceinline__ unsigned lds_address(const char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) const char*)p);
#else
    return 0u;
#endif
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad3h_kernel(const WgradDesc d) {
    constexpr int LEAD = 4, RING = LEAD + 1, NTW = 3;          // chunk c + LEAD is issued during step c: the ring holds chunks c .. c + LEAD
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;                  // 32-channel pairs of the G / D tile
    constexpr int PLANE = RING * 16 * 64;                      // one (pair, plane) ring: RING x 16 pixel rows x 64 B
    constexpr int G_BYTES = AP * 2 * PLANE, D_BYTES = BP * 2 * PLANE;
    constexpr int NI = 2 * (AP + BP);                          // LDS-DMA wave-instructions per K step
    constexpr int IPW = (NI + 3) / 4;
    __shared__ __attribute__((aligned(16))) char smem[G_BYTES + D_BYTES];      // ONE shared object (conv3p.hip)

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
    const int Wp = d.Wd + 1, H = d.Hd, NP = d.P;
    const int off = (th - 1) * Wp - 1;                         // stream position u of G <-> padded pixel u + off

    const int nchunks = (NP + 15) >> 4;
    const int per_z = (nchunks + d.splitk - 1) / d.splitk;
    const int kc0 = z * per_z;
    const int kc1 = min(nchunks, kc0 + per_z);

    const unsigned long long gpa = (unsigned long long)d.gp, dpa = (unsigned long long)d.dp;
    const u32x4h g_rsrc = {(unsigned)gpa, (unsigned)(gpa >> 32) & 0xffffu, d.gp_bytes, 0x00020000u};
    const u32x4h d_rsrc = {(unsigned)dpa, (unsigned)(dpa >> 32) & 0xffffu, d.dp_bytes, 0x00020000u};
    const unsigned smem_base = lds_address(smem);
    const unsigned cstride = (unsigned)NP * 64u;

    // ---- per-lane DMA state: lane = (pixel row of the step, 16-byte piece of the 64-byte LDS row = (chunk parity, half)) ----
    const int drow = lane >> 2, dcp = (lane >> 1) & 1, dhalf = lane & 1;
    unsigned i_base[IPW];                                      // byte offset of this lane's piece at pixel 0 (OOB: channel chunk outside)
    int i_dst[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int I = wave + 4 * j;
        const bool isg = I < 2 * AP;
        const int idx = isg ? I : I - 2 * AP;
        const int pair = idx >> 1, plane = idx & 1;
        const int chunk = ((isg ? g0 : d0) >> 4) + 2 * pair + dcp;
        const bool ok = I < NI && chunk * 16 < (isg ? d.Cg : d.Cd);
        i_base[j] = ok ? (unsigned)chunk * cstride + (unsigned)(plane * 32 + dhalf * 16) : OOB;
        i_dst[j] = __builtin_amdgcn_readfirstlane((isg ? 0 : G_BYTES) + (pair * 2 + plane) * PLANE);
    }
    auto issue = [&](int c, int stage) {
        const int pd = 16 * c + drow;                          // D pixel of this lane
        const int pg = pd + off;                               // G pixel
        bool okg = (unsigned)pg < (unsigned)NP;
        {
            const unsigned r0 = __umulhi((unsigned)pg, d.magic_w);             // pg / Wp = b*H + i
            const unsigned ip = r0 - __umulhi(r0, d.magic_h) * (unsigned)H;
            okg = okg && !((th == 0 && ip == (unsigned)(H - 1)) || (th == 2 && ip == 0u));      // the row above / below lies in another image
        }
        const bool okd = pd < NP;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int I = wave + 4 * j;
            if (NI % 4 != 0 && I >= NI) continue;
            const unsigned dst = smem_base + (unsigned)(i_dst[j] + stage * 1024);
            if (I < 2 * AP) dma16_raw(g_rsrc, dst, okg ? (i_base[j] + (unsigned)pg * 64u) | (i_base[j] & OOB) : OOB);
            else dma16_raw(d_rsrc, dst, okd ? (i_base[j] + (unsigned)pd * 64u) | (i_base[j] & OOB) : OOB);
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

    // fragment addressing: lane group (lane >> 4) = (k half gk, channel half hh); source lane q: row q/4, channels 4(q%4)..+3
    const int q = lane & 15, gk = lane >> 5, hh = (lane >> 4) & 1;
    const int f_lane = (8 * gk + (q >> 2)) * 64 + 32 * hh + 8 * (q & 3);
    const int a_foff = (wm * MT * 2) * PLANE + f_lane;
    const int b_foff = G_BYTES + (wn * NT * 2) * PLANE + f_lane;
    const int wrapfix = gk ? PLANE : 0;                        // the third block of the ring's last step wraps to row 0

    // Blocks 0, 1 of the G fragments and the D fragments of a step lie inside the step's own chunk: they are read during the MFMAs of
    // the PREVIOUS step (`ca`, `cb`); only block 2 (rows 8..11 of the lane group: for the upper k half the first rows of the next
    // chunk) is read after the barrier, under the six MFMAs of tap 0, which does not need it.
    u32x2h ca[2][MT][2], cb[2][NT][2];
    auto read_a = [&](int S, int pl, int i, int blk) { return tr4(smem + a_foff + (i * 2 + pl) * PLANE + S * 1024 + blk * 256); };
    auto read_b = [&](int S, int pl, int j, int blk) { return tr4(smem + b_foff + (j * 2 + pl) * PLANE + S * 1024 + blk * 256); };
    constexpr int NPRE = 4 * (MT + NT);                        // prefetch reads per step, RPS behind each MFMA of taps 1 and 2
    constexpr int RPS = (NPRE + 6 * MT * NT - 1) / (6 * MT * NT);
    auto step = [&](int c, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value, SN = (S + 1) % RING;
        wait_vmcnt<(LEAD - 2) * IPW>();                        // all but the LEAD - 2 youngest chunks: chunks c and c + 1 have landed ...
        lds_barrier();                                         // ... for every wave; nobody reads chunk c - 1 any more
        issue(c + LEAD, (S + LEAD) % RING);
        u32x2h a2[2][MT];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const char* p = smem + a_foff + (i * 2 + pl) * PLANE + S * 1024 + 2 * 256;
                a2[pl][i] = tr4(S == RING - 1 ? p - wrapfix : p);
            }
        f16x8 fb[2][NT];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[pl][j] = __builtin_bit_cast(f16x8, u32x4h{cb[pl][j][0][0], cb[pl][j][0][1], cb[pl][j][1][0], cb[pl][j][1][1]});
        u32x2h na[2][MT][2], nb[2][NT][2];
        int slot = 0;                                          // prefetch reads placed so far
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f16x8 fa[2][MT];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const unsigned w0 = ca[pl][i][0][0], w1 = ca[pl][i][0][1], w2 = ca[pl][i][1][0], w3 = ca[pl][i][1][1], w4 = a2[pl][i][0];
                    u32x4h v;
                    if (t == 0) v = u32x4h{w0, w1, w2, w3};
                    else if (t == 2) v = u32x4h{w1, w2, w3, w4};
                    else v = u32x4h{__builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 16),
                                    __builtin_amdgcn_alignbit(w3, w2, 16), __builtin_amdgcn_alignbit(w4, w3, 16)};
                    fa[pl][i] = __builtin_bit_cast(f16x8, v);
                }
            constexpr int TA[3] = {0, 0, 1}, TB[3] = {0, 1, 0};              // hi*hi, hi*lo, lo*hi
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        __builtin_amdgcn_sched_barrier(0);
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[TA[tt]][i], fb[TB[tt]][j], acc[t][i][j], 0, 0, 0);
                        if (t > 0) {
#pragma unroll
                            for (int u = 0; u < RPS; ++u, ++slot) {
                                if (slot < 4 * MT) { const int r = slot; na[r & 1][(r >> 1) % MT][r / (2 * MT)] = read_a(SN, r & 1, (r >> 1) % MT, r / (2 * MT)); }
                                else if (slot < NPRE) { const int r = slot - 4 * MT; nb[r & 1][(r >> 1) % NT][r / (2 * NT)] = read_b(SN, r & 1, (r >> 1) % NT, r / (2 * NT)); }
                            }
                        }
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) { ca[pl][i][0] = na[pl][i][0]; ca[pl][i][1] = na[pl][i][1]; }
#pragma unroll
            for (int j = 0; j < NT; ++j) { cb[pl][j][0] = nb[pl][j][0]; cb[pl][j][1] = nb[pl][j][1]; }
        }
    };

    if (kc1 > kc0) {
#pragma unroll
        for (int l = 0; l < LEAD; ++l) issue(kc0 + l, l);
        wait_vmcnt<(LEAD - 1) * IPW>();                        // chunk kc0
        lds_barrier();
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) { ca[pl][i][0] = read_a(0, pl, i, 0); ca[pl][i][1] = read_a(0, pl, i, 1); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { cb[pl][j][0] = read_b(0, pl, j, 0); cb[pl][j][1] = read_b(0, pl, j, 1); }
        }
    }
    static_assert(RING == 5 && (LEAD - 1) * IPW <= 9, "the K loop below is unrolled over a ring of five; wait_vmcnt covers the counts");
    int c = kc0;
    for (; c + 4 < kc1; c += 5) {
        step(c, std::integral_constant<int, 0>{});
        step(c + 1, std::integral_constant<int, 1>{});
        step(c + 2, std::integral_constant<int, 2>{});
        step(c + 3, std::integral_constant<int, 3>{});
        step(c + 4, std::integral_constant<int, 4>{});
    }
    if (c < kc1) step(c, std::integral_constant<int, 0>{});
    if (c + 1 < kc1) step(c + 1, std::integral_constant<int, 1>{});
    if (c + 2 < kc1) step(c + 2, std::integral_constant<int, 2>{});
    if (c + 3 < kc1) step(c + 3, std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last steps' look-ahead DMAs: nothing may land after the workgroup has gone)

    const float sc = d.gp_a_inv[0] * d.dp_a_inv[0];
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
                    if (dd < d.Cd) orow[dd] = acc[t][i][j][e] * sc;
                }
            }
}

int wgrad3h_dispatch(const WgradDesc& d, int bm, unsigned blocks, hipStream_t s) {
    const dim3 grid(blocks);
    if (bm == 128) hipLaunchKernelGGL((wgrad3h_kernel<128, 64>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((wgrad3h_kernel<64, 64>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
