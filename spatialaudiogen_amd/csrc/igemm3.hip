// fp32-equivalent implicit-GEMM convolution / transposed convolution / FC on the gfx950 bf16 matrix cores ("bf16x3").
//
// Same contraction, operands, geometry and epilogue as igemm.hip (reference ops: tf.nn.convolution core.py:206,
// tf.nn.conv2d_transpose core.py:140, tf.matmul core.py:79) - only the arithmetic of the K loop differs:
//
//   every fp32 operand v is written as the sum of three bf16 numbers (8 + 8 + 8 mantissa bits)
//       hi = rne(v),  mid = rne(v - hi),  lo = rne(v - hi - mid)            (v - hi and v - hi - mid are exact in fp32)
//   and a*b is evaluated as the six products of weight >= 2^-16
//       mid*mid + hi*lo + lo*hi + hi*mid + mid*hi + hi*hi
//   on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped products (mid*lo, lo*mid, lo*lo) are below
//   2^-24 relative - fp32 rounding level; measured against an fp64 reference the end-to-end error of this path is not
//   larger than that of the exact fp32 MFMA path.  Finite inputs only: inf - inf in the residual gives NaN.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the VECTOR rate (64 FLOP/clk/SIMD, 157 TF/chip) and excludes VALU work on its
// SIMD while it runs; six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 32x32x16 sub-tile
// (2.67x fewer matrix-pipe cycles, roof 2.5 PF / 6 = 417 TF fp32-equivalent) and overlap with VALU.
//
// Structure (CDNA4):
//  * filters are split ONCE at bind time into three bf16 planes, tiled [K/16][plane][n][16]: the tile of one K step
//    is one contiguous block, fetched with fully coalesced buffer_load_dwordx4 and stored with ds_write_b128.
//    (No LDS-DMA here: the compiler has to assume that a DMA in flight aliases every later ds_read / ds_write and
//    drains vmcnt to 0 before them, which would serialise the register-staged activation path below.)
//  * activations are staged THROUGH REGISTERS, one 16-B chunk (4 channels of one output row and tap) per thread:
//    buffer_load_dwordx4 (padding / row / k tails zero-filled by the buffer range check) -> the producer's
//    training-mode batch-norm + ReLU (optional) -> bf16x3 split -> three ds_write_b64 into the A planes.  Each
//    element is split once per workgroup, and the split VALU is interleaved with the MFMAs of the previous K tile.
//  * the consumer side is ds_read_b128 + MFMA only: lane (i, g) reads the 8 bf16 k = 8g..8g+7 of row i from each
//    plane; 32-B LDS rows, 16-B halves swapped by (row >> 3) & 1 -> conflict-free reads, linear writes.
//  * two LDS stages; all global loads run two K tiles ahead (two register sets, no copies: the K loop is unrolled
//    by two), are converted / stored one tile ahead, and every job is unconditional (tiles past the end are
//    zero-filled by the range check) so that a K step is one basic block with compiler-counted vmcnt waits.
#include "igemm_common.h"
#include <type_traits>

namespace sagen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
#else
    return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}

// (a, b) -> packed bf16 pair (round to nearest even) and the exact fp32 residuals
__device__ __forceinline__ unsigned split_pair(float& a, float& b) {
    const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    a -= __builtin_bit_cast(float, pk << 16);
    b -= __builtin_bit_cast(float, pk & 0xffff0000u);
    return pk;
}

// KS = K tiles of 16 per barrier step (small tiles amortise the barrier and the loop overhead over 2 tiles);
// PRO = the producer's batch-norm + ReLU is applied to the activations on the way in.
template <int BM, int BN, int WM, int WN, int KS, bool PRO>
__device__ __forceinline__ void igemm3_body(const IgemmDesc& d) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int BK = 16;
    constexpr int NCH = BM >= 64 ? BM / 64 : 1;        // 16-B activation chunks per thread per K tile (BM*4 chunks / 256 threads)
    constexpr int NBC = (6 * BN + 255) / 256;           // 16-B filter chunks per thread per K tile (3 planes x BN rows x 2)
    constexpr int A_PL = BM * 8, B_PL = BN * 8;         // floats per plane per stage (rows of 32 B)
    constexpr int SUB_F = 3 * A_PL + 3 * B_PL;          // one K tile of 16: three A planes + three filter planes
    constexpr int STAGE_F = KS * SUB_F;
    constexpr int NJOB = NBC + NBC + NCH + 3 * NCH;     // side jobs per K tile: filter stores / loads, activation loads, convert jobs

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_F];
    __shared__ RowInfo s_row[BM];
    __shared__ int s_tapb[MAX_TAPS];
    __shared__ __attribute__((aligned(16))) float s_bn[2][MAX_BN_C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware M-tile remap (see igemm.hip)
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
    const bool uni = d.uniform_taps != 0;

    igemm_setup<BM>(d, m0, tid, uni, s_row, s_tapb, s_bn);

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.x, 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =       // bf16 planes follow the fp32 filter
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // ---- activation loader: chunk q = tid + 256*c -> row q/4, k offset 4*(q%4) ----
    const int kc4 = tid & 3;
    unsigned a_voff[NCH], a_nmlo[NCH], a_nmhi[NCH];
    int a_wofs[NCH];                                  // float offset inside an A plane
    const bool a_active = BM >= 64 || tid < BM * 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int row = (tid + 256 * c) >> 2;
        a_voff[c] = OOB; a_nmlo[c] = 0xffffffffu; a_nmhi[c] = 0xffffffffu; a_wofs[c] = 0;
        if (a_active) {
            const RowInfo ri = s_row[row];
            a_voff[c] = ri.boff + 16u * kc4;
            a_nmlo[c] = ri.nmlo; a_nmhi[c] = ri.nmhi;
            a_wofs[c] = row * 8 + 4 * ((kc4 >> 1) ^ ((row >> 3) & 1)) + 2 * (kc4 & 1);
        }
    }
    // ---- filter loader: chunk q = tid + 256*c -> plane q / (2*BN), row (q % (2*BN)) / 2, 16-B half q & 1 ----
    unsigned b_voff[NBC];
    int b_wofs[NBC];
    bool b_active[NBC];
#pragma unroll
    for (int c = 0; c < NBC; ++c) {
        const int q = tid + 256 * c;
        const int pl = q / (2 * BN), rem = q - pl * (2 * BN);
        const int r = rem >> 1, half = rem & 1;
        const int n = n0 + r;
        b_active[c] = q < 6 * BN;
        b_voff[c] = (b_active[c] && n < d.N) ? (unsigned)(((long)pl * d.N + n) * 32 + 16 * half) : OOB;
        b_wofs[c] = 3 * A_PL + pl * B_PL + r * 8 + 4 * (half ^ ((r >> 3) & 1));
    }

    const int nk = d.Kpad / BK;
    const int nk_per = (nk + d.splitk - 1) / d.splitk;
    const int kc0 = z * nk_per;
    const int kc1 = min(nk, kc0 + nk_per);
    const int ntiles = kc1 - kc0;

    // SGPR tracker of the (tap, channel) position of the next tile to LOAD
    int q_tap = 0, q_th = 0, q_tw = 0, q_c0 = 0;
    if (uni) {
        const int k0 = kc0 * BK;
        if (d.ntaps > 1) { q_tap = k0 >> d.log2Cin; q_c0 = k0 & (d.Cin - 1); q_th = q_tap / d.TW; q_tw = q_tap - q_th * d.TW; }
        else q_c0 = k0;
    }

    // two register sets for the chunks in flight: the tiles of step t+1 (landed, converted / stored during step t)
    // sit in set (t+1)&1 while the loads of step t+2 fill set t&1
    f32x4 araw[2][KS][NCH], braw[2][KS][NBC];
    unsigned abad[2][KS][NCH];
    int ac0[2][KS];                                   // first channel of each tile in flight (batch-norm coefficients)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ac0[p][ks] = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { araw[p][ks][c] = f32x4{0.f, 0.f, 0.f, 0.f}; abad[p][ks][c] = 1; }
#pragma unroll
            for (int c = 0; c < NBC; ++c) braw[p][ks][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // per-tile load state (wave-uniform in the uniform-tap mode); `past` = 1 for tiles beyond this split's range
    unsigned i_tb = 0, i_bit = 0, i_kok = 1, i_past = 0, i_kbyte = 0;
    bool i_hi = false;
    auto begin_load = [&](int kc, int& c0_out) {
        i_past = kc >= kc1 ? 1u : 0u;
        i_kbyte = (unsigned)kc * (unsigned)(d.N * 96);
        if (uni) {
            i_tb = (unsigned)((((q_th * d.tap_sh + d.tap_h0) * d.Win + (q_tw * d.tap_sw + d.tap_w0)) * d.ldx + q_c0) * 4);
            i_bit = (unsigned)(q_tap & 31);
            i_hi = q_tap >= 32;
            c0_out = q_c0;
            q_c0 += BK;
            if (d.ntaps > 1 && q_c0 == d.Cin) {
                q_c0 = 0; ++q_tap; ++q_tw;
                if (q_tw == d.TW) { q_tw = 0; ++q_th; }
            }
        } else {       // per-chunk tap (Cin < 16 or a ragged K tail): table lookup
            const int k = kc * BK + 4 * kc4;
            const bool kok = k < d.K;
            const int tap = (kok && d.ntaps > 1) ? (k >> d.log2Cin) : 0;
            const int cch = d.ntaps > 1 ? (k & (d.Cin - 1)) : k;
            i_tb = (unsigned)(s_tapb[tap] + 4 * cch) - 16u * kc4;
            i_bit = (unsigned)(tap & 31);
            i_hi = tap >= 32;
            i_kok = kok ? 1u : 0u;
        }
    };
    auto load_a = [&](int c, f32x4& dst, unsigned& bad_out) {
        const unsigned word = i_hi ? a_nmhi[c] : a_nmlo[c];
        unsigned bad = ((word >> i_bit) & 1u) | i_past;
        if (!uni) bad |= (i_kok ^ 1u);
        bad_out = bad;
#ifndef SAGEN_ABLATE_A
        dst = bload16(x_rsrc, (a_voff[c] + i_tb) | (bad << 31));
#endif
    };
    auto load_b = [&](int c, f32x4& dst) {
#ifndef SAGEN_ABLATE_B
        dst = bload16(w_rsrc, (b_voff[c] + i_kbyte) | (i_past << 31));
#endif
    };
    auto store_b = [&](int c, const f32x4& src, float* st) {
        if (NBC * 256 == 6 * BN || b_active[c]) *reinterpret_cast<f32x4*>(st + b_wofs[c]) = src;
    };
    // conversion of one activation chunk in three jobs (one bf16 plane each): [batch-norm + ReLU], split level,
    // ds_write_b64.  cv[c] carries the running fp32 residuals between the jobs.
    float cv[KS][NCH][4];
    auto convert_job = [&](int ks, int c, int level, const f32x4& src, unsigned bad, int c0, float* st) {
        if (level == 0) {
            f32x4 v = src;
            if (PRO) {
                const int cc = c0 + 4 * kc4;
                const float4 sc = *reinterpret_cast<const float4*>(&s_bn[0][cc]);
                const float4 sh = *reinterpret_cast<const float4*>(&s_bn[1][cc]);
                const bool ok = bad == 0;
                v[0] = ok ? fmaxf(fmaf(v[0], sc.x, sh.x), 0.f) : 0.f;
                v[1] = ok ? fmaxf(fmaf(v[1], sc.y, sh.y), 0.f) : 0.f;
                v[2] = ok ? fmaxf(fmaf(v[2], sc.z, sh.z), 0.f) : 0.f;
                v[3] = ok ? fmaxf(fmaf(v[3], sc.w, sh.w), 0.f) : 0.f;
            }
            cv[ks][c][0] = v[0]; cv[ks][c][1] = v[1]; cv[ks][c][2] = v[2]; cv[ks][c][3] = v[3];
        }
        u32x2 pk;
        pk[0] = split_pair(cv[ks][c][0], cv[ks][c][1]);
        pk[1] = split_pair(cv[ks][c][2], cv[ks][c][3]);
        if (a_active) *reinterpret_cast<u32x2*>(st + level * A_PL + a_wofs[c]) = pk;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, kk = lane >> 5;
    const int foff = 4 * (kk ^ ((li >> 3) & 1));          // this lane's 16-B half of a 32-B plane row

    // one side job of K tile `ks` of a step: stores / converts move register set PS into LDS stage `st`, loads fill
    // register set PL (begin_load for that tile must have run)
    auto side_job = [&](int job, int ks, auto pl_tag, float* st) {
        constexpr int PL = decltype(pl_tag)::value, PS = PL ^ 1;
        float* sub = st + ks * SUB_F;
        if (job < NBC) store_b(job, braw[PS][ks][job], sub);
        else if (job < 2 * NBC) load_b(job - NBC, braw[PL][ks][job - NBC]);
        else if (job < 2 * NBC + NCH) load_a(job - 2 * NBC, araw[PL][ks][job - 2 * NBC], abad[PL][ks][job - 2 * NBC]);
        else {
            const int cj = job - 2 * NBC - NCH, c = cj / 3;
            convert_job(ks, c, cj - 3 * c, araw[PS][ks][c], abad[PS][ks][c], ac0[PS][ks], sub);
        }
    };

#ifdef SAGEN_TRACE
    unsigned long long* trc = (d.trace && tile_m == d.trace_block && blockIdx.y == 0) ? (unsigned long long*)d.trace + (size_t)wave * 64 * 8 : nullptr;
#define TRC3(ph) do { if (trc && lane == 0 && t < 64) trc[t * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
#else
#define TRC3(ph) do { } while (0)
#endif
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // the six products: hh, hm, mh, hl, lh, mm
    const int nsteps = (ntiles + KS - 1) / KS;
    // ---- pipeline fill: step 0 into LDS stage 0, step 1 in flight into register set 1 ----
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        begin_load(kc0 + ks, ac0[0][ks]);
#pragma unroll
        for (int c = 0; c < NBC; ++c) load_b(c, braw[0][ks][c]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_a(c, araw[0][ks][c], abad[0][ks][c]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        begin_load(kc0 + KS + ks, ac0[1][ks]);
#pragma unroll
        for (int c = 0; c < NBC; ++c) load_b(c, braw[1][ks][c]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_a(c, araw[1][ks][c], abad[1][ks][c]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int c = 0; c < NBC; ++c) store_b(c, braw[0][ks][c], smem + ks * SUB_F);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) convert_job(ks, c, lv, araw[0][ks][c], abad[0][ks][c], ac0[0][ks], smem + ks * SUB_F);
    }
    lds_barrier();

    // one step; P = step & 1 (compile time): consume LDS stage P, fill stage P^1 from register set P^1, load into set P
    auto step = [&](auto parity, int t) {
        constexpr int P = decltype(parity)::value;
        const float* cur = smem + P * STAGE_F;
        float* nxt = smem + (P ^ 1) * STAGE_F;
            TRC3(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            begin_load(kc0 + (t + 2) * KS + ks, ac0[P][ks]);
            bf16x8 aq[3][MT], bq[3][NT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    aq[pl][i] = *reinterpret_cast<const bf16x8*>(cur + ks * SUB_F + pl * A_PL + (wm * WM + i * 32 + li) * 8 + foff);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bq[pl][j] = *reinterpret_cast<const bf16x8*>(cur + ks * SUB_F + 3 * A_PL + pl * B_PL + (wn * WN + j * 32 + li) * 8 + foff);
            }
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#ifndef SAGEN_ABLATE_MFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[TA[tt]][i], bq[TB[tt]][j], acc[i][j], 0, 0, 0);
#else
                        asm volatile("" ::"v"(aq[TA[tt]][i]), "v"(bq[TB[tt]][j]));
#endif
                        // side jobs of this K tile, spread over its MFMAs
                        constexpr int NM1 = 6 * MT * NT;
                        const int idx = (tt * MT + i) * NT + j;
#pragma unroll
                        for (int g = 0; g < NJOB; ++g)
                            if (idx == (g * NM1 / NJOB < NM1 ? g * NM1 / NJOB : NM1 - 1)) side_job(g, ks, std::integral_constant<int, P>{}, nxt);
#ifdef SAGEN_TRACE
                        if (ks == 0 && idx == 0) TRC3(1);
                        if (ks == KS - 1 && idx == NM1 - 1) TRC3(2);
#endif
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TRC3(3);
        lds_barrier();
        TRC3(4);
    };
    for (int t = 0; t < nsteps; t += 2) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nsteps) step(std::integral_constant<int, 1>{}, t + 1);
    }

    igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, smem, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN, int KS, bool PRO>
__global__ __launch_bounds__(256, 2) void igemm3_kernel(const IgemmDesc d) {
    igemm3_body<BM, BN, WM, WN, KS, PRO>(d);
}

template <int BM, int BN, int WM, int WN, int KS>
static int launch_cfg3(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk);
    if (d.in_scale != nullptr || d.bn_in.acc != nullptr)
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, KS, true>), grid, dim3(256), 0, s, d);
    else
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, KS, false>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// ------------------------------------------------------------------------------------------------------------
// igemm3dw_kernel: 3x3 stride-1 SAME convolutions - the three horizontal taps of a filter row share ONE activation tile
// ------------------------------------------------------------------------------------------------------------
// In the generic kernel every activation element is loaded, batch-normalised and split once per TAP (nine times per
// workgroup), and that operand-split VALU - not the matrix pipe - bounds it.  For a dense stride-1 3x3
// conv the input pixel of output pixel q under tap (dh, dw) is simply q + dh*W + dw in the flattened [B*H*W] pixel
// index, so the tile for (dh, channel chunk) is staged ONCE with one halo pixel on either side (BM + 2 rows), and the
// three dw taps read their fragments from it at row offsets 0 / 1 / 2.  What the flattened shift gets wrong - the
// pixel left of column 0 and right of column W-1 is padding, not the neighbouring image row - is repaired on the
// fragments: lanes whose output pixel sits on that image edge zero their A fragments for that tap.
// K order: (dh, channel chunk, dw); loads / split work per MFMA drop 3x, the filter side is unchanged.
// MERGE: the three taps of a group also share ONE barrier step (their three filter tiles are staged together):
// 3x the MFMAs per barrier - for narrow N, where a single tap is only a few MFMAs per wave.
template <int BM, int BN, int WM, int WN, bool PRO, bool MERGE>
__device__ __forceinline__ void igemm3dw_body(const IgemmDesc& d) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4 && BM >= 64, "4 waves per workgroup, BM >= 64");
    constexpr int NCH = BM / 64;                        // 16-B activation chunks per thread per tile (+1 halo chunk on wave 0)
    constexpr int NBC = (6 * BN + 255) / 256;
    constexpr int AR = BM + 8;                          // rows allocated per A plane (BM + 2 used)
    constexpr int A_PL = AR * 8, B_PL = BN * 8;         // floats per plane
    constexpr int NTAP = MERGE ? 3 : 1;                 // filter tiles per stage
    constexpr int A_ST = 3 * A_PL, B_ST = NTAP * 3 * B_PL;   // floats per stage
    constexpr int NM1 = 6 * MT * NT;                    // MFMAs per tap

    __shared__ __attribute__((aligned(16))) float smem[2 * A_ST + 2 * B_ST];
    __shared__ RowInfo s_row[BM];
    __shared__ int s_tapb[1];
    __shared__ __attribute__((aligned(16))) float s_bn[2][MAX_BN_C];
    float* const a_stage = smem;
    float* const b_stage = smem + 2 * A_ST;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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

    igemm_setup<BM>(d, m0, tid, true, s_row, s_tapb, s_bn);

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.x, 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // ---- activation loader: tile row rho <-> flattened pixel q = m0 - 1 + rho; chunk = 4 channels ----
    const int kc4 = tid & 3;
    const int W = d.Win, H = d.Hin;
    unsigned a_voff[NCH + 1], a_hbad[NCH + 1];          // byte offset of pixel q (+ chunk), validity bits per dh (-1, 0, +1)
    int a_wofs[NCH + 1];
    const bool halo_lane = tid < 8;                     // wave 0 also loads tile rows BM, BM+1
#pragma unroll
    for (int c = 0; c <= NCH; ++c) {
        const int rho = c < NCH ? ((tid + 256 * c) >> 2) : BM + (tid >> 2);
        const int q = m0 - 1 + rho;
        unsigned bad = 7u;
        if (q >= 0 && q < d.M && (c < NCH || halo_lane)) {
            const int h = (q / W) % H;
            bad = (h == 0 ? 1u : 0u) | (h == H - 1 ? 4u : 0u);
        }
        a_hbad[c] = bad;
        a_voff[c] = (unsigned)((long)q * d.ldx * 4) + 16u * kc4;
        a_wofs[c] = rho * 8 + 4 * ((kc4 >> 1) ^ ((rho >> 3) & 1)) + 2 * (kc4 & 1);
    }
    // ---- filter loader (as igemm3_body) ----
    unsigned b_voff[NBC];
    int b_wofs[NBC];
    bool b_active[NBC];
#pragma unroll
    for (int c = 0; c < NBC; ++c) {
        const int q = tid + 256 * c;
        const int pl = q / (2 * BN), rem = q - pl * (2 * BN);
        const int r = rem >> 1, half = rem & 1;
        const int n = n0 + r;
        b_active[c] = q < 6 * BN;
        b_voff[c] = (b_active[c] && n < d.N) ? (unsigned)(((long)pl * d.N + n) * 32 + 16 * half) : OOB;
        b_wofs[c] = pl * B_PL + r * 8 + 4 * (half ^ ((r >> 3) & 1));
    }

    // ---- K range of this split, in groups (dh, chunk) of three steps ----
    const int nchunk = d.Cin >> 4;
    const int G = 3 * nchunk;
    const int gper = (G + d.splitk - 1) / d.splitk;
    const int g0 = z * gper;
    const int g1 = min(G, g0 + gper);
    const int ngroups = max(g1 - g0, 0);
    const int nsteps = 3 * ngroups;

    // trackers: next group to LOAD activations for; next step to LOAD filters for
    int la_dh = g0 / nchunk, la_ch = g0 - la_dh * nchunk, la_g = g0;      // dh index 0..2 (= dh + 1)
    int lb_dh = la_dh, lb_ch = la_ch, lb_dw = 0, lb_s = 0;
    unsigned i_tb = 0, i_apast = 0, i_kbyte = 0, i_bpast = 0;
    int i_hsel = 0;
    // state of the group whose raw data sits in araw (set by begin_a)
    int cv_c0 = 0;
    auto begin_a = [&]() {
        i_apast = la_g >= g1 ? 1u : 0u;
        i_hsel = la_dh;
        i_tb = (unsigned)((((la_dh - 1) * W) * d.ldx + la_ch * 16) * 4);
        cv_c0 = la_ch * 16;
        ++la_g; ++la_ch;
        if (la_ch == nchunk) { la_ch = 0; ++la_dh; }
    };
    auto begin_b = [&]() {
        i_bpast = lb_s >= nsteps ? 1u : 0u;
        i_kbyte = (unsigned)((lb_dh * 3 + lb_dw) * nchunk + lb_ch) * (unsigned)(d.N * 96);
        ++lb_s; ++lb_dw;
        if (lb_dw == 3) { lb_dw = 0; ++lb_ch; if (lb_ch == nchunk) { lb_ch = 0; ++lb_dh; } }
    };

    f32x4 araw[NCH + 1], braw[2][NTAP][NBC];
#pragma unroll
    for (int c = 0; c <= NCH; ++c) araw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int c = 0; c < NBC; ++c) braw[p][t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned abad[NCH + 1];
#pragma unroll
    for (int c = 0; c <= NCH; ++c) abad[c] = 1;

    auto load_a = [&](int c) {
        const unsigned bad = ((a_hbad[c] >> i_hsel) & 1u) | i_apast;
        abad[c] = bad;
#ifndef SAGEN_ABLATE_A
        araw[c] = bload16(x_rsrc, (a_voff[c] + i_tb) | (bad << 31));
#endif
    };
    auto load_b = [&](int c, f32x4& dst) {
#ifndef SAGEN_ABLATE_B
        dst = bload16(w_rsrc, (b_voff[c] + i_kbyte) | (i_bpast << 31));
#endif
    };
    auto store_b = [&](int c, const f32x4& src, float* st) {
        if (NBC * 256 == 6 * BN || b_active[c]) *reinterpret_cast<f32x4*>(st + b_wofs[c]) = src;
    };
    float cv[NCH + 1][4];
    auto convert_job = [&](int c, int level, float* st) {
        if (level == 0) {
            f32x4 v = araw[c];
            if (PRO) {
                const int cc = cv_c0 + 4 * kc4;
                const float4 sc = *reinterpret_cast<const float4*>(&s_bn[0][cc]);
                const float4 sh = *reinterpret_cast<const float4*>(&s_bn[1][cc]);
                const bool ok = abad[c] == 0;
                v[0] = ok ? fmaxf(fmaf(v[0], sc.x, sh.x), 0.f) : 0.f;
                v[1] = ok ? fmaxf(fmaf(v[1], sc.y, sh.y), 0.f) : 0.f;
                v[2] = ok ? fmaxf(fmaf(v[2], sc.z, sh.z), 0.f) : 0.f;
                v[3] = ok ? fmaxf(fmaf(v[3], sc.w, sh.w), 0.f) : 0.f;
            }
            cv[c][0] = v[0]; cv[c][1] = v[1]; cv[c][2] = v[2]; cv[c][3] = v[3];
        }
        u32x2 pk;
        pk[0] = split_pair(cv[c][0], cv[c][1]);
        pk[1] = split_pair(cv[c][2], cv[c][3]);
        if (c < NCH || halo_lane) *reinterpret_cast<u32x2*>(st + level * A_PL + a_wofs[c]) = pk;
    };
    // all conversion work of chunk c (wave 0 only for the halo chunk)
    auto convert_chunk = [&](int c, float* st) {
        if (c < NCH || wave == 0) {
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) convert_job(c, lv, st);
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
    // fragment addressing: output row r = wm*WM + i*32 + li reads tile row r + dwi (dwi = 0, 1, 2)
    int a_foff[3][MT];
    bool at_left[MT], at_right[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * WM + i * 32 + li;
        const unsigned nm = s_row[r].nmlo;
        at_left[i] = ((nm >> 3) & 1u) != 0;             // tap (dh 0, dw -1) is padding <=> w == 0 (or the row is past M)
        at_right[i] = ((nm >> 5) & 1u) != 0;            // tap (dh 0, dw +1) is padding <=> w == W - 1
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int rho = r + dwi;
            a_foff[dwi][i] = rho * 8 + 4 * (kk ^ ((rho >> 3) & 1));
        }
    }
    const int b_foff = (wn * WN + li) * 8 + 4 * (kk ^ ((li >> 3) & 1));

    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if constexpr (MERGE) {
        // ---- one barrier step per group: A stage GP + the three filter tiles of the group in B stage GP ----
        // fill: group g0 in stage 0 (activations and filters), filters of group g0+1 in register set 1
        begin_a();
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_a(c);
        if (wave == 0) load_a(NCH);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                begin_b();
#pragma unroll
                for (int c = 0; c < NBC; ++c) load_b(c, braw[p][t][c]);
            }
#pragma unroll
        for (int c = 0; c <= NCH; ++c) convert_chunk(c, a_stage);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < NBC; ++c) store_b(c, braw[0][t][c], b_stage + t * 3 * B_PL);
        lds_barrier();

        auto gstep = [&](auto gp_tag) {
            constexpr int GP = decltype(gp_tag)::value;
            const float* acur = a_stage + GP * A_ST;
            float* anxt = a_stage + (GP ^ 1) * A_ST;
            const float* bcur = b_stage + GP * B_ST;
            float* bnxt = b_stage + (GP ^ 1) * B_ST;
            begin_a();                                    // activations of the next group: loaded early, converted late
            // jobs, in issue order: activation loads | filter stores (group g+1) | filter loads (group g+2) | conversions
            constexpr int NJ = (NCH + 1) + 3 * NBC + 3 * NBC + (NCH + 1);
            constexpr int NMT = 3 * NM1;
#pragma unroll
            for (int dwi = 0; dwi < 3; ++dwi) {
                bf16x8 aq[3][MT], bq[3][NT];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) aq[pl][i] = *reinterpret_cast<const bf16x8*>(acur + pl * A_PL + a_foff[dwi][i]);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bq[pl][j] = *reinterpret_cast<const bf16x8*>(bcur + (dwi * 3 + pl) * B_PL + j * 32 * 8 + b_foff);
                }
                if (dwi != 1) {                           // image-edge repair
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const bool kill = dwi == 0 ? at_left[i] : at_right[i];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            u32x4 u = __builtin_bit_cast(u32x4, aq[pl][i]);
                            u[0] = kill ? 0u : u[0]; u[1] = kill ? 0u : u[1]; u[2] = kill ? 0u : u[2]; u[3] = kill ? 0u : u[3];
                            aq[pl][i] = __builtin_bit_cast(bf16x8, u);
                        }
                    }
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
#ifndef SAGEN_ABLATE_MFMA
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[TA[tt]][i], bq[TB[tt]][j], acc[i][j], 0, 0, 0);
#else
                            asm volatile("" ::"v"(aq[TA[tt]][i]), "v"(bq[TB[tt]][j]));
#endif
                            const int idx = dwi * NM1 + (tt * MT + i) * NT + j;
#pragma unroll
                            for (int g = 0; g < NJ; ++g)
                                if (idx == (g * NMT / NJ < NMT ? g * NMT / NJ : NMT - 1)) {
                                    if (g < NCH + 1) { if (g < NCH || wave == 0) load_a(g); }
                                    else if (g < NCH + 1 + 3 * NBC) {
                                        const int k = g - (NCH + 1), t = k / NBC;
                                        store_b(k - t * NBC, braw[GP ^ 1][t][k - t * NBC], bnxt + t * 3 * B_PL);
                                    } else if (g < NCH + 1 + 6 * NBC) {
                                        const int k = g - (NCH + 1) - 3 * NBC, t = k / NBC;
                                        if (k - t * NBC == 0) begin_b();
                                        load_b(k - t * NBC, braw[GP][t][k - t * NBC]);
                                    } else {
                                        convert_chunk(g - (NCH + 1) - 6 * NBC, anxt);
                                    }
                                }
                        }
            }
            lds_barrier();
        };
        for (int g = 0; g < ngroups; g += 2) {
            gstep(I0{});
            if (g + 1 < ngroups) gstep(I1{});
        }
    } else {
        // ---- pipeline fill: group g0 in A stage 0, filter tile of step 0 in B stage 0, tile of step 1 in registers ----
        begin_a();
    #pragma unroll
        for (int c = 0; c < NCH; ++c) load_a(c);
        if (wave == 0) load_a(NCH);
        begin_b();
    #pragma unroll
        for (int c = 0; c < NBC; ++c) load_b(c, braw[0][0][c]);
        begin_b();
    #pragma unroll
        for (int c = 0; c < NBC; ++c) load_b(c, braw[1][0][c]);
    #pragma unroll
        for (int c = 0; c <= NCH; ++c) convert_chunk(c, a_stage);
    #pragma unroll
        for (int c = 0; c < NBC; ++c) store_b(c, braw[0][0][c], b_stage);
        lds_barrier();

        // one step: GP = parity of the group (A stage), DWI = horizontal tap 0..2, SP = parity of the step (B stage)
        auto step = [&](auto gp_tag, auto dwi_tag, auto sp_tag) {
            constexpr int GP = decltype(gp_tag)::value, DWI = decltype(dwi_tag)::value, SP = decltype(sp_tag)::value;
            const float* acur = a_stage + GP * A_ST;
            float* anxt = a_stage + (GP ^ 1) * A_ST;
            const float* bcur = b_stage + SP * B_ST;
            float* bnxt = b_stage + (SP ^ 1) * B_ST;
            if (DWI == 0) begin_a();                         // addresses of the next group (loaded during this step)
            begin_b();                                       // filter tile two steps ahead

            bf16x8 aq[3][MT], bq[3][NT];
    #pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
    #pragma unroll
                for (int i = 0; i < MT; ++i) {
                    aq[pl][i] = *reinterpret_cast<const bf16x8*>(acur + pl * A_PL + a_foff[DWI][i]);
                }
    #pragma unroll
                for (int j = 0; j < NT; ++j) bq[pl][j] = *reinterpret_cast<const bf16x8*>(bcur + pl * B_PL + j * 32 * 8 + b_foff);
            }
            if (DWI != 1) {                                  // image-edge repair: that neighbour is padding
    #pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool kill = DWI == 0 ? at_left[i] : at_right[i];
    #pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        u32x4 u = __builtin_bit_cast(u32x4, aq[pl][i]);
                        u[0] = kill ? 0u : u[0]; u[1] = kill ? 0u : u[1]; u[2] = kill ? 0u : u[2]; u[3] = kill ? 0u : u[3];
                        aq[pl][i] = __builtin_bit_cast(bf16x8, u);
                    }
                }
            }
            // side jobs: filter store (step s+1) and load (step s+2) on every step; activations of the NEXT group:
            // loads on tap 0, conversion split over taps 1 and 2
            constexpr int NA = DWI == 0 ? NCH + 1 : (DWI == 1 ? (NCH + 1) / 2 : (NCH + 1) - (NCH + 1) / 2);
            constexpr int NJ = 2 * NBC + NA;
    #pragma unroll
            for (int tt = 0; tt < 6; ++tt)
    #pragma unroll
                for (int i = 0; i < MT; ++i)
    #pragma unroll
                    for (int j = 0; j < NT; ++j) {
    #ifndef SAGEN_ABLATE_MFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[TA[tt]][i], bq[TB[tt]][j], acc[i][j], 0, 0, 0);
    #else
                        asm volatile("" ::"v"(aq[TA[tt]][i]), "v"(bq[TB[tt]][j]));
    #endif
                        const int idx = (tt * MT + i) * NT + j;
    #pragma unroll
                        for (int g = 0; g < NJ; ++g)
                            if (idx == (g * NM1 / NJ < NM1 ? g * NM1 / NJ : NM1 - 1)) {
                                if (g < NBC) store_b(g, braw[SP ^ 1][0][g], bnxt);
                                else if (g < 2 * NBC) load_b(g - NBC, braw[SP][0][g - NBC]);
                                else {
                                    const int k = g - 2 * NBC;
                                    if (DWI == 0) { if (k < NCH || wave == 0) load_a(k); }
                                    else if (DWI == 1) convert_chunk(k, anxt);
                                    else convert_chunk((NCH + 1) / 2 + k, anxt);
                                }
                            }
                    }
            lds_barrier();
        };
        for (int g = 0; g < ngroups; g += 2) {
            step(I0{}, I0{}, I0{}); step(I0{}, I1{}, I1{}); step(I0{}, I2{}, I0{});
            if (g + 1 < ngroups) { step(I1{}, I0{}, I1{}); step(I1{}, I1{}, I0{}); step(I1{}, I2{}, I1{}); }
        }

    }

    igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, smem, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN, bool MERGE, bool PRO>
__global__ __launch_bounds__(256, 2) void igemm3dw_kernel(const IgemmDesc d) {
    igemm3dw_body<BM, BN, WM, WN, PRO, MERGE>(d);
}

template <int BM, int BN, int WM, int WN, bool MERGE>
static int launch_cfg3dw(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk);
    if (d.in_scale != nullptr || d.bn_in.acc != nullptr)
        hipLaunchKernelGGL((igemm3dw_kernel<BM, BN, WM, WN, MERGE, true>), grid, dim3(256), 0, s, d);
    else
        hipLaunchKernelGGL((igemm3dw_kernel<BM, BN, WM, WN, MERGE, false>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// called by igemm_launch (igemm.hip) after the shared validation
int igemm3_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s) {
    switch (tile) {
        case TILE_B3_128x128: return launch_cfg3<128, 128, 64, 64, 1>(d, s);
        case TILE_B3_128x64: return launch_cfg3<128, 64, 64, 32, 1>(d, s);
        case TILE_B3_256x64: return launch_cfg3<256, 64, 64, 64, 1>(d, s);
        case TILE_B3_64x64: return launch_cfg3<64, 64, 32, 32, 1>(d, s);
        case TILE_B3_64x128: return launch_cfg3<64, 128, 32, 64, 1>(d, s);
        case TILE_B3_64x256: return launch_cfg3<64, 256, 64, 64, 1>(d, s);
        case TILE_B3_32x128: return launch_cfg3<32, 128, 32, 32, 1>(d, s);
        case TILE_B3_128x32: return launch_cfg3<128, 32, 32, 32, 1>(d, s);
        case TILE_B3_128x64_K2: return launch_cfg3<128, 64, 64, 32, 2>(d, s);
        case TILE_B3_64x64_K2: return launch_cfg3<64, 64, 32, 32, 2>(d, s);
        case TILE_B3_64x128_K2: return launch_cfg3<64, 128, 32, 64, 2>(d, s);
        case TILE_B3_32x128_K2: return launch_cfg3<32, 128, 32, 32, 2>(d, s);
        case TILE_B3_128x32_K2: return launch_cfg3<128, 32, 32, 32, 2>(d, s);
        case TILE_B3DW_128x128: return launch_cfg3dw<128, 128, 64, 64, false>(d, s);
        case TILE_B3DW_128x64: return launch_cfg3dw<128, 64, 64, 32, false>(d, s);
        case TILE_B3DW_256x64: return launch_cfg3dw<256, 64, 64, 64, false>(d, s);
        case TILE_B3DW_64x128: return launch_cfg3dw<64, 128, 32, 64, false>(d, s);
        case TILE_B3DW_64x64: return launch_cfg3dw<64, 64, 32, 32, false>(d, s);
        case TILE_B3DW_64x256: return launch_cfg3dw<64, 256, 64, 64, false>(d, s);
        case TILE_B3DWM_128x64: return launch_cfg3dw<128, 64, 64, 32, true>(d, s);
        case TILE_B3DWM_256x64: return launch_cfg3dw<256, 64, 64, 64, true>(d, s);
        case TILE_B3DWM_64x64: return launch_cfg3dw<64, 64, 32, 32, true>(d, s);
        case TILE_B3DWM_64x128: return launch_cfg3dw<64, 128, 32, 64, true>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "igemm3: bad tile id %d", (int)tile);
    }
}

// fp32 packed filter [N][Kpad] -> three bf16 planes (hi, mid, lo of the bf16x3 split), tiled [Kpad/16][3][N][16] so that
// the 32 rows x 32 B one LDS-DMA instruction fetches are contiguous in memory.
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ wp, long total, int N, int Kpad,
                                                         __bf16* __restrict__ w3) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long n = idx / Kpad;
    const int k = (int)(idx - n * Kpad);
    const float v = wp[idx];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const long o = ((long)(k >> 4) * 3 * N + n) * 16 + (k & 15);        // plane 0 of K tile k/16
    w3[o] = h; w3[o + (long)N * 16] = m; w3[o + (long)N * 32] = (__bf16)r2;
}

int pack_split_launch(float* wp, int N, int Kpad, hipStream_t s) {
    const long total = (long)N * Kpad;
    hipLaunchKernelGGL(pack_split_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, wp, total, N, Kpad,
                       reinterpret_cast<__bf16*>(wp + total));
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
