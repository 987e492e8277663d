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
#include "igemm3_common.h"

namespace sagen {

// KS = K tiles of 16 per barrier step (small tiles amortise the barrier and the loop overhead over 2 tiles);
// PRO = the producer's batch-norm + ReLU is applied to the activations on the way in.
template <int BM, int BN, int WM, int WN, int KS, bool PRO>
__device__ __forceinline__ void igemm3_body(const IgemmDesc& d, const int z) {
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
#ifndef SAGEN_ABLATE_DSW
        if (NBC * 256 == 6 * BN || b_active[c]) *reinterpret_cast<f32x4*>(st + b_wofs[c]) = src;
#else
        asm volatile("" ::"v"(src));
#endif
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
#ifndef SAGEN_ABLATE_DSW
        if (a_active) *reinterpret_cast<u32x2*>(st + level * A_PL + a_wofs[c]) = pk;
#else
        asm volatile("" ::"v"(pk));
#endif
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

    if (d.mm_out != nullptr) {           // fused decoder tail (igemm_tile_ok admits only the tiles compiled here)
        if constexpr (BM <= 128 && BN <= 128 && WM * BN + 7 * 32 <= 2 * STAGE_F && !PRO)
            igemm_epilogue_maskmix<BM, BN, WM, WN, 2 * STAGE_F>(d, acc, smem, m0, n0, tid);
        return;
    }
    if (!igemm_epilogue_rows<BM, BN, WM, WN, 2 * STAGE_F>(d, acc, s_row, smem, n0, tid))
        igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, smem, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN, int KS, bool PRO>
__global__ __launch_bounds__(256, 2) void igemm3_kernel(const IgemmDesc d_in) {
    IgemmDesc d = d_in;
    int z = blockIdx.z;
    if (d.grp.G > 1) {                      // grouped launch: blockIdx.z = group * splitk + z (common.h)
        const int g = d.splitk == 1 ? z : z / d.splitk;
        z -= g * d.splitk;
        igemm_relocate(d, g);
    }
    igemm3_body<BM, BN, WM, WN, KS, PRO>(d, z);
}

template <int BM, int BN, int WM, int WN, int KS>
static int launch_cfg3(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk * d.grp.G);
    if (d.in_scale != nullptr || d.bn_in.acc != nullptr)
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, KS, true>), grid, dim3(256), 0, s, d);
    else
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, KS, false>), grid, dim3(256), 0, s, d);
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
        default: return igemm3dw_dispatch(d, tile, s);
    }
}

// fp32 packed filter [N][Kpad] -> three bf16 planes (hi, mid, lo of the bf16x3 split), tiled [Kpad/16][3][N][16] so that
// the three planes of one K tile of 16 form one contiguous block (fully coalesced 16-byte loads).
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
