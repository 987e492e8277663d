// Implicit-GEMM convolution / transposed convolution / FC on the gfx950 fp32 matrix cores.
//
// Replaces, for the inference path, tf.nn.convolution (core.py:206), tf.nn.conv2d_transpose
// (core.py:140) and tf.matmul (core.py:79) of the reference's TF1 runtime.
//
// Design (CDNA4) — shaped by one measured fact: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles,
// 64 FLOP/clk/SIMD = the vector rate) does NOT overlap with VALU work of either wave on its SIMD
// (per-phase s_memtime traces).  Everything that is not an MFMA is therefore paid in
// full, and the kernel is built to have almost nothing but MFMAs in its K loop:
//  * no im2col and no register staging: A and B tiles go global -> LDS by LDS-DMA
//    (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction: 16 rows x 64 B).  No ds_write phase.
//  * buffer descriptors do the bounds work: a lane whose tap falls in the padding, whose row is
//    >= M, whose k is >= K or whose filter row is >= N sets bit 31 of its offset and the hardware
//    range check writes zeros.  Cost per 16-byte load: 3 VALU (bfe, add, lshl_or).
//  * per-row byte offsets and 64-bit tap-validity masks are computed once per workgroup (LDS);
//    the tap / channel position of a K tile is tracked in SGPRs (no division, no table).
//  * LDS rows are 64 B, unpadded (DMA writes are lane-linear); the 16-B chunk index is XOR-swizzled
//    with (row >> 2) & 3 on the SOURCE side of the DMA and on the fragment read, which makes every
//    ds_read_b128 of an MFMA fragment conflict-free.
//  * k is consumed in a lane-permuted order: lane (i, kk) of the MFMA holds the float4
//    A[i][8u+4kk .. +3]; MFMA r of a group contracts k in {8u+r, 8u+4+r}; the filter is packed
//    [n][k] so B fragments are read the same way: one ds_read_b128 per operand per four MFMAs.
//  * double-buffered LDS, one barrier per K tile; the DMA of tile t+1 flies under the MFMAs of t.
//  * the previous layer's training-mode batch-norm (relu(v*scale+shift)) is applied to A fragments
//    after the LDS read (padding stays exactly zero through the per-row masks).
//  * epilogue: bias / ReLU, depth-to-space scatter (transposed convs), per-channel (sum, sumsq)
//    accumulated with fp64 atomics for batch-norm, or raw split-K partials.
#include "igemm_common.h"
#include <cstdlib>

namespace sagen {

// (row tables, DMA helpers and the epilogue live in igemm_common.h)

template <int BM, int BN, int WM, int WN, int STG, int BK>
__global__ __launch_bounds__(256, (WM * WN >= 4096 ? 2 : (WM * WN >= 2048 ? (BK == 32 ? 2 : 3) : (BK == 32 ? 3 : 4)))) void igemm_kernel(const IgemmDesc d_in) {
    static_assert(BK == 16 || BK == 32, "K tile of 16 or 32");
    IgemmDesc d = d_in;
    int z = blockIdx.z;
    if (d.grp.G > 1) {                      // grouped launch: blockIdx.z = group * splitk + z (common.h)
        const int g = d.splitk == 1 ? z : z / d.splitk;
        z -= g * d.splitk;
        igemm_relocate(d, g);
    }
    constexpr int CPR = BK / 4;            // 16-B chunks per LDS row
    constexpr int RPI = 64 / CPR;          // rows covered by one DMA instruction (64 lanes x 16 B)
    constexpr int RPBR = 64 / BK;          // rows per 256-B LDS bank row: the swizzle key is (row / RPBR) % CPR
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int A_DMA = BM / RPI, B_DMA = BN / RPI;        // DMA instructions per tile (RPI rows each)
    constexpr int A_PW = (A_DMA + 3) / 4, B_PW = (B_DMA + 3) / 4;   // per wave
    constexpr int PER = A_PW + B_PW;                          // DMA instructions per wave per K tile
    // ring depth: 3 stages (two tiles in flight, counted vmcnt) when every wave issues the same number of
    // DMA instructions per tile; otherwise 2 stages with a full drain
    constexpr bool EVEN = (A_DMA % 4 == 0) && (B_DMA % 4 == 0);
    constexpr int STAGES = (EVEN && STG == 3) ? 3 : 2;
    constexpr int NMFMA = (BK / 2) * MT * NT;                 // MFMAs per wave per K tile
    constexpr int TILE_F = (BM + BN) * BK;                    // floats per stage

    __shared__ __attribute__((aligned(16))) float smem[STAGES * TILE_F];
    __shared__ RowInfo s_row[BM];
    __shared__ int s_tapb[MAX_TAPS];          // byte displacement per tap (only for the per-lane tap path)
    __shared__ __attribute__((aligned(16))) float s_bn[2][MAX_BN_C];   // input batch-norm scale / shift per channel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const bool uni = d.uniform_taps != 0;

    igemm_setup<BM>(d, m0, tid, uni, s_row, s_tapb, s_bn);

    // ---- LDS-DMA loader state ----
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.x, 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, d.w_bytes, 0x00020000);
    // LDS position of lane l of DMA instruction `inst`: row inst*RPI + l/CPR, physical chunk l%CPR; it must hold
    // logical chunk (l%CPR) ^ ((row / RPBR) % CPR).  inst*RPI/RPBR is 0 mod CPR for BK=16 and 4*(inst&1) for BK=32,
    // and inst = wave + 4j has the parity of the wave.
    const int src_chunk = (lane % CPR) ^ (((BK == 32 ? 4 * (wave & 1) : 0) + (lane / CPR) / RPBR) % CPR);
    unsigned a_voff[A_PW], a_nmlo[A_PW], a_nmhi[A_PW], b_voff[B_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int inst = wave + 4 * j;                       // wave-uniform
        a_voff[j] = OOB; a_nmlo[j] = 0xffffffffu; a_nmhi[j] = 0xffffffffu;
        if (inst < A_DMA) {
            const RowInfo ri = s_row[inst * RPI + lane / CPR];
            a_voff[j] = ri.boff + 16u * src_chunk;
            a_nmlo[j] = ri.nmlo; a_nmhi[j] = ri.nmhi;
        }
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        const int n = n0 + inst * RPI + lane / CPR;
        b_voff[j] = (inst < B_DMA && n < d.N) ? (unsigned)((long)n * d.Kpad * 4) + 16u * src_chunk : OOB;
    }

    const int nk = d.Kpad / BK;
    const int nk_per = (nk + d.splitk - 1) / d.splitk;
    const int kc0 = z * nk_per;
    const int kc1 = min(nk, kc0 + nk_per);
    const bool prologue = d.in_scale != nullptr || d.bn_in.acc != nullptr;

    // SGPR trackers of the (tap, channel) position: q_* = next tile to ISSUE, p_* = tile being CONTRACTED
    int q_tap = 0, q_th = 0, q_tw = 0, q_c0 = 0, p_tap = 0, p_c0 = 0;
    if (uni) {
        const int k0 = kc0 * BK;
        if (d.ntaps > 1) { q_tap = k0 >> d.log2Cin; q_c0 = k0 & (d.Cin - 1); q_th = q_tap / d.TW; q_tw = q_tap - q_th * d.TW; }
        else q_c0 = k0;
        p_tap = q_tap; p_c0 = q_c0;
    }

    // issue state of the tile being issued (computed by begin_issue, consumed by issue_one)
    unsigned i_tb = 0, i_bit = 0, i_kbyte = 0, i_kok = 1;
    int i_stage = 0;
    bool i_hi = false;
    auto begin_issue = [&](int kc, int stage) {
        i_stage = stage;
        i_kbyte = (unsigned)(kc * (BK * 4));
        if (uni) {
            i_tb = (unsigned)((((q_th * d.tap_sh + d.tap_h0) * d.Win + (q_tw * d.tap_sw + d.tap_w0)) * d.ldx + q_c0) * 4);
            i_bit = (unsigned)(q_tap & 31);
            i_hi = q_tap >= 32;
            q_c0 += BK;
            if (d.ntaps > 1 && q_c0 == d.Cin) {
                q_c0 = 0; ++q_tap; ++q_tw;
                if (q_tw == d.TW) { q_tw = 0; ++q_th; }
            }
        } else {       // per-lane tap (Cin < 16 or a ragged K tail): table lookup
            const int k = kc * BK + 4 * src_chunk;
            const bool kok = k < d.K;
            const int tap = (kok && d.ntaps > 1) ? (k >> d.log2Cin) : 0;
            const int c = d.ntaps > 1 ? (k & (d.Cin - 1)) : k;
            i_tb = (unsigned)(s_tapb[tap] + 4 * c) - 16u * src_chunk;
            i_bit = (unsigned)(tap & 31);
            i_hi = tap >= 32;
            i_kok = kok ? 1u : 0u;
        }
    };
    // g-th DMA instruction of this wave for the tile prepared by begin_issue (g is a compile-time constant)
    auto issue_one = [&](int g) {
        float* st = smem + i_stage * TILE_F;
        if (g < A_PW) {
            const int inst = wave + 4 * g;
            if (EVEN || inst < A_DMA) {
                const unsigned word = i_hi ? a_nmhi[g] : a_nmlo[g];
                unsigned bad = (word >> i_bit) & 1u;
                if (!uni) bad |= (i_kok ^ 1u);
#ifndef SAGEN_ABLATE_A
                dma16(x_rsrc, st + inst * RPI * BK, (a_voff[g] + i_tb) | (bad << 31), 0);
#endif
            }
        } else {
            const int inst = wave + 4 * (g - A_PW);
#ifndef SAGEN_ABLATE_B
            if (EVEN || inst < B_DMA) dma16(w_rsrc, st + (BM + inst * RPI) * BK, b_voff[g - A_PW], i_kbyte);
#endif
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
    // fragment addressing: row (wm*WM + i*32 + li), physical chunk (2u + kk) ^ ((row / RPBR) % CPR)
    const int fsw = (li / RPBR) % CPR;
    // BN prologue state: inverted masks of this lane's fragment rows
    unsigned f_nmlo[MT], f_nmhi[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        f_nmlo[i] = 0; f_nmhi[i] = 0;
        if (prologue) { const RowInfo ri = s_row[wm * WM + i * 32 + li]; f_nmlo[i] = ri.nmlo; f_nmhi[i] = ri.nmhi; }
    }

#ifdef SAGEN_TRACE
    unsigned long long* trc = (d.trace && tile_m == d.trace_block && blockIdx.y == 0) ? (unsigned long long*)d.trace + (size_t)wave * 64 * 8 : nullptr;
#define TRC(ph) do { if (trc && lane == 0 && (kc - kc0) < 64) trc[(kc - kc0) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
#else
#define TRC(ph) do { } while (0)
#endif

    // ---- pipeline fill: STAGES-1 tiles in flight ----
    const int ntiles = kc1 - kc0;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < ntiles) {
            begin_issue(kc0 + t, t);
#pragma unroll
            for (int g = 0; g < PER; ++g) issue_one(g);
        }
    if (STAGES == 3 && ntiles > 1) wait_vmcnt<EVEN ? PER : 0>(); else wait_vmcnt<0>();
    lds_barrier();

    int stage = 0;                                   // ring slot of the tile being contracted
    for (int kc = kc0; kc < kc1; ++kc) {
        TRC(0);
        const bool more = kc + (STAGES - 1) < kc1;   // is there a tile to issue during this iteration?
        int istage = stage + (STAGES - 1);
        if (istage >= STAGES) istage -= STAGES;
        if (more) begin_issue(kc + (STAGES - 1), istage);
        const float* Ab = smem + stage * TILE_F + (wm * WM + li) * BK;
        const float* Bb = smem + stage * TILE_F + (BM + wn * WN + li) * BK;
        // the K tile is contracted in halves of 16 (two 8-deep MFMA groups each), fragments re-read per half
#pragma unroll
        for (int hf = 0; hf < BK / 16; ++hf) {
            float4 af[2][MT], bf[2][NT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int foff = 4 * ((2 * (2 * hf + u) + kk) ^ fsw);
#pragma unroll
                for (int i = 0; i < MT; ++i) af[u][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * BK + foff);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[u][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * BK + foff);
            }
            if (prologue) {
                // relu(v*scale[c] + shift[c]) of the producer's batch-norm; padding / invalid rows stay 0
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = p_c0 + 8 * (2 * hf + u) + 4 * kk;     // channel of af[u][.].x  (prologue => uniform taps)
                    const float4 sc = *reinterpret_cast<const float4*>(&s_bn[0][c]);
                    const float4 sh = *reinterpret_cast<const float4*>(&s_bn[1][c]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const unsigned word = p_tap < 32 ? f_nmlo[i] : f_nmhi[i];
                        const bool ok = ((word >> (p_tap & 31)) & 1u) == 0;
                        float4 v = af[u][i];
                        v.x = ok ? fmaxf(fmaf(v.x, sc.x, sh.x), 0.f) : 0.f;
                        v.y = ok ? fmaxf(fmaf(v.y, sc.y, sh.y), 0.f) : 0.f;
                        v.z = ok ? fmaxf(fmaf(v.z, sc.z, sh.z), 0.f) : 0.f;
                        v.w = ok ? fmaxf(fmaf(v.w, sc.w, sh.w), 0.f) : 0.f;
                        af[u][i] = v;
                    }
                }
            }
            if (hf == 0) TRC(1);
            // MFMAs with the DMA issue of a later tile spread between them (it hides in the 64-cycle MFMA shadow)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const float a = r == 0 ? af[u][i].x : r == 1 ? af[u][i].y : r == 2 ? af[u][i].z : af[u][i].w;
                            const float b = r == 0 ? bf[u][j].x : r == 1 ? bf[u][j].y : r == 2 ? bf[u][j].z : bf[u][j].w;
#ifndef SAGEN_ABLATE_MFMA
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
#else
                            asm volatile("" ::"v"(a), "v"(b));       // keeps the fragment reads alive, no instructions
#endif
                            const int idx = (((hf * 2 + u) * 4 + r) * MT + i) * NT + j;          // 0 .. NMFMA-1
                            // after MFMA idx, issue DMA g when idx == (g+1)*NMFMA/(PER+1) - 1
#pragma unroll
                            for (int g = 0; g < PER; ++g)
                                if (idx == (g + 1) * NMFMA / (PER + 1) - 1 && more) issue_one(g);
                        }
        }
        TRC(2);
        if (uni) {                                   // advance the consume-side tracker
            p_c0 += BK;
            if (d.ntaps > 1 && p_c0 == d.Cin) { p_c0 = 0; ++p_tap; }
        }
        // the next tile to contract (kc+1) must have landed; with 3 stages the one issued just now may still fly
        if (STAGES == 3 && more) wait_vmcnt<EVEN ? PER : 0>(); else wait_vmcnt<0>();
        TRC(3);
        lds_barrier();
        TRC(4);
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }

    igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, smem, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN, int STG, int BK>
static int launch_cfg(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk * d.grp.G);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, STG, BK>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

struct TileCfg { int bm, bn, bk; const char* name; bool split = false; bool dw3 = false; bool s2 = false; bool p3 = false; bool g = false; bool h = false; };
static const TileCfg kTiles[TILE_AUTO] = {
    {128, 128, 16, "igemm_kernel<128,128,64,64,3,16>"}, {128, 64, 16, "igemm_kernel<128,64,64,32,3,16>"},
    {256, 64, 16, "igemm_kernel<256,64,64,64,3,16>"},   {64, 64, 16, "igemm_kernel<64,64,32,32,3,16>"},
    {128, 32, 16, "igemm_kernel<128,32,32,32,2,16>"},   {32, 128, 16, "igemm_kernel<32,128,32,32,2,16>"},
    {128, 128, 16, "igemm_kernel<128,128,64,64,2,16>"}, {128, 64, 16, "igemm_kernel<128,64,64,32,2,16>"},
    {256, 64, 16, "igemm_kernel<256,64,64,64,2,16>"},   {64, 64, 16, "igemm_kernel<64,64,32,32,2,16>"},
    {64, 128, 16, "igemm_kernel<64,128,32,64,3,16>"},   {64, 128, 16, "igemm_kernel<64,128,32,64,2,16>"},
    {64, 256, 16, "igemm_kernel<64,256,64,64,3,16>"},   {64, 256, 16, "igemm_kernel<64,256,64,64,2,16>"},
    {256, 32, 16, "igemm_kernel<256,32,64,32,2,16>"},
    {64, 64, 32, "igemm_kernel<64,64,32,32,2,32>"},     {64, 128, 32, "igemm_kernel<64,128,32,64,2,32>"},
    {128, 64, 32, "igemm_kernel<128,64,64,32,2,32>"},   {128, 128, 32, "igemm_kernel<128,128,64,64,2,32>"},
    {32, 128, 32, "igemm_kernel<32,128,32,32,2,32>"},   {128, 32, 32, "igemm_kernel<128,32,32,32,2,32>"},
    // fp32-equivalent bf16x3 kernels (igemm3.hip)
    {128, 128, 16, "igemm3_kernel<128,128,64,64,1>", true}, {128, 64, 16, "igemm3_kernel<128,64,64,32,1>", true},
    {256, 64, 16, "igemm3_kernel<256,64,64,64,1>", true},   {64, 64, 16, "igemm3_kernel<64,64,32,32,1>", true},
    {64, 128, 16, "igemm3_kernel<64,128,32,64,1>", true},   {64, 256, 16, "igemm3_kernel<64,256,64,64,1>", true},
    {32, 128, 16, "igemm3_kernel<32,128,32,32,1>", true},   {128, 32, 16, "igemm3_kernel<128,32,32,32,1>", true},
    {128, 64, 16, "igemm3_kernel<128,64,64,32,2>", true},   {64, 64, 16, "igemm3_kernel<64,64,32,32,2>", true},
    {64, 128, 16, "igemm3_kernel<64,128,32,64,2>", true},   {32, 128, 16, "igemm3_kernel<32,128,32,32,2>", true},
    {128, 32, 16, "igemm3_kernel<128,32,32,32,2>", true},
    {128, 128, 16, "igemm3dw_kernel<128,128,64,64,false>", true, true}, {128, 64, 16, "igemm3dw_kernel<128,64,64,32,false>", true, true},
    {256, 64, 16, "igemm3dw_kernel<256,64,64,64,false>", true, true},   {64, 128, 16, "igemm3dw_kernel<64,128,32,64,false>", true, true},
    {64, 64, 16, "igemm3dw_kernel<64,64,32,32,false>", true, true},     {64, 256, 16, "igemm3dw_kernel<64,256,64,64,false>", true, true},
    {128, 64, 16, "igemm3dw_kernel<128,64,64,32,true>", true, true},    {256, 64, 16, "igemm3dw_kernel<256,64,64,64,true>", true, true},
    {64, 64, 16, "igemm3dw_kernel<64,64,32,32,true>", true, true},      {64, 128, 16, "igemm3dw_kernel<64,128,32,64,true>", true, true},
    {256, 64, 16, "igemm3s2_kernel<256,64,64,64>", true, false, true},  {128, 64, 16, "igemm3s2_kernel<128,64,64,32>", true, false, true},
    {128, 64, 16, "conv3p_kernel<128,64,64,32>", true, true, false, true},   {128, 128, 16, "conv3p_kernel<128,128,64,64>", true, true, false, true},
    {64, 64, 16, "conv3p_kernel<64,64,32,32>", true, true, false, true},
    {128, 64, 16, "conv3pp_kernel<0>", true, true, false, true},          {128, 64, 16, "conv3pp_kernel<1>", true, true, false, true},
    {128, 64, 16, "conv3g_kernel<128,64,64,32,2,false>", true, false, false, true, true}, {64, 64, 16, "conv3g_kernel<64,64,32,32,2,false>", true, false, false, true, true},
    {64, 128, 16, "conv3g_kernel<64,128,32,64,2,false>", true, false, false, true, true}, {128, 128, 16, "conv3g_kernel<128,128,64,64,1,false>", true, false, false, true, true},
    {128, 64, 16, "conv3h_kernel<128,64,64,32,1>", true, true, false, true, false, true},   {128, 128, 16, "conv3h_kernel<128,128,64,64,1>", true, true, false, true, false, true},
    {64, 64, 16, "conv3h_kernel<64,64,32,32,1>", true, true, false, true, false, true},      {256, 64, 16, "conv3h_kernel<256,64,64,64,1>", true, true, false, true, false, true},
    {128, 64, 16, "conv3g_kernel<128,64,64,32,3,true>", true, false, false, true, true, true},   {64, 64, 16, "conv3g_kernel<64,64,32,32,4,true>", true, false, false, true, true, true},
    {128, 128, 16, "conv3g_kernel<128,128,64,64,2,true>", true, false, false, true, true, true}, {64, 128, 16, "conv3g_kernel<64,128,32,64,3,true>", true, false, false, true, true, true},
    {128, 64, 32, "conv3h_kernel<128,64,64,32,2>", true, true, false, true, false, true}, {64, 64, 32, "conv3h_kernel<64,64,32,32,2>", true, true, false, true, false, true},
    {64, 64, 64, "conv3h_kernel<64,64,32,32,4>", true, true, false, true, false, true},
    {64, 128, 16, "conv3g_kernel<64,128,32,64,2,true,1>", true, false, false, true, true, true}, {64, 128, 16, "conv3g_kernel<64,128,32,64,4,true,1>", true, false, false, true, true, true},
    {128, 128, 16, "conv3g_kernel<128,128,64,64,2,true,1>", true, false, false, true, true, true}, {128, 256, 16, "conv3g_kernel<128,256,64,128,2,true,1>", true, false, false, true, true, true},
    {256, 64, 16, "conv3hr_kernel<256,64,64,64,1>", true, true, false, true, false, true}, {128, 64, 16, "conv3hr_kernel<128,64,64,32,1>", true, true, false, true, false, true},
    {64, 64, 32, "conv3hr_kernel<64,64,32,32,2>", true, true, false, true, false, true},
    {128, 128, 16, "conv3hr_kernel<128,128,64,64,1>", true, true, false, true, false, true},
};
// igemm3s2_kernel: the 7x(7->8)x4 stride-2 stem over a pre-padded dense image
static bool s2_ok(const IgemmDesc& d) {
    return d.Cin == 4 && d.ldx == 4 && d.ntaps == 56 && d.TW == 8 && d.in_sh == 2 && d.in_sw == 2 && d.tap_sh == 1 && d.tap_sw == 1 &&
           d.tap_h0 == 0 && d.tap_w0 == 0 && d.dsh * d.dsw == 1 && d.g_h0 == 0 && d.g_w0 == 0 && d.K == 224 && d.Kpad == 224 &&
           d.in_scale == nullptr && d.bn_in.acc == nullptr && d.x_bstride == (long)d.Hin * d.Win * 4 && d.Wg >= 64 &&
           (d.Wg - 1) * 2 + 8 <= d.Win && (d.Hg - 1) * 2 + 7 <= d.Hin;
}
// igemm3dw_kernel: dense 3x3 stride-1 SAME conv whose output pixel q reads input pixels q + dh*W + dw
static bool dw3_ok(const IgemmDesc& d) {
    return d.ntaps == 9 && d.TW == 3 && d.tap_sh == 1 && d.tap_sw == 1 && d.tap_h0 == -1 && d.tap_w0 == -1 && d.in_sh == 1 &&
           d.in_sw == 1 && d.dsh * d.dsw == 1 && d.Hin == d.Hg && d.Win == d.Wg && d.g_h0 == 0 && d.g_w0 == 0 && d.Cin % 16 == 0 &&
           d.K == 9 * d.Cin && d.Kpad == d.K && d.x_bstride == (long)d.Hin * d.Win * d.ldx && d.Hin >= 2 && d.Win >= 8;
}
bool igemm_tile_split(IgemmTile t) { return t >= 0 && t < TILE_AUTO && kTiles[t].split; }
bool igemm_tile_p3(IgemmTile t) { return t >= 0 && t < TILE_AUTO && kTiles[t].p3; }
// conv3h_kernel tiles can split K by the filter row (dh-split, split-K = 3 exactly); the three-deep-ring variant cannot
bool igemm_tile_dh_split(IgemmTile t) {
    return t >= 0 && t < TILE_AUTO && kTiles[t].p3 && kTiles[t].h && !kTiles[t].g && t != TILE_P3HR_256x64 && t != TILE_P3HR_128x64 && t != TILE_P3HR_64x64_C2 && t != TILE_P3HR_128x128;
}
bool igemm_p3_eligible(const IgemmDesc& d) { return dw3_ok(d) && d.w_split && d.dsh * d.dsw == 1 && d.Cin <= MAX_BN_C; }
static int tile_bm(IgemmTile t) { return (t >= 0 && t < TILE_AUTO) ? kTiles[t].bm : 0; }
static int tile_bn(IgemmTile t) { return (t >= 0 && t < TILE_AUTO) ? kTiles[t].bn : 0; }
int igemm_tile_bm(IgemmTile t) { return tile_bm(t); }
int igemm_tile_bn(IgemmTile t) { return tile_bn(t); }
int igemm_tile_bk(IgemmTile t) { return (t >= 0 && t < TILE_AUTO) ? kTiles[t].bk : 0; }
const char* igemm_tile_name(IgemmTile t) { return (t >= 0 && t < TILE_AUTO) ? kTiles[t].name : "igemm_kernel<?>"; }

static bool uniform_taps_for(const IgemmDesc& d, int bk) {
    return (d.ntaps > 1 ? (d.Cin % bk == 0) : true) && (d.K % bk == 0);
}
// kernels that take the group index of a grouped launch from their grid (common.h): igemm_kernel, igemm3_kernel, conv3h_kernel, conv3hr_kernel, conv3g_kernel
bool igemm_tile_grouped(IgemmTile t) {
    if (t < 0 || t >= TILE_AUTO) return false;
    const TileCfg& k = kTiles[t];
    if (k.s2 || (k.dw3 && !k.p3)) return false;                       // igemm3s2_kernel, igemm3dw_kernel
    if (k.p3 && !k.h && !k.g) return false;                            // conv3p_kernel / conv3pp_kernel
    return true;
}
bool igemm_tile_ok(const IgemmDesc& d, IgemmTile t) {
    if (t < 0 || t >= TILE_AUTO) return false;
    if (cur_group().G > 1 && !igemm_tile_grouped(t)) return false;
    const int bk = kTiles[t].bk;
    if (d.Kpad % bk) return false;
    if (kTiles[t].split && !d.w_split) return false;
    const bool mm_tile = t == TILE_P3GH_MM_64x128_K2 || t == TILE_P3GH_MM_64x128_K4 || t == TILE_P3GH_MM_128x128_K2 || t == TILE_P3GH_MM_128x256_K2;      // conv3g_kernel with the fused decoder tail as epilogue
    if (mm_tile != (d.mm_out != nullptr && kTiles[t].g)) return false;
    if (mm_tile && (d.Wg % kTiles[t].bm || (d.dsw * d.Cout) % kTiles[t].bn)) return false;          // a tile = one mask frame of one window
    if (d.mm_out != nullptr && !mm_tile) {  // fused decoder tail: igemm3_kernel tiles holding one mask frame of one window
        const bool b3 = kTiles[t].split && !kTiles[t].dw3 && !kTiles[t].s2 && !kTiles[t].p3;
        if (!b3 || kTiles[t].bm > 128 || kTiles[t].bn > 128 || d.Cout != 32 || d.dsh * d.dsw <= 1 || d.Wg % kTiles[t].bm ||
            (d.dsw * d.Cout) % kTiles[t].bn || d.splitk != 1 || d.in_scale || d.bn_in.acc || !d.mm_coeffs)
            return false;
    }
    if (kTiles[t].dw3 && !dw3_ok(d)) return false;
    if (kTiles[t].p3 && (d.xp3 == nullptr || (d.splitk != 1 && !(d.splitk == 3 && igemm_tile_dh_split(t))))) return false;
    if (kTiles[t].p3 && (kTiles[t].h ? (d.xp3_fmt != 1 || d.wh2 == nullptr) : d.xp3_fmt != 0)) return false;   // the planes' format decides the family
    if (!kTiles[t].p3 && d.xp3 != nullptr && d.x == nullptr) return false;      // only the planes were provided
    if (kTiles[t].g) return conv3g_ok(d);                                        // gathered operand tiles: any stride / tap set on the plane rows
    if (kTiles[t].p3) return true;                                               // (the producer's BN+ReLU is already in the planes)
    if (kTiles[t].s2 && !s2_ok(d)) return false;
    if ((d.in_scale || d.bn_in.acc) && !uniform_taps_for(d, bk)) return false;
    return true;
}

IgemmTile igemm_pick_tile(const IgemmDesc& d) {
    static const char* force = getenv("SAGEN_FORCE_TILE");               // tuning knob: IgemmTile index
    if (force && d.M > 128 && d.N >= 64 && igemm_tile_ok(d, (IgemmTile)atoi(force))) return (IgemmTile)atoi(force);
    static const bool fp32_only = getenv("SAGEN_FP32_ONLY") != nullptr;
    if (d.mm_out != nullptr) return (d.xp3 != nullptr && d.xp3_fmt == 1 && d.wh2 != nullptr && conv3g_ok(d)) ? TILE_P3GH_MM_64x128_K2 : TILE_B3_64x128;
    auto blocks = [&](IgemmTile t) { return (long)cdiv(d.M, tile_bm(t)) * cdiv(d.N, tile_bn(t)) * d.splitk; };
    const long want = 2 * 256;                 // >= 2 workgroups per CU
    if (d.w_split && !fp32_only) {             // the bf16x3 kernels are the faster family wherever their planes exist
        const bool pro = d.in_scale != nullptr || d.bn_in.acc != nullptr;
        if (s2_ok(d)) return TILE_B3S2_128x64;                                  // the 7x7/2 stem
        if (d.xp3 != nullptr && d.xp3_fmt == 1 && dw3_ok(d) && d.splitk == 1) {   // two fp16 planes: three products per multiply
            const long np = d.p3_np;
            if ((long)cdiv(np, 126) * cdiv(d.N, 64) >= 3 * 256) return TILE_P3H_128x64;
            return TILE_P3H_64x64;
        }
        if (d.xp3 != nullptr && dw3_ok(d) && d.splitk == 1) {                   // pre-split activation planes: LDS-DMA -> MFMA only
            const long np = d.p3_np;
            if (d.N <= 64) return cdiv(np, 128) >= want ? TILE_P3_128x64 : TILE_P3_64x64;
            if ((long)cdiv(np, 128) * cdiv(d.N, 64) >= want) return TILE_P3_128x64;
            return TILE_P3_64x64;
        }
        if (d.xp3 != nullptr && !dw3_ok(d) && d.splitk == 1 && conv3g_ok(d)) {  // planes + any other geometry: gathered operand tiles
            if (d.xp3_fmt == 1) return blocks(TILE_P3GH_128x64_K3) >= 256 + 128 ? TILE_P3GH_128x64_K3 : TILE_P3GH_64x64_K4;
            return blocks(TILE_P3G_128x64_K2) >= 256 + 128 ? TILE_P3G_128x64_K2 : TILE_P3G_64x64_K2;
        }
        if (dw3_ok(d) && (!pro || uniform_taps_for(d, 16))) {                   // 3x3 stride 1: shared horizontal taps
            if (d.N <= 64) return blocks(TILE_B3DW_128x64) >= want ? TILE_B3DW_128x64 : TILE_B3DWM_64x64;
            if (blocks(TILE_B3DW_128x128) >= 256 + 128) return TILE_B3DW_128x128;
            return blocks(TILE_B3DWM_128x64) >= 256 + 128 ? TILE_B3DWM_128x64 : TILE_B3DWM_64x64;
        }
        if (d.M <= 32) return TILE_B3_32x128;
        if (d.N <= 32) return TILE_B3_128x32;
        if (d.N <= 64) return blocks(TILE_B3_128x64) >= want ? TILE_B3_128x64 : TILE_B3_64x64;
        if (blocks(TILE_B3_128x128) >= 256 + 128) return TILE_B3_128x128;
        if (blocks(TILE_B3_64x128) >= want) return TILE_B3_64x128_K2;
        return TILE_B3_64x64_K2;
    }
    if (d.M <= 32) return TILE_32x128;
    if (d.N <= 32) return TILE_128x32;
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

int igemm_launch(const IgemmDesc& d_in, IgemmTile tile, hipStream_t s) {
    IgemmDesc d = d_in;
    if ((!d.x && !d.xp3) || !d.w || (!d.y && !d.splitk_ws)) return fail(SAGEN_ERR_NULL, "igemm: null operand");
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
    {   // 32-bit buffer addressing + exact host-side bounds analysis of the taps
        const int nb = d.M / (d.Hg * d.Wg) + (d.M % (d.Hg * d.Wg) ? 1 : 0);
        const long xb = (long)nb * d.x_bstride * 4, wb = (long)d.N * d.Kpad * 4;
        if (xb >= (1L << 31) || wb >= (1L << 31) || (d.w_split && wb / 2 * 3 >= (1L << 31)))
            return fail(SAGEN_ERR_UNSUPPORTED, "igemm: operand of %ld bytes exceeds 2 GiB buffer addressing (use a smaller batch)", xb > wb ? xb : wb);
        d.x_bytes = (unsigned)xb;
        d.w_bytes = (unsigned)wb;
        bool inside = true;
        for (int t = 0; t < d.ntaps && inside; ++t) {
            const int th = t / d.TW, dh = th * d.tap_sh + d.tap_h0, dw = (t - th * d.TW) * d.tap_sw + d.tap_w0;
            const long h_lo = (long)d.g_h0 * d.in_sh + dh, h_hi = (long)(d.g_h0 + d.Hg - 1) * d.in_sh + dh;
            const long w_lo = (long)d.g_w0 * d.in_sw + dw, w_hi = (long)(d.g_w0 + d.Wg - 1) * d.in_sw + dw;
            inside = h_lo >= 0 && h_hi < d.Hin && w_lo >= 0 && w_hi < d.Win;
        }
        d.no_bounds = inside ? 1 : 0;
        if ((d.in_scale || d.bn_in.acc) && (d.Cin % 16 || d.K % 16 || d.ntaps == 1 || d.Cin > MAX_BN_C))
            return fail(SAGEN_ERR_UNSUPPORTED, "igemm: input batch-norm needs a multi-tap conv with Cin %% 16 == 0 and Cin <= %d", MAX_BN_C);
        if (!inside && d.ntaps > 64) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: padded conv with %d taps (max 64)", d.ntaps);
    }
    if (tile == TILE_AUTO) tile = igemm_pick_tile(d);
    d.grp = cur_group();
    if (d.grp.G > 1 && !igemm_tile_grouped(tile)) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: %s has no grouped launch", igemm_tile_name(tile));
    if (!igemm_tile_ok(d, tile)) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: %s cannot run this problem (Kpad=%d Cin=%d)", igemm_tile_name(tile), d.Kpad, d.Cin);
    // wave-uniform tap per K tile: every K tile lies inside one tap (and there is no ragged K tail)
    d.uniform_taps = uniform_taps_for(d, kTiles[tile].bk) ? 1 : 0;
    if (d.ntaps > MAX_TAPS) return fail(SAGEN_ERR_UNSUPPORTED, "igemm: %d taps (max %d)", d.ntaps, MAX_TAPS);
    if (kTiles[tile].g) return conv3g_dispatch(d, tile, s);
    if (kTiles[tile].h) return conv3h_dispatch(d, tile, s);
    if (kTiles[tile].p3) return conv3p_dispatch(d, tile, s);
    if (kTiles[tile].s2) return igemm3s2_dispatch(d, tile, s);
    if (kTiles[tile].split) return igemm3_dispatch(d, tile, s);
    switch (tile) {
        case TILE_128x128: return launch_cfg<128, 128, 64, 64, 3, 16>(d, s);
        case TILE_128x64: return launch_cfg<128, 64, 64, 32, 3, 16>(d, s);
        case TILE_256x64: return launch_cfg<256, 64, 64, 64, 3, 16>(d, s);
        case TILE_64x64: return launch_cfg<64, 64, 32, 32, 3, 16>(d, s);
        case TILE_128x32: return launch_cfg<128, 32, 32, 32, 2, 16>(d, s);
        case TILE_32x128: return launch_cfg<32, 128, 32, 32, 2, 16>(d, s);
        case TILE_128x128_S2: return launch_cfg<128, 128, 64, 64, 2, 16>(d, s);
        case TILE_128x64_S2: return launch_cfg<128, 64, 64, 32, 2, 16>(d, s);
        case TILE_256x64_S2: return launch_cfg<256, 64, 64, 64, 2, 16>(d, s);
        case TILE_64x64_S2: return launch_cfg<64, 64, 32, 32, 2, 16>(d, s);
        case TILE_64x128: return launch_cfg<64, 128, 32, 64, 3, 16>(d, s);
        case TILE_64x128_S2: return launch_cfg<64, 128, 32, 64, 2, 16>(d, s);
        case TILE_64x256: return launch_cfg<64, 256, 64, 64, 3, 16>(d, s);
        case TILE_64x256_S2: return launch_cfg<64, 256, 64, 64, 2, 16>(d, s);
        case TILE_256x32: return launch_cfg<256, 32, 64, 32, 2, 16>(d, s);
        case TILE_64x64_K32: return launch_cfg<64, 64, 32, 32, 2, 32>(d, s);
        case TILE_64x128_K32: return launch_cfg<64, 128, 32, 64, 2, 32>(d, s);
        case TILE_128x64_K32: return launch_cfg<128, 64, 64, 32, 2, 32>(d, s);
        case TILE_128x128_K32: return launch_cfg<128, 128, 64, 64, 2, 32>(d, s);
        case TILE_32x128_K32: return launch_cfg<32, 128, 32, 32, 2, 32>(d, s);
        case TILE_128x32_K32: return launch_cfg<128, 32, 32, 32, 2, 32>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "igemm: bad tile id %d", (int)tile);
    }
}

// -----------------------------------------------------------------------------------------
// split-K reduction + bias + activation + (optional) row replication
// -----------------------------------------------------------------------------------------
// out[(m*rep + r)*ldy + n] = act(sum_z ws[z][m][n] + bias[n]).  One thread per 4 columns (N % 4 == 0) or per
// column; fully parallel over M x N.
template <int V>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws_, int splitk, int M, int N,
                                                            const float* __restrict__ bias, int relu,
                                                            float* __restrict__ y_, int ldy, int rep, float* __restrict__ amax_out_, const GroupInfo gi) {
    const float* __restrict__ ws = SAGEN_GRP(ws_);
    float* __restrict__ y = SAGEN_GRP(y_);
    float* __restrict__ amax_out = SAGEN_GRP(amax_out_);
    const int NV = N / V;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    // no early return: the lanes past the end of a partial last wave keep v = 0, skip their loads / stores and meet the others at ONE
    // converged igemm_publish_amax (its DPP / permlane reduction must not run under a partial EXEC mask: ADVICE r05)
    const bool ok = idx < (long)M * NV;
    const int m = ok ? (int)(idx / NV) : 0, n = ok ? (int)(idx - (long)m * NV) * V : 0;
    float v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = 0.f;
    if (ok) {
        const long MN = (long)M * N;
        const float* p = ws + (long)m * N + n;
        int zz = 0;
        if (V == 4)
            for (; zz + 4 <= splitk; zz += 4, p += 4 * MN) {          // four partials in flight
                const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + MN);
                const float4 t2 = *reinterpret_cast<const float4*>(p + 2 * MN), t3 = *reinterpret_cast<const float4*>(p + 3 * MN);
                v[0] += (t0.x + t1.x) + (t2.x + t3.x); v[1 % V] += (t0.y + t1.y) + (t2.y + t3.y);
                v[2 % V] += (t0.z + t1.z) + (t2.z + t3.z); v[3 % V] += (t0.w + t1.w) + (t2.w + t3.w);
            }
        for (; zz < splitk; ++zz, p += MN) {
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] += t.x; v[1 % V] += t.y; v[2 % V] += t.z; v[3 % V] += t.w;
            } else {
                v[0] += p[0];
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (bias) v[i] += bias[n + i];
            if (relu) v[i] = fmaxf(v[i], 0.f);
        }
        for (int r = 0; r < rep; ++r) {
            float* o = y + ((long)m * rep + r) * ldy + n;
            if (V == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
            else o[0] = v[0];
        }
    }
    if (amax_out != nullptr) {                                        // block-uniform condition: every lane of every wave arrives
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) a = fmaxf(a, fabsf(v[i]));
        igemm_publish_amax(amax_out, a);
    }
}

// The same for FEW outputs and MANY partials (weight gradients of small layers: 28 KB of gradient from 256 pixel ranges took 27 us
// with one thread walking all 256 partials of its four columns): eight threads per column group, each summing every eighth partial
// (four loads in flight), combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void splitk_reduce_sliced_kernel(const float* __restrict__ ws_, int splitk, int M, int N,
                                                                   const float* __restrict__ bias, int relu,
                                                                   float* __restrict__ y_, int ldy, int rep, float* __restrict__ amax_out_, const GroupInfo gi) {
    const float* __restrict__ ws = SAGEN_GRP(ws_);
    float* __restrict__ y = SAGEN_GRP(y_);
    float* __restrict__ amax_out = SAGEN_GRP(amax_out_);
    __shared__ float4 red[8][32];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int NV = N / 4;
    const long idx = (long)blockIdx.x * 32 + c;
    const bool ok = idx < (long)M * NV;
    const int m = ok ? (int)(idx / NV) : 0, n = ok ? (int)(idx - (long)m * NV) * 4 : 0;
    const long MN = (long)M * N;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        const float* p = ws + (long)m * N + n + (long)sl * MN;
        int z = sl;
        for (; z + 24 < splitk; z += 32, p += 32 * MN) {
            const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + 8 * MN);
            const float4 t2 = *reinterpret_cast<const float4*>(p + 16 * MN), t3 = *reinterpret_cast<const float4*>(p + 24 * MN);
            v.x += (t0.x + t1.x) + (t2.x + t3.x); v.y += (t0.y + t1.y) + (t2.y + t3.y);
            v.z += (t0.z + t1.z) + (t2.z + t3.z); v.w += (t0.w + t1.w) + (t2.w + t3.w);
        }
        for (; z < splitk; z += 8, p += 8 * MN) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
    }
    red[sl][c] = v;
    __syncthreads();
    if (sl == 0 && ok) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 t = red[k][c]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        for (int r = 0; r < rep; ++r) *reinterpret_cast<float4*>(y + ((long)m * rep + r) * ldy + n) = v;
    }
    if (amax_out != nullptr)
        igemm_publish_amax(amax_out, (sl == 0 && ok) ? fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) : 0.f);
}

// Same sum, plus the per-channel (sum, sumsq) of the raw sums over each block of SPLITK_RB rows
// (training-mode batch-norm statistics of a split-K conv).  Thread t owns 4 columns n = 4*(t % CN4) and rows
// (t / CN4) + i*(256 / CN4) of its row block; grid (row blocks, column blocks of 4*CN4).
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ ws_, int splitk, int M, int N,
                                                                  float* __restrict__ y_, int ldy, double* __restrict__ stats_,
                                                                  int CN4, int RB, const GroupInfo gi) {
    const float* __restrict__ ws = SAGEN_GRP(ws_);
    float* __restrict__ y = SAGEN_GRP(y_);
    double* __restrict__ stats = SAGEN_GRP(stats_);
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x;
    const int cl = tid % CN4, rl = tid / CN4, RL = 256 / CN4;
    const int n = (blockIdx.y * CN4 + cl) * 4;
    const int r0 = blockIdx.x * RB;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    const long MN = (long)M * N;
    if (n < N) {
        for (int r = rl; r < RB; r += RL) {
            const int m = r0 + r;
            if (m >= M) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* p = ws + (long)m * N + n;
            int zz = 0;
            for (; zz + 4 <= splitk; zz += 4, p += 4 * MN) {      // four partials in flight (a serial loop is latency-bound)
                const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + MN);
                const float4 t2 = *reinterpret_cast<const float4*>(p + 2 * MN), t3 = *reinterpret_cast<const float4*>(p + 3 * MN);
                v.x += (t0.x + t1.x) + (t2.x + t3.x); v.y += (t0.y + t1.y) + (t2.y + t3.y);
                v.z += (t0.z + t1.z) + (t2.z + t3.z); v.w += (t0.w + t1.w) + (t2.w + t3.w);
            }
            for (; zz < splitk; ++zz, p += MN) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
            *reinterpret_cast<float4*>(y + (long)m * ldy + n) = v;
        }
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    if (rl == 0 && n < N) {
        for (int k = 1; k < RL; ++k) {
            const float4 a = red[0][k * CN4 + cl], b = red[1][k * CN4 + cl];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        atomicAdd(&stats[n + 0], (double)s.x); atomicAdd(&stats[n + 1], (double)s.y);
        atomicAdd(&stats[n + 2], (double)s.z); atomicAdd(&stats[n + 3], (double)s.w);
        atomicAdd(&stats[N + n + 0], (double)q.x); atomicAdd(&stats[N + n + 1], (double)q.y);
        atomicAdd(&stats[N + n + 2], (double)q.z); atomicAdd(&stats[N + n + 3], (double)q.w);
    }
}

int splitk_reduce_launch(const float* ws, int splitk, int M, int N, const float* bias, int relu,
                         float* y, int ldy, int rep, double* stats, hipStream_t s, float* amax_out) {
    const GroupInfo gi = cur_group();
    if (stats) {
        if (amax_out) return fail(SAGEN_ERR_UNSUPPORTED, "split-K reduce: statistics and amax_out together");
        if (N % 4 || ldy % 4 || rep != 1 || bias || relu)
            return fail(SAGEN_ERR_UNSUPPORTED, "split-K reduce with statistics needs N %% 4 == 0 and a plain epilogue");
        const int CN4 = N >= 256 ? 64 : (N >= 128 ? 32 : 16);         // 256 / 128 / 64 columns per workgroup
        // rows per workgroup: fatter workgroups for big M (fewer fp64 atomics), never fewer than ~2 workgroups per CU
        const int ncb = cdiv(N, CN4 * 4);
        int RB = SPLITK_RB;
        while (RB < 128 && (long)cdiv(M, 2 * RB) * ncb >= 512) RB *= 2;
        dim3 grid(cdiv(M, RB), ncb, gi.G);
        hipLaunchKernelGGL(splitk_reduce_stats_kernel, grid, dim3(256), 0, s, ws, splitk, M, N, y, ldy, stats, CN4, RB, gi);
    } else if (N % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)y % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) && splitk >= 16 &&
               (long)M * (N / 4) <= 65536) {
        hipLaunchKernelGGL(splitk_reduce_sliced_kernel, dim3(cdiv((long)M * (N / 4), 32), 1, gi.G), dim3(256), 0, s, ws, splitk, M, N, bias,
                           relu, y, ldy, rep, amax_out, gi);
    } else if (N % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)y % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0)) {
        hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(cdiv((long)M * (N / 4), 256), 1, gi.G), dim3(256), 0, s, ws, splitk, M, N, bias,
                           relu, y, ldy, rep, amax_out, gi);
    } else {
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cdiv((long)M * N, 256), 1, gi.G), dim3(256), 0, s, ws, splitk, M, N, bias, relu,
                           y, ldy, rep, amax_out, gi);
    }
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// conv2d_transpose in SCATTER form (tf.nn.conv2d_transpose, core.py:96-153; the mask decoder's deconv5 .. deconv2 at inference,
// model.py:302-305): out[b, y*sh+p, x*sw+q, o] += in[b, y, x, c] * W[p, q, o, c] (written below for stride 1).  As a stride-1 conv over the OUTPUT grid (igemm's depth-to-space
// form) every output pixel contracts all kh*kw taps, of which only those landing inside the small input are non-zero - 5.4 of 15 for
// deconv5 (3x6 -> 5x10), 7.7 of 15 for deconv4: the rest multiplies padding.  The scatter form contracts each INPUT pixel once,
//     T[m = (b, y, x)][n = (p, q, o)] = sum_c in[m][c] * W[p][q][o][c]          (a plain GEMM: M = B*Hin*Win, K = Cin, N = kh*kw*Cout)
// and this pass gathers out[b, y', x', o] = act(bias[o] + sum_z sum_{p, q valid} T_z[(b, y'-p, x'-q)][(p, q, o)]) in a fixed order -
// it IS the split-K reducer of that GEMM.  36 % / 51 % of the matrix work and of the filter streaming of the conv form.
// -----------------------------------------------------------------------------------------
// PT x QT = the most taps per dimension that can reach one output pixel (ceil(kh / sh) x ceil(kw / sw)): the loops are fully unrolled
// and every load is issued before the first add - with run-time loops each thread walked its 5..15 taps one memory round trip at
// a time (12 us for the 9 MB of deconv5's partials)
template <int PT, int QT>
__global__ __launch_bounds__(256) void deconv_gather_kernel(const float* __restrict__ ws_, int splitk, const DeconvGather g_, const GroupInfo gi) {
    const float* __restrict__ ws = SAGEN_GRP(ws_);
    DeconvGather g = g_;
    g.y = SAGEN_GRP(g.y); g.amax_out = SAGEN_GRP(g.amax_out);
    const int Hout = g.Hin * g.sh + g.kh - g.sh, Wout = g.Win * g.sw + g.kw - g.sw, C4 = g.Cout >> 2;
    const int rows = g.y1 - g.y0;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (idx < (long)g.B * rows * Wout * C4) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int xo = (int)(pix % Wout), yo = g.y0 + (int)((pix / Wout) % rows), b = (int)(pix / ((long)Wout * rows));
        const long M = (long)g.B * g.R * g.Win, N = (long)g.kh * g.kw * g.Cout;
        float4 acc = g.bias ? *reinterpret_cast<const float4*>(g.bias + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        // taps (p, q) with (yo - p, xo - q) divisible by the strides and inside the band / the row: p = p_lo + i*sh <= p_hi, q alike
        const int y_in0 = g.in_row0 * g.sh, y_in1 = (g.in_row0 + g.R - 1) * g.sh;       // yo - p must lie in [y_in0, y_in1]
        int p_lo = yo % g.sh;
        const int p_hi = min(g.kh - 1, yo - y_in0);
        if (yo - p_lo > y_in1) p_lo += (yo - p_lo - y_in1 + g.sh - 1) / g.sh * g.sh;
        int q_lo = xo % g.sw;
        const int q_hi = min(g.kw - 1, xo);
        const int x_in1 = (g.Win - 1) * g.sw;
        if (xo - q_lo > x_in1) q_lo += (xo - q_lo - x_in1 + g.sw - 1) / g.sw * g.sw;
        const long zstride = M * N;
        const float* const base = ws + 4 * c4;
        for (int z = 0; z < splitk; ++z) {
            float4 v[PT][QT];
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int p = p_lo + i * g.sh;
                const bool pok = p <= p_hi;
                const int yi = pok ? (yo - p) / g.sh - g.in_row0 : 0;
#pragma unroll
                for (int j = 0; j < QT; ++j) {
                    const int q = q_lo + j * g.sw;
                    const bool ok = pok && q <= q_hi;
                    const int xi = ok ? (xo - q) / g.sw : 0;
                    const float* t = base + z * zstride + (((long)b * g.R + yi) * g.Win + xi) * N + (long)((ok ? p : 0) * g.kw + (ok ? q : 0)) * g.Cout;
                    v[i][j] = ok ? *reinterpret_cast<const float4*>(t) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < QT; ++j) { acc.x += v[i][j].x; acc.y += v[i][j].y; acc.z += v[i][j].z; acc.w += v[i][j].w; }
        }
        if (g.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        *reinterpret_cast<float4*>(g.y + (((long)b * Hout + yo) * Wout + xo) * g.ldy + 4 * c4) = acc;
        amax = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
    if (g.amax_out != nullptr) igemm_publish_amax(g.amax_out, amax);
}

int deconv_gather_launch(const float* ws, int splitk, const DeconvGather& g, hipStream_t s) {
    if (!ws || !g.y) return fail(SAGEN_ERR_NULL, "deconv_gather: null argument");
    if (g.Cout % 4 || g.ldy % 4 || ((uintptr_t)g.y % 16) || (g.bias && ((uintptr_t)g.bias % 16)))
        return fail(SAGEN_ERR_UNSUPPORTED, "deconv_gather: Cout / ldy must be multiples of 4 and the pointers 16-byte aligned");
    const int Hout = g.Hin * g.sh + g.kh - g.sh;
    if (g.y0 < 0 || g.y1 > Hout || g.y1 <= g.y0 || g.in_row0 < 0 || g.R <= 0 || g.in_row0 + g.R > g.Hin)
        return fail(SAGEN_ERR_SHAPE, "deconv_gather: rows [%d, %d) of %d / input band [%d, +%d) of %d", g.y0, g.y1, Hout, g.in_row0, g.R, g.Hin);
    const long total = (long)g.B * (g.y1 - g.y0) * (g.Win * g.sw + g.kw - g.sw) * (g.Cout / 4);
    const int pt = cdiv(g.kh, g.sh), qt = cdiv(g.kw, g.sw);
    const GroupInfo gi = cur_group();
    const dim3 grid(cdiv(total, 256), 1, gi.G);
    if (pt <= 2 && qt <= 2) hipLaunchKernelGGL((deconv_gather_kernel<2, 2>), grid, dim3(256), 0, s, ws, splitk, g, gi);
    else if (pt <= 2 && qt <= 3) hipLaunchKernelGGL((deconv_gather_kernel<2, 3>), grid, dim3(256), 0, s, ws, splitk, g, gi);
    else if (pt <= 3 && qt <= 5) hipLaunchKernelGGL((deconv_gather_kernel<3, 5>), grid, dim3(256), 0, s, ws, splitk, g, gi);
    else return fail(SAGEN_ERR_UNSUPPORTED, "deconv_gather: %d x %d taps per output pixel", pt, qt);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// filter repacking
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int ntaps, int cin_src, int cin_pad,
                                                        int cout, float* __restrict__ wp, int Npad, int Kpad, int tw_src,
                                                        int tw_pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Npad * Kpad) return;
    const int n = (int)(idx / Kpad), k = (int)(idx - (long)n * Kpad);
    float v = 0.f;
    if (n < cout && k < ntaps * cin_pad) {
        int tap = k / cin_pad;
        const int c = k - tap * cin_pad;
        bool ok = c < cin_src;
        if (tw_pad > 0) {                      // packed tap rows of tw_pad over source rows of tw_src
            const int th = tap / tw_pad, tw = tap - th * tw_pad;
            ok = ok && tw < tw_src;
            tap = th * tw_src + tw;
        }
        if (ok) v = w[((long)tap * cin_src + c) * cout + n];
    }
    wp[idx] = v;
}

int pack_conv_launch(const float* w_hwio, int ntaps, int cin_src, int cin_pad, int cout, float* wp, int Npad,
                     int Kpad, hipStream_t s, int tw_src, int tw_pad) {
    const long total = (long)Npad * Kpad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w_hwio, ntaps, cin_src, cin_pad,
                       cout, wp, Npad, Kpad, tw_src, tw_pad);
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

// -----------------------------------------------------------------------------------------
// all filter packs of a context in ONE launch (the training step re-packs every filter every step: 110 small launches, 0.9 ms)
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ float pack_source(const PackJob& j, int n, int k) {
    const float* w = j.src;
    switch (j.kind) {
        case PACK_CONV: {          // Wp[n][(tap, c)] = W_hwio[tap][c][n]
            const int ntaps = j.p[0], cin_src = j.p[1], cin_pad = j.p[2], cout = j.p[3], tw_src = j.p[4], tw_pad = j.p[5];
            if (n >= cout || k >= ntaps * cin_pad) return 0.f;
            int tap = k / cin_pad;
            const int c = k - tap * cin_pad;
            bool ok = c < cin_src;
            if (tw_pad > 0) {
                const int th = tap / tw_pad, tw = tap - th * tw_pad;
                ok = ok && tw < tw_src;
                tap = th * tw_src + tw;
            }
            return ok ? w[((long)tap * cin_src + c) * cout + n] : 0.f;
        }
        case PACK_DECONV: {        // n = (ry, rx, o), k = (dp, dq, c): W[ry + sh*dp][rx + sw*dq][o][c]
            const int kh = j.p[0], kw = j.p[1], cout = j.p[2], cin = j.p[3], sh = j.p[4], sw = j.p[5], ntw = j.p[6];
            const int nth = (kh + sh - 1) / sh;
            if (n >= sh * sw * cout || k >= nth * ntw * cin) return 0.f;
            const int ry = n / (sw * cout), rx = (n / cout) % sw, o = n % cout;
            const int tap = k / cin, c = k - tap * cin;
            const int dp = tap / ntw, dq = tap - dp * ntw;
            const int pp = ry + sh * dp, q = rx + sw * dq;
            return (pp < kh && q < kw) ? w[(((long)pp * kw + q) * cout + o) * cin + c] : 0.f;
        }
        case PACK_DECONV_PHASE: {  // n = o, k = (dp, dq, c) over the taps of phase (ry, rx) only: W[ry + 2*dp][rx + 2*dq][o][c]
            const int kh = j.p[0], kw = j.p[1], cout = j.p[2], cin = j.p[3], ry = j.p[4], rx = j.p[5], ntw = j.p[6];
            const int nth = (kh - ry + 1) / 2;
            if (n >= cout || k >= nth * ntw * cin) return 0.f;
            const int tap = k / cin, c = k - tap * cin;
            const int dp = tap / ntw, dq = tap - dp * ntw;
            const int pp = ry + 2 * dp, q = rx + 2 * dq;
            return (pp < kh && q < kw) ? w[(((long)pp * kw + q) * cout + n) * cin + c] : 0.f;
        }
        case PACK_FLIPT: {         // Wp[n = ci][(tap', co)] = W_hwio[ntaps-1-tap'][ci][co]
            const int ntaps = j.p[0], cin = j.p[1], cout = j.p[2];
            if (n >= cin || k >= ntaps * cout) return 0.f;
            const int tap = k / cout, co = k - tap * cout;
            return w[((long)(ntaps - 1 - tap) * cin + n) * cout + co];
        }
        default: {                 // PACK_ROWS: row-major [rows][cols] -> [rows][Kpad]
            const int rows = j.p[0], cols = j.p[1];
            return (n < rows && k < cols) ? w[(long)n * cols + k] : 0.f;
        }
    }
}

// writes elements (n, k .. k+3) of job j: the fp32 pack and the three bf16 planes, tiled [Kpad/16][plane][N][16] (same arithmetic as
// pack_split_kernel, igemm3.hip)
__device__ __forceinline__ void pack_store4(const PackJob& j, int n, int k, const float v[4]) {
    const long idx = (long)n * j.Kpad + k;
    *reinterpret_cast<float4*>(j.dst + idx) = make_float4(v[0], v[1], v[2], v[3]);
    __bf16* w3 = reinterpret_cast<__bf16*>(j.dst + (long)j.N * j.Kpad);
    __bf16 h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        const float r1 = v[e] - (float)h[e];
        m[e] = (__bf16)r1;
        l[e] = (__bf16)(r1 - (float)m[e]);
    }
    const long o = ((long)(k >> 4) * 3 * j.N + n) * 16 + (k & 15);
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<bf16x4_t*>(w3 + o) = bf16x4_t{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<bf16x4_t*>(w3 + o + (long)j.N * 16) = bf16x4_t{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<bf16x4_t*>(w3 + o + (long)j.N * 32) = bf16x4_t{l[0], l[1], l[2], l[3]};
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, int njobs, int block0) {
    // job of this block: the last one whose first_block <= block (wave-uniform binary search); block0: a launch over a suffix / prefix
    // of the table keeps the table's block numbering
    const int block = (int)blockIdx.x + block0;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const int blk = block - j.first_block;
    if (j.kind == PACK_CONV) {
        // HWIO -> [N][K] is a transpose: a 32 (k) x 32 (n) tile per block, read along n (the source's fast index: 128-byte runs;
        // thread-per-destination-element reads fetched 335 MB for 123 MB of variables), written along k
        __shared__ float tile[32][33];
        const int tiles_k = (j.Kpad + 31) >> 5;
        const int tn = blk / tiles_k, tk = blk - tn * tiles_k;
        const int n0 = tn * 32, k0 = tk * 32;
        const int nl = threadIdx.x & 31, kl = threadIdx.x >> 5;
        if ((j.p[2] & 31) == 0 && j.p[5] == 0) {       // the tile lies inside ONE tap (cin_pad % 32 == 0, no tap-row padding): one division per workgroup
            const int cin_src = j.p[1], cin_pad = j.p[2], cout = j.p[3];
            const int tap = k0 / cin_pad, c0 = k0 - tap * cin_pad;
            const bool in = n0 + nl < cout && tap < j.p[0];
            const float* w = j.src + ((long)tap * cin_src + c0) * cout + n0 + nl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = kl + 8 * r;
                tile[c][nl] = in && c0 + c < cin_src ? w[(long)c * cout] : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[kl + 8 * r][nl] = pack_source(j, n0 + nl, k0 + kl + 8 * r);
        }
        __syncthreads();
        const int n = n0 + (threadIdx.x >> 3), k = k0 + 4 * (threadIdx.x & 7);
        if (n >= j.N || k >= j.Kpad) return;            // (Kpad % 16 == 0: a group of four is inside or outside as a whole)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[4 * (threadIdx.x & 7) + e][threadIdx.x >> 3];
        pack_store4(j, n, k, v);
        return;
    }
    const unsigned idx = ((unsigned)blk * 256 + threadIdx.x) * 4;              // (N * Kpad < 2^31 for every filter of the network)
    if (idx >= (unsigned)j.N * (unsigned)j.Kpad) return;
    const int n = (int)(idx / (unsigned)j.Kpad), k = (int)(idx - (unsigned)n * (unsigned)j.Kpad);      // Kpad % 16 == 0: the four elements share the row
    float v[4];
    if (j.kind == PACK_FLIPT && (j.p[2] & 3) == 0) {            // four consecutive co of one tap: one 16-byte load
        const int ntaps = j.p[0], cin = j.p[1], cout = j.p[2];
        const int tap = k / cout, co = k - tap * cout;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < cin && tap < ntaps) q = *reinterpret_cast<const float4*>(j.src + ((long)(ntaps - 1 - tap) * cin + n) * cout + co);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack_source(j, n, k + e);
    }
    pack_store4(j, n, k, v);
}

int pack_multi_launch(const PackJob* jobs_dev, int njobs, int nblocks, hipStream_t s, int block0) {
    if (!jobs_dev || njobs <= 0 || nblocks <= 0) return fail(SAGEN_ERR_NULL, "pack_multi: no jobs");
    hipLaunchKernelGGL(pack_multi_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, njobs, block0);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
