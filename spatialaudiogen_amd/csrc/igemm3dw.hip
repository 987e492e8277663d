// bf16x3 contraction for dense 3x3 stride-1 SAME convolutions (reference op: tf.nn.convolution, core.py:206, as used by
// the ResNet trunk, resnet.py / model.py:181-205).  Arithmetic and operand staging as igemm3.hip; see below for what
// is shared between the taps.
#include "igemm3_common.h"

namespace sagen {

// ------------------------------------------------------------------------------------------------------------
// igemm3dw_kernel: 3x3 stride-1 SAME convolutions - the three horizontal taps of a filter row share ONE activation tile
// ------------------------------------------------------------------------------------------------------------
// In the generic kernel every activation element is loaded, batch-normalised and split once per TAP (nine times per
// workgroup), and that operand-split VALU - not the matrix pipe - bounds it.  For a dense stride-1 3x3
// conv the input pixel of output pixel q under tap (dh, dw) is simply q + dh*W + dw in the flattened [B*H*W] pixel
// index, so the tile for (dh, channel chunk) is staged ONCE with one halo pixel on either side (BM + 2 rows), and the
// three dw taps read their fragments from it at row offsets 0 / 1 / 2.  What the flattened shift gets wrong - the
// pixel left of column 0 and right of column W-1 is padding, not the neighbouring image row - is repaired by the
// LDS layout: the tile keeps one extra, permanently zero slot between consecutive image rows, so the shifted read of an
// edge pixel lands on a zero without any per-fragment select.
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
    constexpr int AR = BM + 2 + BM / 8 + 2;             // slots per A plane: BM + 2 pixels + one zero gap per image row (W >= 8)
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

    for (int i = tid; i < 2 * A_ST; i += 256) a_stage[i] = 0.f;      // the gap slots stay zero for the whole kernel
    igemm_setup<BM>(d, m0, tid, true, s_row, s_tapb, s_bn);          // (ends with a barrier)

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.x, 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // ---- activation loader: tile row rho <-> flattened pixel q = m0 - 1 + rho; chunk = 4 channels ----
    const int kc4 = tid & 3;
    const int W = d.Win, H = d.Hin;
    // slot of flattened pixel q in the tile: its distance from pixel m0-1 plus one gap per image row crossed
    const int rid0 = (m0 - 1 + W) / W - 1;              // row index (over the whole batch) of pixel m0-1; -1 for pixel -1
    auto slot_of = [&](int q) { return (q - (m0 - 1)) + ((q + W) / W - 1 - rid0); };
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
        const int sl = slot_of(q);
        a_wofs[c] = sl * 8 + 4 * ((kc4 >> 1) ^ ((sl >> 3) & 1)) + 2 * (kc4 & 1);
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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int sc = slot_of(m0 + wm * WM + i * 32 + li);          // slot of the centre tap
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int sl = sc + dwi - 1;
            a_foff[dwi][i] = sl * 8 + 4 * (kk ^ ((sl >> 3) & 1));
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

    if (!igemm_epilogue_rows<BM, BN, WM, WN, 2 * A_ST + 2 * B_ST>(d, acc, s_row, smem, n0, tid))
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

int igemm3dw_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s) {
    switch (tile) {
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
        default: return fail(SAGEN_ERR_UNSUPPORTED, "igemm3dw: bad tile id %d", (int)tile);
    }
}

}  // namespace sagen
