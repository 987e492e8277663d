// bf16x3 contraction for the ResNet stem: 7x7 stride-2 convolution over the zero-bordered 4-channel image (reference:
// resnet.py:133 `conv1`, tf.nn.convolution core.py:206; SAME 7x7/2 == VALID on the padded image).
//
// K is ordered (dh, dw, c) with the seven horizontal taps padded to eight (a zero filter tap), so the K range of one filter
// row dh is, for every output pixel, ONE contiguous run of 8 pixels x 4 channels of input row 2*ho + dh, and
// neighbouring output pixels read runs that are shifted by two pixels.  Per filter row the workgroup therefore stages
// the input pixels its BM output pixels need exactly once - one 16-byte pixel per thread-chunk, split into the three
// bf16 planes - at LDS slot 2*(q - m0) + 8*(output rows crossed) (+ dw), and the MFMA A fragment of output pixel q for
// taps dw = 4s + 2g, 4s + 2g + 1 is the 16-byte read at slot(q) + 4s + 2g: contiguous across lanes, no swizzle.
// The generic kernel loads and splits every input pixel ~3.5 times per filter row (once per tap) with a per-chunk tap
// table lookup; here it is once, with no address arithmetic in the loop.  Arithmetic, filter planes and epilogue
// as igemm3.hip.
#include "igemm3_common.h"

namespace sagen {

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm3s2_body(const IgemmDesc& d) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int GAP = 8;                               // slots between the runs of consecutive output rows (even, >= 6)
    constexpr int MAXROWS = BM / 64 + 1;                 // output rows a tile can touch (Wg >= 64)
    constexpr int AS = 2 * BM + GAP * MAXROWS + 8;       // pixel slots per A plane
    constexpr int NCS = (AS + 255) / 256;                // pixel chunks per thread per filter row
    constexpr int NBC = (2 * 6 * BN + 255) / 256;        // filter chunks per thread per filter row (two K tiles of 16)
    constexpr int A_PL = AS * 2, B_PL = BN * 8;          // floats per plane (8 B per pixel; 32 B per filter row)
    constexpr int A_ST = 3 * A_PL, B_ST = 2 * 3 * B_PL;
    constexpr int NM1 = 6 * MT * NT;                     // MFMAs per K tile of 16

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

    // ---- pixel loader: slot sigma = tid + 256*c -> (output row j of the tile, input column) ----
    const int Wg = d.Wg, Hg = d.Hg, Win = d.Win, Hin = d.Hin;
    const int R0 = m0 / Wg, wo0 = m0 - R0 * Wg;          // first output row (over the batch) and column of the tile
    unsigned a_voff[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
        const int sg = tid + 256 * c;
        a_voff[c] = OOB;
        // row j starts at slot sb_j: sb_0 = 0 (column 2*wo0), sb_j = 2*((R0+j)*Wg - m0) + GAP*j
        int j = 0, sb = 0, col0 = 2 * wo0;
#pragma unroll
        for (int jj = 1; jj < MAXROWS; ++jj) {
            const int s2 = 2 * ((R0 + jj) * Wg - m0) + GAP * jj;
            if (sg >= s2) { j = jj; sb = s2; col0 = 0; }
        }
        const int col = col0 + (sg - sb);
        const int R = R0 + j;
        const int b = R / Hg, ho = R - b * Hg;
        if (sg < AS && col < Win)                        // (rows past the batch fall outside the buffer -> zero fill)
            a_voff[c] = (unsigned)((((long)b * Hin + 2 * ho) * Win + col) * 16);
    }
    // ---- filter loader: the two K tiles of a filter row are contiguous in the tiled planes: chunk q = tid + 256*c ->
    //      K tile q / (6*BN), plane, row, 16-B half ----
    unsigned b_voff[NBC];
    int b_wofs[NBC];
    bool b_active[NBC];
#pragma unroll
    for (int c = 0; c < NBC; ++c) {
        const int q = tid + 256 * c;
        const int kt = q / (6 * BN), q1 = q - kt * (6 * BN);
        const int pl = q1 / (2 * BN), rem = q1 - pl * (2 * BN);
        const int r = rem >> 1, half = rem & 1;
        const int n = n0 + r;
        b_active[c] = q < 12 * BN;
        b_voff[c] = (b_active[c] && n < d.N) ? (unsigned)((((long)kt * 3 + pl) * d.N + n) * 32 + 16 * half) : OOB;
        b_wofs[c] = (kt * 3 + pl) * B_PL + r * 8 + 4 * (half ^ ((r >> 3) & 1));
    }

    // K range of this split, in filter rows
    const int G = 7;
    const int gper = (G + d.splitk - 1) / d.splitk;
    const int g0 = z * gper;
    const int g1 = min(G, g0 + gper);
    const int ngroups = max(g1 - g0, 0);

    int la_g = g0, lb_g = g0;                            // next filter row to load pixels / filters for
    unsigned i_tb = 0, i_apast = 0, i_kbyte = 0, i_bpast = 0;
    auto begin_a = [&]() {
        i_apast = la_g >= g1 ? 1u : 0u;
        i_tb = (unsigned)(la_g * Win * 16);
        ++la_g;
    };
    auto begin_b = [&]() {
        i_bpast = lb_g >= g1 ? 1u : 0u;
        i_kbyte = (unsigned)(2 * lb_g) * (unsigned)(d.N * 96);
        ++lb_g;
    };
    f32x4 araw[NCS], braw[2][NBC];
#pragma unroll
    for (int c = 0; c < NCS; ++c) araw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < NBC; ++c) braw[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int c) {
#ifndef SAGEN_ABLATE_A
        araw[c] = bload16(x_rsrc, (a_voff[c] + i_tb) | (i_apast << 31));
#endif
    };
    auto load_b = [&](int c, f32x4& dst) {
#ifndef SAGEN_ABLATE_B
        dst = bload16(w_rsrc, (b_voff[c] + i_kbyte) | (i_bpast << 31));
#endif
    };
    auto store_b = [&](int c, const f32x4& src, float* st) {
        if (NBC * 256 == 12 * BN || b_active[c]) *reinterpret_cast<f32x4*>(st + b_wofs[c]) = src;
    };
    auto convert_chunk = [&](int c, float* st) {
        float v0 = araw[c][0], v1 = araw[c][1], v2 = araw[c][2], v3 = araw[c][3];
        const int sg = tid + 256 * c;
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) {
            u32x2 pk;
            pk[0] = split_pair(v0, v1);
            pk[1] = split_pair(v2, v3);
            if (NCS * 256 == AS || sg < AS) *reinterpret_cast<u32x2*>(st + lv * A_PL + sg * 2) = pk;
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
    // fragment addressing: float offset of slot(q) + 2*kk inside a plane; K tile s adds 4 slots (= 8 floats)
    int a_foff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = m0 + wm * WM + i * 32 + li;
        const int slot = 2 * (q - m0) + GAP * (q / Wg - R0) + 2 * kk;
        a_foff[i] = min(slot, AS - 8) * 2;               // (rows past M read garbage slots, never stored)
    }
    const int b_foff = (wn * WN + li) * 8 + 4 * (kk ^ ((li >> 3) & 1));

    // ---- pipeline fill: filter row g0 in stage 0, filters of row g0+1 in registers ----
    begin_a();
#pragma unroll
    for (int c = 0; c < NCS; ++c) load_a(c);
    begin_b();
#pragma unroll
    for (int c = 0; c < NBC; ++c) load_b(c, braw[0][c]);
    begin_b();
#pragma unroll
    for (int c = 0; c < NBC; ++c) load_b(c, braw[1][c]);
#pragma unroll
    for (int c = 0; c < NCS; ++c) convert_chunk(c, a_stage);
#pragma unroll
    for (int c = 0; c < NBC; ++c) store_b(c, braw[0][c], b_stage);
    lds_barrier();

    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm
    auto gstep = [&](auto gp_tag) {
        constexpr int GP = decltype(gp_tag)::value;
        const float* acur = a_stage + GP * A_ST;
        float* anxt = a_stage + (GP ^ 1) * A_ST;
        const float* bcur = b_stage + GP * B_ST;
        float* bnxt = b_stage + (GP ^ 1) * B_ST;
        begin_a();                                       // pixels of the next filter row: loaded early, converted late
        begin_b();                                       // filters two rows ahead
        // jobs in issue order: pixel loads | filter stores (row g+1) | filter loads (row g+2) | conversions
        constexpr int NJ = NCS + NBC + NBC + NCS;
        constexpr int NMT = 2 * NM1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 aq[3][MT], bq[3][NT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MT; ++i) aq[pl][i] = *reinterpret_cast<const bf16x8*>(acur + pl * A_PL + a_foff[i] + 8 * ks);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bq[pl][j] = *reinterpret_cast<const bf16x8*>(bcur + (ks * 3 + pl) * B_PL + j * 32 * 8 + b_foff);
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
                        const int idx = ks * NM1 + (tt * MT + i) * NT + j;
#pragma unroll
                        for (int g = 0; g < NJ; ++g)
                            if (idx == (g * NMT / NJ < NMT ? g * NMT / NJ : NMT - 1)) {
                                if (g < NCS) load_a(g);
                                else if (g < NCS + NBC) store_b(g - NCS, braw[GP ^ 1][g - NCS], bnxt);
                                else if (g < NCS + 2 * NBC) load_b(g - NCS - NBC, braw[GP][g - NCS - NBC]);
                                else convert_chunk(g - NCS - 2 * NBC, anxt);
                            }
                    }
        }
        lds_barrier();
    };
    for (int g = 0; g < ngroups; g += 2) {
        gstep(std::integral_constant<int, 0>{});
        if (g + 1 < ngroups) gstep(std::integral_constant<int, 1>{});
    }

    if (!igemm_epilogue_rows<BM, BN, WM, WN, 2 * A_ST + 2 * B_ST>(d, acc, s_row, smem, n0, tid))
        igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, smem, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm3s2_kernel(const IgemmDesc d) {
    igemm3s2_body<BM, BN, WM, WN>(d);
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg3s2(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk);
    hipLaunchKernelGGL((igemm3s2_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int igemm3s2_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s) {
    switch (tile) {
        case TILE_B3S2_256x64: return launch_cfg3s2<256, 64, 64, 64>(d, s);
        case TILE_B3S2_128x64: return launch_cfg3s2<128, 64, 64, 32>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "igemm3s2: bad tile id %d", (int)tile);
    }
}

}  // namespace sagen
