// conv3p_kernel: dense 3x3 stride-1 SAME convolution (reference op: tf.nn.convolution, core.py:206, as used by the
// ResNet18 trunk, resnet.py:141-190 / 215-235) whose activation operand arrives ALREADY SPLIT into its three bf16
// planes ("P3" tensors, p3.hip).  Arithmetic is the bf16x3 scheme of igemm3.hip (six bf16 MFMA products per fp32
// product, fp32 accumulate); what changes is who does the split:
//
//   igemm3dw_kernel   loads fp32 activations, applies the producer's BN+ReLU and splits them in its K loop - once per
//                     workgroup and per filter row, i.e. 3x redundantly, ~6 VALU per MFMA: issue-bound at 35 % matrix-pipe
//                     utilisation (profiles/r01_pmc_per_launch.json);
//   conv3p_kernel     the elementwise pass that has to touch the tensor anyway (BN+ReLU of conv_1, the residual merge,
//                     the max-pool) writes the planes once; the K loop here is LDS-DMA -> ds_read_b128 -> MFMA and
//                     nothing else: no VGPR staging, no conversion VALU, no ds_write.
//
// P3 layout (bf16): [Cin/16][NP][3 planes][16 channels], NP = B*H*(W+1): every image row carries ONE trailing zero
// pixel.  With it the input pixel of (padded) output pixel p under tap (dh, dw) is simply p + dh*(W+1) + dw: the pad
// pixel is the right-hand padding of its own row and the left-hand padding of the next one, so a tile needs no edge
// masks and its LDS slots are contiguous (the igemm3dw gap slots broke the bank swizzle: 24-27 % conflict cycles).  The
// (dh, 16-channel chunk) operand tile of a workgroup is ONE contiguous run of (BM+2)*96 bytes in HBM.  The GEMM runs
// over the padded pixel index; rows that are pad pixels are computed and dropped (1/W of the work).
// Top / bottom image edges: a pixel whose row h+dh falls outside the image sets bit 31 of its DMA offset (three
// precomputed offsets per DMA lane) and the buffer range check writes zeros.
//
// K order (dh, chunk, dw): per group the workgroup stages the activation tile once (with one halo pixel either side)
// and the three dw filter tiles; the three horizontal taps read their A fragments at slot offsets 0 / 1 / 2.
// LDS image = the global byte order (slot stride 96 B, plane stride 32 B) with the two 16-byte halves of a 32-byte
// plane row swapped when (slot >> 3) & 1 - applied on the DMA source address and on the fragment read - which makes
// every ds_read_b128 conflict-free for any slot base.  Ring of STAGES stages, one barrier per group, counted vmcnt.
#include "igemm3_common.h"

namespace sagen {

template <int N> __device__ __forceinline__ void wait_vmcnt_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256, (STAGES == 2 && BM * 96 + BN * 288 <= 36 * 1024) ? 2 : 1) void conv3p_kernel(const IgemmDesc d) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    static_assert(BN % 32 == 0 && BM % 32 == 0, "tile granularity");
    // The MFMA tile has BM rows, the workgroup OWNS the first BME = BM - 2 of them: the activation image is then exactly
    // BM slots (BME outputs + one halo pixel either side) = BM*96 bytes = a whole number of 1 KiB DMA instructions, evenly
    // divisible among the waves for BM = 128 / 256.  Rows BME, BME+1 read past the image and are dropped (1.6 % of the MFMAs).
    constexpr int BME = BM - 2;
    constexpr int A_INST = BM * 6 / 64;                    // LDS-DMA wave-instructions (1 KiB each) of one activation stage
    static_assert(BM * 6 % 64 == 0, "activation image must be whole DMA instructions");
    constexpr int B_IPT = 3 * BN / 32;                     // per tap: 3 planes x BN rows x 32 B
    constexpr int B_INST = 3 * B_IPT;
    constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;   // slots per wave
    constexpr int A_BYTES = A_INST * 1024, B_BYTES = B_INST * 1024;
    constexpr int ST_BYTES = A_BYTES + B_BYTES;
    constexpr int NM1 = 6 * MT * NT;                       // MFMAs per tap
    constexpr int NMG = 3 * NM1;                           // MFMAs per group
    constexpr int EPI_BYTES = BM * (int)sizeof(RowInfo) + 2 * WAVES_M * BN * 4;
    constexpr int SMEM_BYTES = STAGES * ST_BYTES + 256 > EPI_BYTES ? STAGES * ST_BYTES + 256 : EPI_BYTES;   // +256: reads of the dropped rows
    // ONE shared object: a second __shared__ array makes hipcc drain vmcnt before the ds_reads of every step
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int gm = gridDim.x;
    int tile_m;
    {   // XCD-aware remap: each XCD (own L2) owns a contiguous run of M tiles (neighbours share halo rows and the filter)
        const int bid = blockIdx.x;
        const int q = gm >> 3, r = gm & 7, xcd = bid & 7, j = bid >> 3;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = tile_m * BME;                // first PADDED pixel of the tile
    const int n0 = blockIdx.y * BN;
    const int z = blockIdx.z;
    const int W = d.Win, H = d.Hin, Wp = W + 1, NP = d.p3_np;
    const int nchunk = d.Cin >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.xp3, 0, d.xp3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // ---- activation DMA lanes: unit U of the stage image = (slot, plane, half); slot <-> padded pixel m0 - 1 + slot ----
    unsigned a_v0[A_PW], a_v1[A_PW], a_v2[A_PW], a_cur[A_PW];     // per vertical tap dh = -1 / 0 / +1; the current one
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int inst = wave + 4 * j;
        const int U = inst * 64 + lane;
        const int slot = U / 6, rem = U - 6 * slot;
        const int pl = rem >> 1, half = rem & 1;
        const int p = m0 - 1 + slot;
        unsigned bad = 7u;
        if (inst < A_INST && p >= 0 && p < NP) {
            const int h = (p / Wp) % H;
            bad = (h == 0 ? 1u : 0u) | (h == H - 1 ? 4u : 0u);
        }
        const int base = p * 96 + pl * 32 + 16 * (half ^ ((slot >> 3) & 1));
        a_v0[j] = (bad & 1u) ? OOB : (unsigned)(base - Wp * 96);
        a_v1[j] = (bad & 2u) ? OOB : (unsigned)base;
        a_v2[j] = (bad & 4u) ? OOB : (unsigned)(base + Wp * 96);
        a_cur[j] = OOB;
    }
    // ---- filter DMA lanes: per tap the image is [plane][BN rows][32 B] ----
    unsigned b_voff[B_PW];
    int b_tapoff[B_PW];                         // wave-uniform: byte offset of this slot's tap inside a (dh, chunk) group
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        const int tap = inst / B_IPT, r = inst - tap * B_IPT;
        const int L = r * 64 + lane;
        const int pl = L / (2 * BN), n = (L >> 1) % BN, half = L & 1;
        b_voff[j] = (inst < B_INST && n0 + n < d.N) ? (unsigned)((pl * d.N + n0 + n) * 32 + 16 * (half ^ ((n >> 3) & 1))) : OOB;
        b_tapoff[j] = __builtin_amdgcn_readfirstlane(tap * nchunk * d.N * 96);
    }
    // number of DMA instructions this wave issues per group (wave-uniform; differs by at most 2 between waves)
    int my_cnt = 0;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) my_cnt += (wave + 4 * j < A_INST) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < B_PW; ++j) my_cnt += (wave + 4 * j < B_INST) ? 1 : 0;
    my_cnt = __builtin_amdgcn_readfirstlane(my_cnt);
    constexpr int CNT_MAX = A_PW + B_PW;

    // ---- K range of this split: groups (dh, chunk) ----
    const int G = 3 * nchunk;
    const int gper = (G + d.splitk - 1) / d.splitk;
    const int g0 = z * gper;
    const int g1 = min(G, g0 + gper);
    const int ngroups = max(g1 - g0, 0);

    // issue state (SGPRs): the group being issued
    int q_dh = g0 / nchunk, q_ch = g0 - q_dh * nchunk, cur_dh = -1;
    unsigned i_asoff = 0, i_bsoff = 0;
    char* i_stage = smem;
    auto begin_issue = [&](int stage) {
        i_stage = smem + stage * ST_BYTES;
        if (q_dh != cur_dh) {                   // twice per kernel: the vertical tap changes -> per-lane offsets / edge validity of the new filter row
            cur_dh = q_dh;
#pragma unroll
            for (int j = 0; j < A_PW; ++j) a_cur[j] = q_dh == 0 ? a_v0[j] : (q_dh == 1 ? a_v1[j] : a_v2[j]);
        }
        i_asoff = (unsigned)q_ch * d.xp3_cstride;
        i_bsoff = (unsigned)((q_dh * 3) * nchunk + q_ch) * (unsigned)(d.N * 96);
        ++q_ch;
        if (q_ch == nchunk) { q_ch = 0; ++q_dh; }
    };
    auto issue_one = [&](int s) {               // s = compile-time slot index: A slots first, then B slots
        if (s < A_PW) {
            const int inst = wave + 4 * s;
#ifndef P3_ABLATE_DMA
            if (A_INST % 4 == 0 || inst < A_INST) dma16(x_rsrc, (float*)(i_stage + inst * 1024), a_cur[s], i_asoff);
#endif
        } else {
            const int j = s - A_PW;
            const int inst = wave + 4 * j;
#ifndef P3_ABLATE_DMA
            if (4 * (j + 1) <= B_INST || inst < B_INST)
                dma16(w_rsrc, (float*)(i_stage + A_BYTES + inst * 1024), b_voff[j], i_bsoff + (unsigned)b_tapoff[j]);
#endif
        }
    };
    // wait until at most one group (keep_one) / nothing of this wave's DMAs is still in flight
    auto wait_keep = [&](bool keep_one) {
        if (!keep_one) { wait_vmcnt_n<0>(); return; }
        if (my_cnt == CNT_MAX) wait_vmcnt_n<CNT_MAX>();
        else if (my_cnt == CNT_MAX - 1) wait_vmcnt_n<(CNT_MAX > 1 ? CNT_MAX - 1 : 0)>();
        else wait_vmcnt_n<(CNT_MAX > 2 ? CNT_MAX - 2 : 0)>();
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing (bytes inside a stage) ----
    const int li = lane & 31, kk = lane >> 5;
    int a_foff[3][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int sl = wm * WM + i * 32 + li + dwi;              // output row r sits at slot r + 1; tap dw reads slot r + dw
            a_foff[dwi][i] = sl * 96 + 16 * (kk ^ ((sl >> 3) & 1));
        }
    const int b_foff = A_BYTES + (wn * WN + li) * 32 + 16 * (kk ^ ((li >> 3) & 1));

    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm

    // ---- pipeline fill ----
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < ngroups) {
            begin_issue(t);
#pragma unroll
            for (int s = 0; s < CNT_MAX; ++s) issue_one(s);
        }

    int stage = 0;
    // one group: wait for its tiles, then its 18*MT*NT MFMAs with the DMA of a later group (MORE) spread between them
    auto group = [&](auto more_tag, bool keep_one) {
        constexpr bool MORE = decltype(more_tag)::value;
        // group `it` must have landed for every wave; the stage the next issue overwrites has been read by every wave
        wait_keep(keep_one);
        lds_barrier();
        int istage = stage + (STAGES - 1);
        if (istage >= STAGES) istage -= STAGES;
        if (MORE) begin_issue(istage);
        const char* st = smem + stage * ST_BYTES;
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            bf16x8 aq[3][MT], bq[3][NT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MT; ++i) aq[pl][i] = *reinterpret_cast<const bf16x8*>(st + a_foff[dwi][i] + pl * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bq[pl][j] = *reinterpret_cast<const bf16x8*>(st + b_foff + (dwi * 3 + pl) * (BN * 32) + j * 32 * 32);
            }
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#ifndef P3_ABLATE_MFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[TA[tt]][i], bq[TB[tt]][j], acc[i][j], 0, 0, 0);
#else
                        asm volatile("" ::"v"(aq[TA[tt]][i]), "v"(bq[TB[tt]][j]));
#endif
                        const int idx = dwi * NM1 + (tt * MT + i) * NT + j;
                        // DMA slot s goes out after MFMA (s+1)*NMG/(CNT_MAX+1) - 1: spread over the group's MFMAs
#ifdef P3_DMA_EARLY
#pragma unroll
                        for (int s = 0; s < CNT_MAX; ++s)
                            if (MORE && idx == s) issue_one(s);
#else
#pragma unroll
                        for (int s = 0; s < CNT_MAX; ++s)
                            if (MORE && idx == (s + 1) * NMG / (CNT_MAX + 1) - 1) issue_one(s);
#endif
                    }
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    };
    {
        const int nmore = max(ngroups - (STAGES - 1), 0);       // groups during which a later group is issued
        int it = 0;
        for (; it < nmore; ++it) group(std::true_type{}, STAGES == 3);
        for (; it < ngroups; ++it) group(std::false_type{}, STAGES == 3 && it + 1 < ngroups);
    }
    wait_vmcnt_n<0>();
    __syncthreads();

    // ---- epilogue: per-row output geometry (pad pixels, the two overlap rows and rows >= NP are dropped) ----
    RowInfo* s_row = reinterpret_cast<RowInfo*>(smem);
    float* red = reinterpret_cast<float*>(smem + BM * sizeof(RowInfo));
    for (int r = tid; r < BM; r += 256) {
        const int p = m0 + r;
        RowInfo ri;
        ri.boff = 0; ri.nmlo = 0; ri.nmhi = 0; ri.hrem = 0; ri.wrem = 0; ri.pad = 0; ri.rowoff = 0;
        if (r < BME && p < NP) {
            const int row = p / Wp, w = p - row * Wp;           // row = b*H + h
            if (w < W) {
                ri.hrem = 1; ri.wrem = 1;
                ri.rowoff = ((long)row * W + w) * d.ldy;
            }
        }
        s_row[r] = ri;
    }
    __syncthreads();
    igemm_epilogue<BM, BN, WM, WN>(d, acc, s_row, red, m0, n0, z, tid);
}

template <int BM, int BN, int WM, int WN, int STAGES>
static int launch_conv3p(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.p3_np, BM - 2), cdiv(d.N, BN), d.splitk);
    hipLaunchKernelGGL((conv3p_kernel<BM, BN, WM, WN, STAGES>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int conv3p_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s) {
    if (!d.xp3 || d.p3_np <= 0) return fail(SAGEN_ERR_NULL, "conv3p: the P3 activation planes are missing");
    if (d.splitk != 1) return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: no split-K");
    switch (tile) {
        case TILE_P3_128x64: return launch_conv3p<128, 64, 64, 32, 2>(d, s);
        case TILE_P3_128x128: return launch_conv3p<128, 128, 64, 64, 2>(d, s);
        case TILE_P3_128x128_S3: return launch_conv3p<128, 128, 64, 64, 3>(d, s);
        case TILE_P3_256x64_S3: return launch_conv3p<256, 64, 64, 64, 3>(d, s);
        case TILE_P3_64x64: return launch_conv3p<64, 64, 32, 32, 2>(d, s);
        case TILE_P3_64x128: return launch_conv3p<64, 128, 32, 64, 2>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: bad tile id %d", (int)tile);
    }
}

}  // namespace sagen
