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
    constexpr int EPI_BYTES = BM * BN * 4 + BM * 4 + 2 * (256 / (BN / 4)) * BN * 4;
    constexpr int SMEM_BYTES = STAGES * ST_BYTES + 256 > EPI_BYTES ? STAGES * ST_BYTES + 256 : EPI_BYTES;   // +256: reads of the dropped rows
    // ONE shared object: a second __shared__ array makes hipcc drain vmcnt before the ds_reads of every step
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int gm = gridDim.x;
#ifdef SAGEN_TRACE      // dev builds: s_memtime stamps of every workgroup's wave 0 -> d.trace[block][16]
    unsigned long long* trc = (d.trace && wave == 0 && blockIdx.y == 0) ? (unsigned long long*)d.trace + (size_t)blockIdx.x * 16 : nullptr;
#define TRC(k) do { if (trc && lane == 0) trc[k] = __builtin_readcyclecounter(); } while (0)
#else
#define TRC(k) do { } while (0)
#endif
    TRC(0);
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
    // ---- K range of this split: groups (dh, chunk) ----
    const int G = 3 * nchunk;
    const int gper = (G + d.splitk - 1) / d.splitk;
    const int g0 = z * gper;
    const int g1 = min(G, g0 + gper);
    const int ngroups = max(g1 - g0, 0);

    // the filter tiles of the first group go out NOW: they only need n0 / lane, and fly while the activation lanes are set up
    {
        const int dh0 = g0 / nchunk, ch0 = g0 - dh0 * nchunk;
        const unsigned bs0 = (unsigned)((dh0 * 3) * nchunk + ch0) * (unsigned)(d.N * 96);
        if (ngroups > 0) {
#pragma unroll
            for (int j = 0; j < B_PW; ++j) {
                const int inst = wave + 4 * j;
#ifndef P3_ABLATE_DMA
                if (4 * (j + 1) <= B_INST || inst < B_INST) dma16(w_rsrc, (float*)(smem + A_BYTES + inst * 1024), b_voff[j], bs0 + (unsigned)b_tapoff[j]);
#endif
            }
        }
    }
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
            const unsigned row = __umulhi((unsigned)p, d.p3_magic_wp);          // p / Wp  (exact: p * Wp < 2^32, conv3p_dispatch)
            const int h = (int)(row - __umulhi(row, d.p3_magic_h) * (unsigned)H);
            bad = (h == 0 ? 1u : 0u) | (h == H - 1 ? 4u : 0u);
        }
        const int base = p * 96 + pl * 32 + 16 * (half ^ ((slot >> 3) & 1));
        a_v0[j] = (bad & 1u) ? OOB : (unsigned)(base - Wp * 96);
        a_v1[j] = (bad & 2u) ? OOB : (unsigned)base;
        a_v2[j] = (bad & 4u) ? OOB : (unsigned)(base + Wp * 96);
        a_cur[j] = OOB;
    }
    // number of DMA instructions this wave issues per group (wave-uniform; differs by at most 2 between waves)
    int my_cnt = 0;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) my_cnt += (wave + 4 * j < A_INST) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < B_PW; ++j) my_cnt += (wave + 4 * j < B_INST) ? 1 : 0;
    my_cnt = __builtin_amdgcn_readfirstlane(my_cnt);
    constexpr int CNT_MAX = A_PW + B_PW;

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

    TRC(1);
    // ---- pipeline fill ----
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < ngroups) {
            begin_issue(t);
#pragma unroll
            for (int s = 0; s < CNT_MAX; ++s)
                if (t > 0 || s < A_PW) issue_one(s);          // (the first group's filter tiles were issued at kernel entry)
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
        // fragments double-buffered over the three horizontal taps: the reads of tap t+1 are issued before the MFMAs of tap t,
        // so only the first read burst after the barrier is exposed
        bf16x8 aq[2][3][MT], bq[2][3][NT];
        auto load_frags = [&](int buf, int dwi) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MT; ++i) aq[buf][pl][i] = *reinterpret_cast<const bf16x8*>(st + a_foff[dwi][i] + pl * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bq[buf][pl][j] = *reinterpret_cast<const bf16x8*>(st + b_foff + (dwi * 3 + pl) * (BN * 32) + j * 32 * 32);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int cb = dwi & 1;
#ifndef P3_NO_SCHED
            __builtin_amdgcn_sched_barrier(0);            // one scheduling region per tap (hipcc otherwise sinks the prefetch reads to their use)
#endif
            if (dwi < 2) load_frags(cb ^ 1, dwi + 1);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#ifndef P3_ABLATE_MFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[cb][TA[tt]][i], bq[cb][TB[tt]][j], acc[i][j], 0, 0, 0);
#else
                        asm volatile("" ::"v"(aq[cb][TA[tt]][i]), "v"(bq[cb][TB[tt]][j]));
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
#ifndef P3_NO_SCHED
            // issue order of the region: one fragment read of the next tap behind each of the first MFMAs, DMAs behind their MFMA
#pragma unroll
            for (int k = 0; k < NM1; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
                if (dwi < 2 && k < 3 * (MT + NT)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
#pragma unroll
                for (int s = 0; s < CNT_MAX; ++s)
                    if (MORE && dwi * NM1 + k == (s + 1) * NMG / (CNT_MAX + 1) - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    };
    {
        const int nmore = max(ngroups - (STAGES - 1), 0);       // groups during which a later group is issued
        int it = 0;
        TRC(2);
        for (; it < nmore; ++it) { group(std::true_type{}, STAGES == 3); if (it == 0) TRC(3); }
        for (; it < ngroups; ++it) group(std::false_type{}, STAGES == 3 && it + 1 < ngroups);
    }
    TRC(4);
    wait_vmcnt_n<0>();
    __syncthreads();

    // ---- epilogue: the tile goes through LDS so that every output row leaves as 16-byte coalesced stores ----
    // (the element-wise MFMA-layout epilogue of igemm_common.h cost 15k cycles per 128x64 tile here - a third of the
    //  workgroup's life at K = 576; this one ~2k.)  Pad pixels, the two overlap rows and rows >= NP are dropped.
    constexpr int TPR = BN / 4;                    // threads per output row (one float4 each)
    constexpr int RPP = 256 / TPR;                 // rows per pass
    float* tile = reinterpret_cast<float*>(smem);                              // [BM][BN]
    int* s_dense = reinterpret_cast<int*>(smem + BM * BN * 4);                 // [BM] dense pixel index or -1
    float* red = reinterpret_cast<float*>(smem + BM * BN * 4 + BM * 4);        // [2][RPP][BN]
    static_assert(BM * BN * 4 + BM * 4 + 2 * RPP * BN * 4 <= SMEM_BYTES, "epilogue staging must fit the tile ring");
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)          // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
                tile[(wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * BN + wn * WN + j * 32 + li] = acc[i][j][e];
    for (int r = tid; r < BM; r += 256) {
        const int p = m0 + r;
        int dense = -1;
        if (r < BME && p < NP) {
            const int row = (int)__umulhi((unsigned)p, d.p3_magic_wp);   // p / Wp = b*H + h: one pad pixel per preceding row
            if (p - row * Wp < W) dense = p - row;
        }
        s_dense[r] = dense;
    }
    __syncthreads();
    TRC(5);
    {
        const int c4 = tid % TPR, rg = tid / TPR;
        const int n = n0 + 4 * c4;
        const bool full = n + 3 < d.N, vec_ok = full && (d.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.y) & 15) == 0);
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
            bias.x = n < d.N ? d.bias[n] : 0.f; bias.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
            bias.z = n + 2 < d.N ? d.bias[n + 2] : 0.f; bias.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NPASS = BM / RPP;
        // all LDS reads first (independent), then the stores: one latency, not NPASS of them
        int dn[NPASS];
        float4 tv[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) dn[k] = s_dense[rg + k * RPP];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) tv[k] = *reinterpret_cast<const float4*>(tile + (rg + k * RPP) * BN + 4 * c4);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            if (dn[k] < 0) continue;
            float4 v = tv[k];
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
            if (d.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            float* dst = d.y + (long)dn[k] * d.ldy + n;
            if (vec_ok) *reinterpret_cast<float4*>(dst) = v;
            else {
                if (n < d.N) dst[0] = v.x;
                if (n + 1 < d.N) dst[1] = v.y;
                if (n + 2 < d.N) dst[2] = v.z;
                if (n + 3 < d.N) dst[3] = v.w;
            }
        }
        if (d.stats != nullptr) {                 // per-channel (sum, sumsq) of the raw output -> fp64 accumulators [2][N]
            *reinterpret_cast<float4*>(red + (0 * RPP + rg) * BN + 4 * c4) = cs;
            *reinterpret_cast<float4*>(red + (1 * RPP + rg) * BN + 4 * c4) = cq;
            __syncthreads();
            for (int t = tid; t < 2 * BN; t += 256) {
                const int which = t / BN, col = t - which * BN;
                if (n0 + col < d.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < RPP; ++g) sum += red[(which * RPP + g) * BN + col];
                    atomicAdd(&d.stats[(long)which * d.N + n0 + col], (double)sum);
                }
            }
        }
    }
    TRC(6);
#undef TRC
}

template <int BM, int BN, int WM, int WN, int STAGES>
static int launch_conv3p(const IgemmDesc& d, hipStream_t s) {
    dim3 grid(cdiv(d.p3_np, BM - 2), cdiv(d.N, BN), d.splitk);
    hipLaunchKernelGGL((conv3p_kernel<BM, BN, WM, WN, STAGES>), grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int conv3p_dispatch(const IgemmDesc& d_in, IgemmTile tile, hipStream_t s) {
    IgemmDesc d = d_in;
    if (!d.xp3 || d.p3_np <= 0) return fail(SAGEN_ERR_NULL, "conv3p: the P3 activation planes are missing");
    if (d.splitk != 1) return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: no split-K");
    if ((long)(d.p3_np + 512) * (d.Win + 1) >= (1L << 32)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: too many pixels for 32-bit index arithmetic");
    d.p3_magic_wp = (unsigned)((1UL << 32) / (unsigned)(d.Win + 1)) + 1u;
    d.p3_magic_h = (unsigned)((1UL << 32) / (unsigned)d.Hin) + 1u;
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
