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
// (dh, 16-channel chunk) operand tile of a workgroup is ONE contiguous run of BM*96 bytes in HBM.  The GEMM runs
// over the padded pixel index; rows that are pad pixels are computed and dropped (1/W of the work).
// Top / bottom image edges: a pixel whose row h+dh falls outside the image sets bit 31 of its DMA offset (three
// precomputed offsets per DMA lane) and the buffer range check writes zeros.
//
// K order (dh, chunk, dw): per group the workgroup stages the activation tile once (with one halo pixel either side)
// and the three dw filter tiles; the three horizontal taps read their A fragments at slot offsets 0 / 1 / 2.
// LDS image = the global byte order (slot stride 96 B, plane stride 32 B) with the two 16-byte halves of a 32-byte
// plane row swapped when (slot >> 3) & 1 - applied on the DMA source address and on the fragment read - which makes
// every ds_read_b128 conflict-free for any slot base.
//
// One output tile per workgroup, two workgroups per CU (128x64 tile: 60 KiB of LDS each): while one is in its K loop the
// other is typically in its prologue / epilogue.  A PERSISTENT variant (workgroups pulling tiles from per-XCD ticket
// counters, the next tile's first group issued under the current tile's last group, epilogue through the freed ring stage)
// was built and measured (tools/ubench_p3.py): two workgroups that are BOTH permanently in their K loops take
// 3.4k cycles per group instead of 2.1k, and the layer 116 us instead of 88 - not kept.  Tiles 256x64 and 64x128 (one
// workgroup per CU) were also built: never faster than 128x64 / 64x64 inside the forward (the tuner's isolated timing liked
// 64x128 for the 512-channel layers, 85 us, where it then ran 114 us) - removed.
#include "igemm3_common.h"
#include <cstdlib>

namespace sagen {

template <int N> __device__ __forceinline__ void wait_vmcnt_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int conv3p_wgs_per_cu(int BM, int BN) { return 2 * (2 * (BM * 96 + BN * 288) + 512) <= 160 * 1024 ? 2 : 1; }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, conv3p_wgs_per_cu(BM, BN)) void conv3p_kernel(const IgemmDesc d) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    static_assert(BN % 32 == 0 && BM % 64 == 0, "tile granularity");
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
    constexpr int CNT_MAX = A_PW + B_PW;
    // epilogue staging (inside ONE ring stage): one wave-row of the output tile at a time, the dense-row table, the statistics partials
    constexpr int TPR = BN / 4;                            // threads per output row (one float4 each)
    constexpr int RPP = 256 / TPR;                         // rows per pass
    constexpr int NPASS = WM / RPP;
    static_assert(WM % RPP == 0, "a wave row is a whole number of store passes");
    constexpr int EPI_TILE = WM * BN * 4, EPI_DENSE = BM * 4, EPI_RED = 2 * RPP * BN * 4;
    static_assert(EPI_TILE + EPI_DENSE + EPI_RED <= ST_BYTES, "epilogue staging must fit one ring stage");
    constexpr int SMEM_BYTES = 2 * ST_BYTES + 256;         // +256: fragment reads of the dropped rows
    // ONE shared object: a second __shared__ array makes hipcc drain vmcnt before the ds_reads of every step
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
#ifdef SAGEN_TRACE      // dev builds: s_memtime stamps of every workgroup's wave 0 -> d.trace[block][16]
    unsigned long long* trc = (d.trace && wave == 0) ? (unsigned long long*)d.trace + (size_t)blockIdx.x * 16 : nullptr;
    unsigned long long tmark = 0;
#define TRC(k) do { if (trc && lane == 0) trc[k] = __builtin_readcyclecounter(); } while (0)
#define TRC_MARK() do { if (trc) tmark = __builtin_readcyclecounter(); } while (0)
#define TRC_ACC(k) do { if (trc) { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) trc[k] += now_ - tmark; tmark = now_; } } while (0)
#else
#define TRC(k) do { } while (0)
#define TRC_MARK() do { } while (0)
#define TRC_ACC(k) do { } while (0)
#endif
    TRC(0);
    const int W = d.Win, H = d.Hin, Wp = W + 1, NP = d.p3_np;
    const int nchunk = d.Cin >> 4;
    const int G = 3 * nchunk;                   // groups (dh, chunk) per tile, >= 3
    const int nM = (NP + BME - 1) / BME, nN = (d.N + BN - 1) / BN;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.xp3, 0, d.xp3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // ---- block -> tile: XCD x (observed: block b runs on XCD b % 8 - a speed assumption only) owns M tiles [x*per, (x+1)*per),
    //      so vertically neighbouring tiles, which share input rows, meet in one L2.  Tile t of the run: tm = x*per + t / nN, tn = t % nN ----
    const int xcd = blockIdx.x & 7;
    const int per = (nM + 7) >> 3;
    auto range_tiles = [&](int x) { return max(min((x + 1) * per, nM) - x * per, 0) * nN; };
    // ---- per-tile lane state ----
    int m0 = 0, n0 = 0;
    unsigned a_v0[A_PW], a_v1[A_PW], a_v2[A_PW], a_cur[A_PW];     // per vertical tap dh = -1 / 0 / +1; the current one
    unsigned b_voff[B_PW];
    int b_tapoff[B_PW];                         // wave-uniform: byte offset of this slot's tap inside a (dh, chunk) group
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        b_tapoff[j] = __builtin_amdgcn_readfirstlane((inst / B_IPT) * nchunk * d.N * 96);
    }
    auto setup_tile = [&](int tk) {
        const int x = tk >> 24, t = tk & 0xffffff;
        const int q = t / nN;
        m0 = (x * per + q) * BME;               // first PADDED pixel of the tile
        n0 = (t - q * nN) * BN;
        // activation DMA lanes: unit U of the stage image = (slot, plane, half); slot <-> padded pixel m0 - 1 + slot
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
        }
        // filter DMA lanes: per tap the image is [plane][BN rows][32 B]
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const int inst = wave + 4 * j;
            const int r = inst % B_IPT;
            const int L = r * 64 + lane;
            const int pl = L / (2 * BN), n = (L >> 1) % BN, half = L & 1;
            b_voff[j] = (inst < B_INST && n0 + n < d.N) ? (unsigned)((pl * d.N + n0 + n) * 32 + 16 * (half ^ ((n >> 3) & 1))) : OOB;
        }
    };

    // issue state (SGPRs): the group being issued
    int q_dh = 0, q_ch = 0, cur_dh = -1;
    unsigned i_asoff = 0, i_bsoff = 0;
    char* i_stage = smem;
    auto begin_issue = [&](int stage) {
        i_stage = smem + stage * ST_BYTES;
        if (q_dh != cur_dh) {                   // the vertical tap changes -> per-lane offsets / edge validity of the new filter row
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

    // ---- first tile ----
    const int t_run = blockIdx.x >> 3;
    if (t_run >= range_tiles(xcd)) return;
    const int ticket = (xcd << 24) | t_run;
    setup_tile(ticket);
    int stage = 0;
    q_dh = 0; q_ch = 0; cur_dh = -1;
    begin_issue(0);
#pragma unroll
    for (int s = CNT_MAX - 1; s >= 0; --s) issue_one(s);        // filter tiles first
    TRC(1);
    TRC_MARK();

    // one group: its 18*MT*NT MFMAs with the DMA of the next group (ISSUE) spread between them.
    // The SOURCE ORDER IS THE SCHEDULE: every MFMA slot is fenced with sched_barrier(0).  Left alone, hipcc's scheduler
    // (a) sinks the prefetch reads back to their first use and (b) clusters the MFMAs of one accumulator back to back -
    // a dependent 16-pass MFMA issues every 64 cycles instead of 32 (measured: 3.8k instead of 2.1k cycles per group).
    auto group = [&](auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const char* st = smem + stage * ST_BYTES;
        // fragments double-buffered over the three horizontal taps: one read of tap t+1 behind each of the first MFMAs of tap t
        bf16x8 fq[2][3 * (MT + NT)];             // [buffer][plane * (MT+NT) + (i | MT + j)]
        constexpr int NF = 3 * (MT + NT);
        auto load_frag = [&](int buf, int dwi, int f) {
            const int pl = f / (MT + NT), r = f - pl * (MT + NT);
            if (r < MT) fq[buf][f] = *reinterpret_cast<const bf16x8*>(st + a_foff[dwi][r] + pl * 32);
            else fq[buf][f] = *reinterpret_cast<const bf16x8*>(st + b_foff + (dwi * 3 + pl) * (BN * 32) + (r - MT) * 32 * 32);
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) load_frag(0, 0, f);
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int cb = dwi & 1;
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int k = (tt * MT + i) * NT + j;             // MFMA slot inside the tap
                        const int idx = dwi * NM1 + k;
                        __builtin_amdgcn_sched_barrier(0);
#ifndef P3_ABLATE_MFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[cb][TA[tt] * (MT + NT) + i], fq[cb][TB[tt] * (MT + NT) + MT + j],
                                                                            acc[i][j], 0, 0, 0);
#else
                        asm volatile("" ::"v"(fq[cb][TA[tt] * (MT + NT) + i]), "v"(fq[cb][TB[tt] * (MT + NT) + MT + j]));
#endif
                        // side jobs of this slot: fragment reads of the next tap (spread over the tap's slots), DMA issue
                        if (dwi < 2) {
#pragma unroll
                            for (int f = 0; f < NF; ++f)
                                if (f * NM1 / NF == k) load_frag(cb ^ 1, dwi + 1, f);
                        }
#ifndef P3_DMA_STRIDE        // DMA slot s goes out behind MFMA s*P3_DMA_STRIDE: front-loaded, so the tiles have most of the group to land
#define P3_DMA_STRIDE 1
#endif
#pragma unroll
                        for (int s = 0; s < CNT_MAX; ++s)
                            if (ISSUE && idx == (P3_DMA_STRIDE > 0 ? min(s * P3_DMA_STRIDE, NMG - 1) : (s + 1) * NMG / (CNT_MAX + 1) - 1)) issue_one(CNT_MAX - 1 - s);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage ^= 1;
    };

    const bool ldy_ok = (d.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.y) & 15) == 0);
    {
        const int m0c = m0, n0c = n0;
        // ================= K loop: groups 0 .. G-2 =================
        for (int it = 0; it + 1 < G; ++it) {
            // group `it` must have landed for every wave; the stage the next issue overwrites has been read by every wave
            wait_vmcnt_n<0>();
            lds_barrier();
            begin_issue(stage ^ 1);
            group(std::true_type{});
        }
        // ================= last group =================
        wait_vmcnt_n<0>();
        lds_barrier();
        group(std::false_type{});
        TRC_ACC(2);

        // ================= epilogue through the ring stage the last group did not use =================
        // Every output row leaves as 16-byte coalesced stores (the element-wise MFMA-layout epilogue of igemm_common.h cost
        // 15k cycles per 128x64 tile).  Pad pixels, the two overlap rows and rows >= NP are dropped.
        char* const epi = smem + (stage ^ 1) * ST_BYTES;
        float* const tile = reinterpret_cast<float*>(epi);                             // [WM][BN]
        int* const s_dense = reinterpret_cast<int*>(epi + EPI_TILE);                   // [BM] dense pixel index or -1
        float* const red = reinterpret_cast<float*>(epi + EPI_TILE + EPI_DENSE);       // [2][RPP][BN]
        lds_barrier();                           // every wave is done with the last group's fragments
        for (int r = tid; r < BM; r += 256) {
            const int p = m0c + r;
            int dense = -1;
            if (r < BME && p < NP) {
                const int row = (int)__umulhi((unsigned)p, d.p3_magic_wp);   // p / Wp = b*H + h: one pad pixel per preceding row
                if (p - row * Wp < W) dense = p - row;
            }
            s_dense[r] = dense;
        }
        const int c4 = tid % TPR, rg = tid / TPR;
        const int n = n0c + 4 * c4;
        const bool full = n + 3 < d.N, vec_ok = full && (ldy_ok);
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
            bias.x = n < d.N ? d.bias[n] : 0.f; bias.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
            bias.z = n + 2 < d.N ? d.bias[n + 2] : 0.f; bias.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int part = 0; part < WAVES_M; ++part) {
            if (wm == part) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e)      // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
                            tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * BN + wn * WN + j * 32 + li] = acc[i][j][e];
            }
            lds_barrier();
            // all LDS reads first (independent), then the stores: one latency, not NPASS of them
            int dn[NPASS];
            float4 tv[NPASS];
#pragma unroll
            for (int k = 0; k < NPASS; ++k) dn[k] = s_dense[part * WM + rg + k * RPP];
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
            if (part + 1 < WAVES_M) lds_barrier();        // the staging rows are rewritten by the next wave row
        }
        if (d.stats != nullptr) {                 // per-channel (sum, sumsq) of the raw output -> fp64 accumulators [2][N]
            *reinterpret_cast<float4*>(red + (0 * RPP + rg) * BN + 4 * c4) = cs;
            *reinterpret_cast<float4*>(red + (1 * RPP + rg) * BN + 4 * c4) = cq;
            lds_barrier();
            for (int t = tid; t < 2 * BN; t += 256) {
                const int which = t / BN, col = t - which * BN;
                if (n0c + col < d.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < RPP; ++g) sum += red[(which * RPP + g) * BN + col];
                    atomicAdd(&d.stats[(long)which * d.N + n0c + col], (double)sum);
                }
            }
        }
        TRC_ACC(3);
#ifdef SAGEN_TRACE
        if (trc && lane == 0) trc[4] += 1;
#endif
    }
    TRC(6);
#undef TRC
#undef TRC_MARK
#undef TRC_ACC
}

// ------------------------------------------------------------------------------------------------------------------------
// conv3pp_kernel: two conv3p workgroups fused into ONE 8-wave workgroup whose halves ("teams" of 4 waves; waves w and w + 4 share
// a SIMD) run half a K group apart, held there by the workgroup barrier: while one team issues its 36 MFMAs per wave, the other
// reads the first fragments of its next group and waits.  Why: a lone conv3p workgroup keeps its SIMDs' matrix pipes 57 % busy
// inside the K loop (barrier -> fragment reads -> 36 MFMAs, one wave per SIMD), two INDEPENDENT workgroups on a CU 68 % (they
// drift into the same phase: 3.4k cycles per group each instead of 2.0k), and stages 4-5 launch fewer workgroups than there are
// CUs.  Phase-locked, each SIMD's pipe alternates between its two waves.
//   MODE 0 (pair)   : the teams own two vertically neighbouring 126-row tiles of one N tile (stage 3: 812 tiles);
//   MODE 1 (split K): the teams own the two halves of the K range of ONE tile and team 1's accumulators are added to team 0's
//                     through LDS before the epilogue (stages 4-5: 416 / 216 tiles for 256 CUs).
// Each team keeps conv3p_kernel's two-stage LDS ring (60 KiB) and its K-loop body; per group and team:
//   [read tap-0 fragments of group g] barrier [issue the DMA of group g+1, 36 MFMAs, taps 1-2 fragments, wait for the DMA] barrier
// with team 1 one barrier behind team 0.
// ------------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512, 1) void conv3pp_kernel(const IgemmDesc d) {
    constexpr int BM = 128, BN = 64, WM = 64, WN = 32;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    constexpr int BME = BM - 2;
    constexpr int A_INST = BM * 6 / 64;
    constexpr int B_IPT = 3 * BN / 32;
    constexpr int B_INST = 3 * B_IPT;
    constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;
    constexpr int A_BYTES = A_INST * 1024, B_BYTES = B_INST * 1024;
    constexpr int ST_BYTES = A_BYTES + B_BYTES;
    constexpr int NM1 = 6 * MT * NT;
    constexpr int NMG = 3 * NM1;
    constexpr int CNT_MAX = A_PW + B_PW;
    constexpr int TPR = BN / 4;
    constexpr int RPP = 256 / TPR;
    constexpr int NPASS = WM / RPP;
    constexpr int EPI_TILE = WM * BN * 4, EPI_DENSE = BM * 4, EPI_RED = 2 * RPP * BN * 4;
    static_assert(EPI_TILE + EPI_DENSE + EPI_RED <= ST_BYTES, "epilogue staging must fit one ring stage");
    constexpr int TEAM_BYTES = 2 * ST_BYTES + 256;
    static_assert(MT * NT * 16 * 256 * 4 <= TEAM_BYTES, "the K-split reduction stages team 1's accumulators in its own ring");
    __shared__ __attribute__((aligned(16))) char smem_all[2 * TEAM_BYTES];

    const int tid = threadIdx.x & 255;                        // inside the team
    const int lane = tid & 63;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    char* const smem = smem_all + team * TEAM_BYTES;
#ifdef SAGEN_TRACE      // dev builds: s_memtime deltas of team 0 / wave 0 per K-loop phase -> d.trace[block][16]
    unsigned long long* trc = (d.trace && wave == 0 && team == 0) ? (unsigned long long*)d.trace + (size_t)blockIdx.x * 16 : nullptr;
    unsigned long long tmark = 0;
#define PP_T0(k) do { if (trc && lane == 0) trc[k] = __builtin_readcyclecounter(); } while (0)
#define PP_MARK() do { if (trc) tmark = __builtin_readcyclecounter(); } while (0)
#define PP_ACC(k) do { if (trc) { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) trc[k] += now_ - tmark; tmark = now_; } } while (0)
#else
#define PP_T0(k) do { } while (0)
#define PP_MARK() do { } while (0)
#define PP_ACC(k) do { } while (0)
#endif
    PP_T0(0);

    const int W = d.Win, H = d.Hin, Wp = W + 1, NP = d.p3_np;
    const int nchunk = d.Cin >> 4;
    const int G = 3 * nchunk;
    const int nM = (NP + BME - 1) / BME, nN = (d.N + BN - 1) / BN;
    const int nMw = MODE == 0 ? (nM + 1) >> 1 : nM;           // M tiles (pairs of them) per workgroup column
    // K range of this team (MODE 1: halves; a group index >= G is an empty group: zero operands)
    const int Gt = MODE == 0 ? G : (G + 1) >> 1;
    const int g_first = MODE == 0 ? 0 : team * Gt;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.xp3, 0, d.xp3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    const int xcd = blockIdx.x & 7;
    const int per = (nMw + 7) >> 3;
    const int t_run = blockIdx.x >> 3;
    const int my_tiles = max(min((xcd + 1) * per, nMw) - xcd * per, 0) * nN;
    if (t_run >= my_tiles) return;                            // (whole workgroup)
    const int qq = t_run / nN;
    const int mw = xcd * per + qq;
    const int m0 = (MODE == 0 ? 2 * mw + team : mw) * BME;    // first PADDED pixel of this team's tile (>= NP: an empty tile, all rows dropped)
    const int n0 = (t_run - qq * nN) * BN;

    unsigned a_v0[A_PW], a_v1[A_PW], a_v2[A_PW], a_cur[A_PW];
    unsigned b_voff[B_PW];
    int b_tapoff[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        b_tapoff[j] = __builtin_amdgcn_readfirstlane((inst / B_IPT) * nchunk * d.N * 96);
    }
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int inst = wave + 4 * j;
        const int U = inst * 64 + lane;
        const int slot = U / 6, rem = U - 6 * slot;
        const int pl = rem >> 1, half = rem & 1;
        const int p = m0 - 1 + slot;
        unsigned bad = 7u;
        if (inst < A_INST && p >= 0 && p < NP) {
            const unsigned row = __umulhi((unsigned)p, d.p3_magic_wp);
            const int h = (int)(row - __umulhi(row, d.p3_magic_h) * (unsigned)H);
            bad = (h == 0 ? 1u : 0u) | (h == H - 1 ? 4u : 0u);
        }
        const int base = p * 96 + pl * 32 + 16 * (half ^ ((slot >> 3) & 1));
        a_v0[j] = (bad & 1u) ? OOB : (unsigned)(base - Wp * 96);
        a_v1[j] = (bad & 2u) ? OOB : (unsigned)base;
        a_v2[j] = (bad & 4u) ? OOB : (unsigned)(base + Wp * 96);
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        const int r = inst % B_IPT;
        const int L = r * 64 + lane;
        const int pl = L / (2 * BN), n = (L >> 1) % BN, half = L & 1;
        b_voff[j] = (inst < B_INST && n0 + n < d.N) ? (unsigned)((pl * d.N + n0 + n) * 32 + 16 * (half ^ ((n >> 3) & 1))) : OOB;
    }

    // issue state: the group being issued
    int q_dh = g_first / nchunk, q_ch = g_first - q_dh * nchunk, cur_dh = -1;
    unsigned i_asoff = 0, i_bsoff = 0, i_dead = 0;
    char* i_stage = smem;
    auto begin_issue = [&](int stage, bool past_end = false) {
        i_stage = smem + stage * ST_BYTES;
        // past the K range (odd G in MODE 1; the issue slots of the last group): every lane reads range-check zeros - the K loop
        // stays one branch-free instruction stream
        i_dead = (q_dh >= 3 || past_end) ? OOB : 0u;
        if (q_dh != cur_dh) {
            cur_dh = q_dh;
#pragma unroll
            for (int j = 0; j < A_PW; ++j) a_cur[j] = q_dh == 0 ? a_v0[j] : (q_dh == 1 ? a_v1[j] : a_v2[j]);
        }
        i_asoff = (unsigned)q_ch * d.xp3_cstride;
        i_bsoff = (unsigned)((q_dh * 3) * nchunk + q_ch) * (unsigned)(d.N * 96);
        ++q_ch;
        if (q_ch == nchunk) { q_ch = 0; ++q_dh; }
    };
    auto issue_one = [&](int s) {
        if (s < A_PW) {
            const int inst = wave + 4 * s;
            if (A_INST % 4 == 0 || inst < A_INST) dma16(x_rsrc, (float*)(i_stage + inst * 1024), a_cur[s] | i_dead, i_dead ? 0u : i_asoff);
        } else {
            const int j = s - A_PW;
            const int inst = wave + 4 * j;
            if (4 * (j + 1) <= B_INST || inst < B_INST)
                dma16(w_rsrc, (float*)(i_stage + A_BYTES + inst * 1024), b_voff[j] | i_dead, i_dead ? 0u : i_bsoff + (unsigned)b_tapoff[j]);
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
    int a_foff[3][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int sl = wm * WM + i * 32 + li + dwi;
            a_foff[dwi][i] = sl * 96 + 16 * (kk ^ ((sl >> 3) & 1));
        }
    const int b_foff = A_BYTES + (wn * WN + li) * 32 + 16 * (kk ^ ((li >> 3) & 1));
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};
    constexpr int NF = 3 * (MT + NT);

    // ---- prologue: group 0 of this team ----
    int stage = 0;
    begin_issue(0);
#pragma unroll
    for (int s = CNT_MAX - 1; s >= 0; --s) issue_one(s);
    wait_vmcnt_n<0>();
    lds_barrier();
    if (team == 1) lds_barrier();                              // team 1 runs one barrier (half a group) behind team 0

    // Load phase = everything that is not an MFMA: ALL 27 fragments of the group (108 VGPRs) and the DMA of the next group are
    // issued while the other team computes; the MFMA phase is 36 back-to-back MFMAs.  (With the next tap's fragments and the DMA
    // issue inside the MFMA phase, as in conv3p_kernel, a phase took 1.9k cycles instead of 1.15k: s_memtime stamps, tools/ubench_p3.py.)
    bf16x8 fq[3][NF];
    PP_MARK();
    for (int it = 0; it < Gt; ++it) {
        const char* st = smem + stage * ST_BYTES;
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int pl = f / (MT + NT), r = f - pl * (MT + NT);
#ifdef PP_ABL_READS      // (timing experiments only: wrong results)
                if (it == 0)
#endif
                {
                if (r < MT) fq[dwi][f] = *reinterpret_cast<const bf16x8*>(st + a_foff[dwi][r] + pl * 32);
                else fq[dwi][f] = *reinterpret_cast<const bf16x8*>(st + b_foff + (dwi * 3 + pl) * (BN * 32) + (r - MT) * 32 * 32);
                }
            }
        begin_issue(stage ^ 1, it + 1 >= Gt);                 // the other stage was consumed by the previous load phase
#ifndef PP_ABL_DMA
#pragma unroll
        for (int s = CNT_MAX - 1; s >= 0; --s) issue_one(s);
#endif
        PP_ACC(8);
        lds_barrier();
        PP_ACC(9);
        // ---- MFMA phase ----
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        __builtin_amdgcn_sched_barrier(0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[dwi][TA[tt] * (MT + NT) + i], fq[dwi][TB[tt] * (MT + NT) + MT + j],
                                                                            acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            PP_ACC(10 + dwi);
        }
#ifndef PP_ABL_WAIT
        wait_vmcnt_n<0>();                                     // the next group's tiles (issued in the load phase) have landed
#endif
        PP_ACC(13);
        stage ^= 1;
        if (!(team == 1 && it == Gt - 1)) lds_barrier();
        PP_ACC(14);
    }
#ifdef SAGEN_TRACE
    if (trc && lane == 0) trc[4] = (unsigned long long)Gt;
#endif
    lds_barrier();                                             // both teams are out of their K loops

    // ---- MODE 1: team 1's partial sums -> team 0 ----
    if (MODE == 1) {
        float* const xch = reinterpret_cast<float*>(smem_all + TEAM_BYTES);       // team 1's ring
        if (team == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[((i * NT + j) * 16 + e) * 256 + tid] = acc[i][j][e];
        }
        lds_barrier();
        if (team == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((i * NT + j) * 16 + e) * 256 + tid];
        }
        lds_barrier();                                         // (team 1's epilogue staging reuses the exchange area)
    }
    const bool active = MODE == 0 || team == 0;                // (an inactive team still walks through the epilogue's barriers)

    // ---- epilogue (conv3p_kernel's, per team, in the team's own ring) ----
    const bool ldy_ok = (d.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.y) & 15) == 0);
    char* const epi = smem;
    float* const tile = reinterpret_cast<float*>(epi);
    int* const s_dense = reinterpret_cast<int*>(epi + EPI_TILE);
    float* const red = reinterpret_cast<float*>(epi + EPI_TILE + EPI_DENSE);
    for (int r = tid; r < BM; r += 256) {
        const int p = m0 + r;
        int dense = -1;
        if (active && r < BME && p < NP) {
            const int row = (int)__umulhi((unsigned)p, d.p3_magic_wp);
            if (p - row * Wp < W) dense = p - row;
        }
        s_dense[r] = dense;
    }
    const int c4 = tid % TPR, rg = tid / TPR;
    const int n = n0 + 4 * c4;
    const bool full = n + 3 < d.N, vec_ok = full && (ldy_ok);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias) {
        bias.x = n < d.N ? d.bias[n] : 0.f; bias.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
        bias.z = n + 2 < d.N ? d.bias[n + 2] : 0.f; bias.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
    }
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int part = 0; part < WAVES_M; ++part) {
        if (wm == part) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * BN + wn * WN + j * 32 + li] = acc[i][j][e];
        }
        lds_barrier();
        int dn[NPASS];
        float4 tv[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) dn[k] = s_dense[part * WM + rg + k * RPP];
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
        if (part + 1 < WAVES_M) lds_barrier();
    }
    if (d.stats != nullptr) {
        *reinterpret_cast<float4*>(red + (0 * RPP + rg) * BN + 4 * c4) = cs;
        *reinterpret_cast<float4*>(red + (1 * RPP + rg) * BN + 4 * c4) = cq;
        lds_barrier();
        if (active) {
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
    PP_T0(6);
#undef PP_T0
#undef PP_MARK
#undef PP_ACC
}

template <int MODE>
static int launch_conv3pp(const IgemmDesc& d, hipStream_t s) {
    const int nM = cdiv(d.p3_np, 126), nMw = MODE == 0 ? (nM + 1) / 2 : nM;
    const int per = (nMw + 7) / 8;
    const int grid = 8 * per * cdiv(d.N, 64);
    hipLaunchKernelGGL((conv3pp_kernel<MODE>), dim3(grid), dim3(512), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv3p(const IgemmDesc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.p3_np, BM - 2) * cdiv(d.N, BN);
    const int per = (cdiv(d.p3_np, BM - 2) + 7) / 8;                  // M tiles per XCD run; blocks past the end of a run exit at once
    const int grid = 8 * per * cdiv(d.N, BN);
    (void)tiles;
    hipLaunchKernelGGL((conv3p_kernel<BM, BN, WM, WN>), dim3(grid), dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int conv3p_dispatch(const IgemmDesc& d_in, IgemmTile tile, hipStream_t s) {
    IgemmDesc d = d_in;
    if (!d.xp3 || d.p3_np <= 0) return fail(SAGEN_ERR_NULL, "conv3p: the P3 activation planes are missing");
    if (d.splitk != 1) return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: no split-K");
    // the two mul-hi divisions (pixel / (Win+1), then row / Hin) are exact while dividend * divisor < 2^32; plane offsets are 32-bit
    if ((long)(d.p3_np + 512) * (d.Win + 1) >= (1L << 32) || ((long)(d.p3_np + 512) / (d.Win + 1) + 1) * d.Hin >= (1L << 32))
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: too many pixels for 32-bit index arithmetic");
    if ((long)d.p3_np * 96 >= (1L << 31) || (long)d.xp3_cstride * (d.Cin / 16) >= (1L << 31) || d.xp3_bytes == 0)
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: the activation planes exceed 2 GiB buffer addressing (use a smaller batch)");
    if (cdiv(d.p3_np, 62) * (long)cdiv(d.N, 64) >= (1L << 24)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: too many tiles");
    d.p3_magic_wp = (unsigned)((1UL << 32) / (unsigned)(d.Win + 1)) + 1u;
    d.p3_magic_h = (unsigned)((1UL << 32) / (unsigned)d.Hin) + 1u;
    switch (tile) {
        case TILE_P3_128x64: return launch_conv3p<128, 64, 64, 32>(d, s);
        case TILE_P3_128x128: return launch_conv3p<128, 128, 64, 64>(d, s);
        case TILE_P3_64x64: return launch_conv3p<64, 64, 32, 32>(d, s);
        case TILE_P3PP_PAIR: return launch_conv3pp<0>(d, s);
        case TILE_P3PP_SPLITK: return launch_conv3pp<1>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "conv3p: bad tile id %d", (int)tile);
    }
}

}  // namespace sagen
