// conv3g_kernel: strided / arbitrary-tap convolutions over PRE-SPLIT activation planes (reference op: tf.nn.convolution, core.py:206,
// as the first block of ResNet stages 3-5 uses it: the 3x3 stride-2 conv_1 and the 1x1 stride-2 shortcut, resnet.py:200-236).
// conv3p_kernel's arithmetic and K loop - LDS-DMA -> ds_read_b128 -> MFMA, no operand split, no ds_write - without its
// restriction to dense stride-1 3x3 geometry:
//
//   conv3p_kernel   the (dh, chunk) operand tile of a workgroup is ONE contiguous run of the plane tensor and the three horizontal
//                   taps read it at slot offsets 0 / 1 / 2;
//   conv3g_kernel   every K tile = (tap, 16-channel chunk) is GATHERED: LDS-DMA takes a per-lane global address, so the lane that
//                   fills (slot r, plane, half) fetches the 16 bytes of input pixel origin(r) + tap displacement - any stride, any
//                   tap set whose horizontal reach stays within [-1, Win] (the pad pixel closing every plane row IS the SAME padding
//                   left and right); rows above / below the image set bit 31 of the offset and the range check writes zeros.
//
// Before: these layers ran on igemm3_kernel, which loads fp32 activations, applies the producer's BN + ReLU and splits them into
// the three bf16 planes INSIDE the K loop (19 non-MFMA instructions per MFMA, 27 % matrix-pipe busy: profiles/r03_pmc_per_launch.json)
// - although the block merge that produced their input had already written exactly those planes for the stride-1 convs of its stage.
//
// P3 layout (p3.hip): [Cin/16][NP][3 planes][16 ch] bf16, NP = B*Hin*(Win+1).  Filter planes [K/16][3][N][16], k = (tap, channel).
// One output tile per workgroup; M = dense output pixels (b, ho, wo); KS K tiles per barrier step, two-stage ring.
#include "igemm3_common.h"

namespace sagen {

typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));

// H2: the planes are two fp16 planes per operand (conv3h.hip: 64 B per pixel and chunk, three products per multiply, the tile scaled
// back by 2^-(ka + kw) in the epilogue); otherwise three bf16 planes (96 B, six products)
constexpr int conv3g_wgs_per_cu(int BM, int BN, int KS, bool H2) {
    const int lds = 2 * KS * (BM + BN) * (H2 ? 64 : 96) + 512;
    return 3 * lds <= 160 * 1024 ? 3 : (2 * lds <= 160 * 1024 ? 2 : 1);
}

// EPI 0: dense NHWC rows (+ bias / ReLU / batch-norm statistics); EPI 1: the fused decoder tail of deconv1 (igemm_epilogue_maskmix:
// sigmoid + track-weighted sums per mask bin, IgemmDesc::mm_*) on the depth-to-space tile
template <int BM, int BN, int WM, int WN, int KS, bool H2, int EPI = 0>
__global__ __launch_bounds__(256, conv3g_wgs_per_cu(BM, BN, KS, H2)) void conv3g_kernel(const IgemmDesc d_in) {
    IgemmDesc d = d_in;
    if (d.grp.G > 1) igemm_relocate(d, (int)blockIdx.z);              // grouped launch (common.h)
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int NPL = H2 ? 2 : 3, SB = NPL * 32, UPS = NPL * 2;      // planes, bytes and 16-byte units per slot
    constexpr int NPR = H2 ? 3 : 6;                                    // matrix products per fp32 product
    constexpr int A_INST = BM * UPS / 64, B_INST = BN * UPS / 64;      // 1 KiB LDS-DMA wave-instructions per K tile
    static_assert(BM * UPS % 64 == 0 && BN * UPS % 64 == 0, "operand images must be whole DMA instructions");
    constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;    // slots per wave and K tile
    constexpr int A_BYTES = A_INST * 1024, B_BYTES = B_INST * 1024;
    constexpr int T_BYTES = A_BYTES + B_BYTES;                         // one K tile: activation image, then filter image
    constexpr int ST_BYTES = KS * T_BYTES;
    constexpr int SPT = A_PW + B_PW;                                   // DMA slots per wave and K tile
    constexpr int CNT = KS * SPT;                                      // ... and group
    constexpr int NM1 = NPR * MT * NT;                                 // MFMAs per K tile
    constexpr int NMG = KS * NM1;
    static_assert(CNT <= NMG, "one DMA slot per MFMA slot at most");
    constexpr int NF = NPL * (MT + NT);
    constexpr int SMEM_BYTES = 2 * ST_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];     // ONE shared object (a second one makes hipcc drain vmcnt before every ds_read)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Wp = d.Win + 1;
    const int HinP = d.xp3_rows > 0 ? d.xp3_rows : d.Hin;        // image rows held by the planes (from row xp3_row0 on)
    const int nchunk = d.Cin >> 4;
    const int NKT = d.ntaps * nchunk;
    const int G = (NKT + KS - 1) / KS;
    const int nM = (d.M + BM - 1) / BM, nN = (d.N + BN - 1) / BN;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.xp3, 0, d.xp3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = H2 ? __builtin_amdgcn_make_buffer_rsrc((void*)d.wh2, 0, d.wh2_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)(d.w + (size_t)d.N * d.Kpad), 0, d.w_bytes / 2 * 3, 0x00020000);

    // block -> tile: XCD x owns a contiguous run of M tiles (neighbouring tiles share input rows in one L2)
    // (grouped launch: group g rotates the owner by g, so that the XCD a short tile list leaves without tiles - 14 tiles of 256 rows at
    //  stage 5 are 2 + 2 + .. + 0 - is another one for every group; a workgroup's physical XCD stays blockIdx.x & 7: the grid is a
    //  multiple of 8 wide, and all tiles of one owner still run on one XCD)
    const int xcd = (blockIdx.x + blockIdx.z) & 7;
    const int per = (nM + 7) >> 3;
    const int t_run = blockIdx.x >> 3;
    const int my_tiles = max(min((xcd + 1) * per, nM) - xcd * per, 0) * nN;
    if (t_run >= my_tiles) return;
    const int tq = t_run / nN;
    const int m0 = (xcd * per + tq) * BM, n0 = (t_run - tq * nN) * BN;

    // ---- per-lane DMA state ----
    // activation unit U = inst*64 + lane of a tile image = (slot, plane, half); slot <-> output row m0 + slot
    unsigned a_base[A_PW], a_bad[A_PW], a_cur[A_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int inst = wave + 4 * j;
        const int U = inst * 64 + lane;
        const int slot = U / UPS, rem = U - UPS * slot;
        // which 16-byte unit of the pixel lands at position `rem` of the slot (the swizzles of conv3p.hip / conv3h.hip)
        const int unit = H2 ? (rem ^ ((slot >> 2) & 3)) : ((rem & ~1) | ((rem & 1) ^ ((slot >> 3) & 1)));
        const int m = m0 + slot;
        a_base[j] = 0; a_bad[j] = 0xffffffffu;
        if (inst < A_INST && m < d.M) {
            const unsigned row = __umulhi((unsigned)m, d.p3_magic_wp);                 // m / Wg = b*Hg + ho   (exact: conv3g_dispatch)
            const int wo = m - (int)row * d.Wg;
            const unsigned b = __umulhi(row, d.p3_magic_h);                            // row / Hg
            const int ho = (int)row - (int)b * d.Hg;
            // (the output grid may start at (g_h0, g_w0); the planes may hold only the image rows from xp3_row0 on)
            const int hi = (d.g_h0 + ho) * d.in_sh + d.tap_h0 - d.xp3_row0, wi = (d.g_w0 + wo) * d.in_sw + d.tap_w0;
            const int pix = ((int)b * HinP + hi) * Wp + wi;                           // may be -1 at the very first pixel: wraps to an out-of-range offset
            a_base[j] = (unsigned)(pix * SB + unit * 16);
            unsigned bad = 0;
            for (int th = 0; th * d.TW < d.ntaps; ++th) {
                const int hh = hi + th * d.tap_sh;
                if (hh < 0 || hh >= HinP) bad |= 1u << th;
            }
            a_bad[j] = bad;
        }
    }
    unsigned b_voff[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        const int L = inst * 64 + lane;
        const int pl = L / (2 * BN), n = (L >> 1) % BN, half = L & 1;
        b_voff[j] = (inst < B_INST && n0 + n < d.N) ? (unsigned)((pl * d.N + n0 + n) * 32 + 16 * (half ^ ((n >> 3) & 1))) : OOB;
    }

    // issue state (SGPRs): the K tile being issued = (tap q_tap = (q_th, q_tw), chunk q_ch)
    int q_th = 0, q_tw = 0, q_ch = 0, q_kt = 0;
    unsigned i_asoff = 0, i_bsoff = 0, i_dead = 0;
    char* i_tile = smem;
    auto begin_tile = [&](char* dst) {               // branch-free: the K loop stays one basic block (selects, no scalar branches)
        i_tile = dst;
        i_dead = q_kt >= NKT ? OOB : 0u;           // past the K range (NKT % KS != 0): range-check zeros
        // per-lane validity of the rows above / below the image under filter row q_th.  The tap displacement goes into the VECTOR
        // offset: the buffer range check looks at the vector offset alone, and the origin pixel of a row under SAME padding may lie
        // above the tensor (a "negative" offset that only the displacement makes valid) - the scalar offset carries the chunk only
        const unsigned tapd = (unsigned)((q_th * d.tap_sh * Wp + q_tw * d.tap_sw) * SB);
#pragma unroll
        for (int j = 0; j < A_PW; ++j) a_cur[j] = ((a_bad[j] >> q_th) & 1u) ? OOB : a_base[j] + tapd;
        i_asoff = (unsigned)q_ch * d.xp3_cstride;
        i_bsoff = (unsigned)q_kt * (unsigned)(d.N * SB);
        ++q_kt; ++q_ch;
        const int wrap_c = q_ch == nchunk ? 1 : 0;
        q_ch = wrap_c ? 0 : q_ch;
        q_tw += wrap_c;
        const int wrap_w = q_tw == d.TW ? 1 : 0;
        q_tw = wrap_w ? 0 : q_tw;
        q_th += wrap_w;
    };
    auto issue_one = [&](int s) {                   // s = compile-time slot inside a K tile: activation slots first
        if (s < A_PW) {
            const int inst = wave + 4 * s;
            if (A_INST % 4 == 0 || inst < A_INST) dma16(x_rsrc, (float*)(i_tile + inst * 1024), a_cur[s] | i_dead, i_dead ? 0u : i_asoff);
        } else {
            const int j = s - A_PW;
            const int inst = wave + 4 * j;
            if (B_INST % 4 == 0 || inst < B_INST) dma16(w_rsrc, (float*)(i_tile + A_BYTES + inst * 1024), b_voff[j] | i_dead, i_dead ? 0u : i_bsoff);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing (bytes inside a K tile image)
    const int li = lane & 31, kk = lane >> 5;
    int a_foff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int sl = wm * WM + i * 32 + li;
        a_foff[i] = H2 ? sl * 64 + 16 * (kk ^ ((sl >> 2) & 3)) : sl * 96 + 16 * (kk ^ ((sl >> 3) & 1));     // plane pl: ^ (pl * 32) / + pl * 32
    }
    const int b_foff = A_BYTES + (wn * WN + li) * 32 + 16 * (kk ^ ((li >> 3) & 1));
    constexpr int TA[6] = {H2 ? 1 : 0, 0, H2 ? 0 : 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // bf16x3: hh, hm, mh, hl, lh, mm; fp16x2: lh, hl, hh

    // ---- prologue: group 0 ----
    int stage = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        begin_tile(smem + ks * T_BYTES);
#pragma unroll
        for (int s = SPT - 1; s >= 0; --s) issue_one(s);
    }

    // one group: KS K tiles of 6*MT*NT MFMAs; the DMA of the next group is spread over the first MFMA slots; fragments are
    // double-buffered over the K tiles.  Every MFMA slot is fenced (conv3p.hip: left alone hipcc sinks the prefetch reads and
    // clusters the MFMAs of one accumulator).
    auto group = [&](auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const char* st = smem + stage * ST_BYTES;
        char* const nst = smem + (stage ^ 1) * ST_BYTES;
        bf16x8 fq[2][NF];                            // (raw 16-byte fragments: reinterpreted as 8 x fp16 for the fp16x2 planes)
        auto load_frag = [&](int buf, int ks, int f) {
            const int pl = f / (MT + NT), r = f - pl * (MT + NT);
            const char* tb = st + ks * T_BYTES;
            if (r < MT) fq[buf][f] = *reinterpret_cast<const bf16x8*>(tb + (H2 ? (a_foff[r] ^ (pl * 32)) : a_foff[r] + pl * 32));
            else fq[buf][f] = *reinterpret_cast<const bf16x8*>(tb + b_foff + pl * (BN * 32) + (r - MT) * 32 * 32);
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) load_frag(0, 0, f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1;
#pragma unroll
            for (int tt = 0; tt < NPR; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int k = (tt * MT + i) * NT + j;
                        const int idx = ks * NM1 + k;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (H2)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, fq[cb][TA[tt] * (MT + NT) + i]),
                                                                               __builtin_bit_cast(f16x8g, fq[cb][TB[tt] * (MT + NT) + MT + j]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[cb][TA[tt] * (MT + NT) + i], fq[cb][TB[tt] * (MT + NT) + MT + j],
                                                                                acc[i][j], 0, 0, 0);
                        if (ks + 1 < KS) {
#pragma unroll
                            for (int f = 0; f < NF; ++f)
                                if (f * NM1 / NF == k) load_frag(cb ^ 1, ks + 1, f);
                        }
                        if (ISSUE && idx < CNT) {
                            const int tks = idx / SPT, s = idx - tks * SPT;
                            if (s == 0) begin_tile(nst + tks * T_BYTES);
                            issue_one(SPT - 1 - s);                       // filter slots first
                        }
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage ^= 1;
    };

    for (int it = 0; it + 1 < G; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // group `it` has landed (this wave's share); the barrier makes it everyone's
        lds_barrier();
        group(std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    group(std::false_type{});

    float osc = 1.f;
    if constexpr (H2) osc = d.h2_a_inv[0] * d.h2_w_inv[0];
    lds_barrier();                                           // every wave is done with the last group's fragments: the whole ring is free
    if constexpr (EPI == 1) {
        // ---- fused decoder tail: the tile (one mask frame of one window: conv3g_ok) through the ring a wave-row at a time ----
        if constexpr (H2) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] *= osc;            // 2^-(ka + kw): exact
        }
        igemm_epilogue_maskmix<BM, BN, WM, WN, SMEM_BYTES / 4>(d, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
        return;
    }
    // ---- epilogue straight from the accumulators (conv3h.hip: a workgroup's way out is bound by the number of instructions its waves
    // issue - the form staged through LDS took 400 - 1 000 of them): branch-free, a row beyond M gets the out-of-range buffer offset
    // and the scale 0 (which keeps it out of the sums); C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    {
        const int colb = n0 + wn * WN + li;                          // column of this lane in N block j = 0
        const bool plain = d.bias == nullptr && !d.relu_out;
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.y, 0, d.y_bytes, 0x00020000);
        const unsigned ldy4 = (unsigned)d.ldy * 4u;
        const int mrow = m0 + wm * WM + 4 * kk;                      // first output row of this lane
        const unsigned rbase = (unsigned)mrow * ldy4;
        float bias_j[NT], cs[NT], cq[NT];
        float amax = 0.f;
        unsigned col4[NT];                                           // byte offset of the lane's column, OOB beyond N
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool ok = colb + 32 * j < d.N;
            col4[j] = ok ? (unsigned)(colb + 32 * j) * 4u : OOB;
            bias_j[j] = (d.bias != nullptr && ok) ? d.bias[colb + 32 * j] : 0.f;
            cs[j] = 0.f; cq[j] = 0.f;
        }
        auto epilogue = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;         // whole N tile inside N, no bias, no ReLU
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = i * 32 + (e & 3) + 8 * (e >> 2);  // (compile-time row offset)
                    const bool ok = mrow + ro < d.M;
                    const float sc = ok ? osc : 0.f;
                    const unsigned roff = ok ? rbase + (unsigned)ro * ldy4 + (FAST ? (unsigned)(colb * 4) : 0u) : OOB;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float v = acc[i][j][e] * sc;
                        if (FAST) {
                            cs[j] += v;
                            cq[j] = __builtin_fmaf(v, v, cq[j]);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, roff, 128 * j, 0);
                        } else {
                            const unsigned off = ((roff | col4[j]) & OOB) ? OOB : roff + col4[j];
                            if (col4[j] & OOB) v = 0.f;
                            cs[j] += v;
                            cq[j] = __builtin_fmaf(v, v, cq[j]);
                            v += bias_j[j];
                            if (d.relu_out) v = fmaxf(v, 0.f);
                            amax = fmaxf(amax, (off & OOB) ? 0.f : fabsf(v));       // (IgemmDesc::amax_out: what is STORED only)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, off, 0, 0);
                        }
                    }
                }
        };
        // exact max |y| of what this launch stores (round 6: the audio encoder's convs on planes publish the maximum of their half of the
        // concat buffer like the register-staged kernels' epilogues do); only the general form carries it
        if (plain && n0 + BN <= d.N && d.amax_out == nullptr) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (d.amax_out != nullptr) igemm_publish_amax(d.amax_out, amax);           // (uniform condition: every lane of every wave arrives)
        if (d.stats != nullptr) {                 // per-channel (sum, sumsq) of the raw output -> fp64 accumulators [2][N]
            lds_barrier();                        // every wave is done with the last group's fragments: the ring is free
            float* const red = reinterpret_cast<float*>(smem);               // [WAVES_M * 2 (lane halves)][2][BN]
            static_assert(WAVES_M * 2 * 2 * BN * 4 <= SMEM_BYTES, "statistics staging must fit the ring");
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                red[((wm * 2 + kk) * 2 + 0) * BN + wn * WN + j * 32 + li] = cs[j];
                red[((wm * 2 + kk) * 2 + 1) * BN + wn * WN + j * 32 + li] = cq[j];
            }
            lds_barrier();
            for (int t = tid; t < 2 * BN; t += 256) {
                const int which = t / BN, col = t - which * BN;
                if (n0 + col < d.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < WAVES_M * 2; ++g) sum += red[(g * 2 + which) * BN + col];
                    atomicAdd(&d.stats[(long)which * d.N + n0 + col], (double)sum);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int KS, bool H2, int EPI = 0>
static int launch_conv3g(const IgemmDesc& d, hipStream_t s) {
    const int per = (cdiv(d.M, BM) + 7) / 8;
    const int grid = 8 * per * cdiv(d.N, BN);
    hipLaunchKernelGGL((conv3g_kernel<BM, BN, WM, WN, KS, H2, EPI>), dim3(grid, 1, d.grp.G), dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// can conv3g_kernel run this problem (given the planes)?  dense NHWC output, taps whose horizontal reach stays on the plane row +
// its pad pixel, at most 32 filter rows, 16-channel chunks
bool conv3g_ok(const IgemmDesc& d) {
    if (d.Cin % 16 || d.K != d.ntaps * d.Cin || d.Kpad != d.K) return false;
    if (d.ntaps < 1 || d.TW < 1 || d.ntaps % d.TW || d.ntaps / d.TW > 32) return false;
    if (d.in_scale != nullptr || d.bn_in.acc != nullptr) return false;
    if (d.mm_out != nullptr) {       // the fused decoder tail: a depth-to-space tile of 64 grid columns x (4 pixels x 32 tracks) = one mask frame of one window
        if (d.xp3_fmt != 1 || d.Cout != 32 || d.dsh * d.dsw <= 1 || d.Wg % 64 || (d.dsw * d.Cout) % 128 || d.splitk != 1 || !d.mm_coeffs) return false;
    } else {
        if (d.dsh * d.dsw != 1 || d.g_h0 != 0 || d.g_w0 != 0) return false;
        if (d.y_rstride != (long)d.Wg * d.ldy || (d.M > d.Hg * d.Wg && d.y_bstride != (long)d.Hg * d.Wg * d.ldy)) return false;
    }
    // horizontal reach of the taps over the whole grid: the pad pixel closing every plane row is column -1 and column Win
    const int t_lo = std::min(0, (d.TW - 1) * d.tap_sw), t_hi = std::max(0, (d.TW - 1) * d.tap_sw);
    const int w_lo = d.g_w0 * d.in_sw + d.tap_w0 + t_lo, w_hi = (d.g_w0 + d.Wg - 1) * d.in_sw + d.tap_w0 + t_hi;
    return w_lo >= -1 && w_hi <= d.Win;
}

int conv3g_dispatch(const IgemmDesc& d_in, IgemmTile tile, hipStream_t s) {
    IgemmDesc d = d_in;
    if (!d.xp3 || d.p3_np <= 0) return fail(SAGEN_ERR_NULL, "conv3g: the P3 activation planes are missing");
    if (d.splitk != 1) return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: no split-K");
    if (!conv3g_ok(d)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: geometry not supported");
    if ((long)(d.M + 512) * d.Wg >= (1L << 32) || ((long)(d.M + 512) / d.Wg + 1) * d.Hg >= (1L << 32))
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: too many pixels for 32-bit index arithmetic");
    if ((long)d.p3_np * 96 >= (1L << 31) || (long)d.xp3_cstride * (d.Cin / 16) >= (1L << 31) || d.xp3_bytes == 0)
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: the activation planes exceed 2 GiB buffer addressing (use a smaller batch)");
    const bool mm = tile == TILE_P3GH_MM_64x128_K2 || tile == TILE_P3GH_MM_64x128_K4 || tile == TILE_P3GH_MM_128x128_K2 || tile == TILE_P3GH_MM_128x256_K2;
    if (mm != (d.mm_out != nullptr)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: the fused decoder tail and the tile do not match");
    const bool h2 = mm || tile == TILE_P3GH_128x64_K3 || tile == TILE_P3GH_64x64_K4 || tile == TILE_P3GH_128x128_K2 || tile == TILE_P3GH_64x128_K3;
    if (h2 != (d.xp3_fmt == 1)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: the planes' format does not match the tile");
    if (h2 && (!d.wh2 || !d.h2_a_inv || !d.h2_w_inv)) return fail(SAGEN_ERR_NULL, "conv3g: the fp16x2 filter planes / scales are missing");
    if (!mm) {
        const long y_bytes = ((long)(d.M - 1) * d.ldy + d.N) * 4;       // extent of the dense output for the epilogue's buffer stores
        if (y_bytes >= (1L << 31)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: the output exceeds 2 GiB buffer addressing (use a smaller batch)");
        d.y_bytes = (unsigned)y_bytes;
    }
    d.p3_magic_wp = (unsigned)((1UL << 32) / (unsigned)d.Wg) + 1u;      // (reused fields: here the divisors are the OUTPUT grid's Wg, Hg)
    d.p3_magic_h = (unsigned)((1UL << 32) / (unsigned)d.Hg) + 1u;
    switch (tile) {
        case TILE_P3G_128x64_K2: return launch_conv3g<128, 64, 64, 32, 2, false>(d, s);
        case TILE_P3G_64x64_K2: return launch_conv3g<64, 64, 32, 32, 2, false>(d, s);
        case TILE_P3G_64x128_K2: return launch_conv3g<64, 128, 32, 64, 2, false>(d, s);
        case TILE_P3G_128x128_K1: return launch_conv3g<128, 128, 64, 64, 1, false>(d, s);
        case TILE_P3GH_128x64_K3: return launch_conv3g<128, 64, 64, 32, 3, true>(d, s);
        case TILE_P3GH_64x64_K4: return launch_conv3g<64, 64, 32, 32, 4, true>(d, s);
        case TILE_P3GH_128x128_K2: return launch_conv3g<128, 128, 64, 64, 2, true>(d, s);
        case TILE_P3GH_64x128_K3: return launch_conv3g<64, 128, 32, 64, 3, true>(d, s);
        case TILE_P3GH_MM_64x128_K2: return launch_conv3g<64, 128, 32, 64, 2, true, 1>(d, s);
        case TILE_P3GH_MM_64x128_K4: return launch_conv3g<64, 128, 32, 64, 4, true, 1>(d, s);
        case TILE_P3GH_MM_128x128_K2: return launch_conv3g<128, 128, 64, 64, 2, true, 1>(d, s);
        case TILE_P3GH_MM_128x256_K2: return launch_conv3g<128, 256, 64, 128, 2, true, 1>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "conv3g: bad tile id %d", (int)tile);
    }
}

}  // namespace sagen
