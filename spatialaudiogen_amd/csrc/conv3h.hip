// conv3h_kernel: conv3p_kernel (dense 3x3 stride-1 SAME convolution over pre-split activation planes; reference op tf.nn.convolution,
// core.py:206, as the ResNet18 trunk uses it: resnet.py:141-190 / 215-235) on TWO fp16 planes per operand instead of three bf16
// planes - three matrix products per fp32 product instead of six, 4 instead of 6 bytes per operand element.
//
// Arithmetic ("fp16x2").  An fp32 value v with |v| in fp16's NORMAL range splits as
//     hi = rne_f16(v),  lo = rne_f16(v - hi)            (v - hi is exact in fp32)
// into 11 + 11 significant bits plus the sign of the residual: |v - hi - lo| <= 2^-23.5 |v| - fp32's own resolution.  The product
// a*b is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation; the dropped lo*lo is below 2^-22
// relative.  bf16 was chosen for the general kernels because it has fp32's EXPONENT range; fp16 does not (normal range 6.1e-5 ..
// 65504, below that the absolute resolution stays at 2^-25 - v_mfma honours fp16 subnormals: tools/probe/f16_denorm.hip), so this
// kernel is used only where the operands' magnitudes are known, and both are scaled into the middle of that range by exact powers
// of two:
//   * filters: per layer, 2^kw with max |w| 2^kw in [512, 1024) (h2_filter_pack: absmax on the device, planes [K/16][2][N][16]);
//   * activations: the plane-writing pass (p3.hip, format 1) derives 2^ka from STATISTICS the forward already holds - channel c of
//     relu(bn(y)) has mean beta_c and standard deviation |gamma_c| sqrt(var_c / (var_c + eps)); the residual branch of a block merge adds
//     the bound of the block input (tracked from pass to pass) or, behind a 1x1 projection, |mean| + 8 sigma of the projection's output
//     (its conv accumulates (sum, sumsq) like the batch-norm convs do): bound = max_c(|beta_c| + 8 std_c) [+ residual bound] is scaled
//     into [512, 1024) and the planes saturate at +-65000 - 64 times beyond eight standard deviations; exact zeros (ReLU) stay exact.
// The epilogue multiplies the tile by 2^-(ka + kw) (exact).  Measured against an fp64 evaluation of the network this path is as accurate as the
// six-product bf16x3 path (fewer products = fewer fp32 accumulation roundings): DESIGN.md 3.2, profiles/r04_accuracy_modes.jsonl.
//
// Everything else is conv3p_kernel: P layout [Cin/16][NP][2 planes][16 ch] fp16 (64 B per pixel and chunk), NP = B*H*(W+1) with one
// zero pixel closing every image row; K order (dh, chunk, dw): per group the workgroup stages the activation tile once (with one
// halo pixel either side) and the three dw filter tiles by LDS-DMA; fragments by ds_read_b128; two-stage ring; one barrier per
// group.  The LDS image of a slot is 64 B = four 16-byte units (plane, half); unit u of slot s sits at position u ^ ((s >> 2) & 3):
// the 16 lanes of one ds_read_b128 group then cover all 64 banks for ANY slot base (with 96-byte slots conv3p_kernel needs only the
// half swap).  40 KiB of LDS per 128x64 workgroup: three per CU.
#include "igemm3_common.h"
#include "h2_planes.h"

namespace sagen {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// KC = 16-channel chunks per barrier step (group): the deep layers (Cin 256 / 512: 48 / 96 groups of only 9*MT*NT MFMAs) are bound by
// the per-group latency (barrier, DMA round trip, first fragments) - two chunks per group halve the groups.  No slack behind the ring:
// the fragment reads of the two dropped rows run past the activation image into the filter image of the SAME stage.
//
// AR = depth of the ACTIVATION ring.  AR = 2: one two-stage ring of (activation image, filter images) - group g+1 is issued under
// group g.  AR = 3 (conv3hr_kernel): the activation images have a ring of their own, three deep, the filter images keep two stages:
// under group g the workgroup issues the filter images of group g+1 FIRST and then the activation image of group g+2, and the wait
// in front of group g+1 is the counted vmcnt(activation instructions of one group) - everything older than the youngest activation
// image has landed, that image has a whole further group to arrive.  The activation images are the HBM / far-L2 reads (the filter is
// the same 147 KB..4.7 MB for every workgroup and sits in the L2), so the extra depth goes where the latency is, for 16 KB of LDS more
// per 256x64 workgroup (72 KB: still two per CU) instead of the 28 KB a third full stage would cost (84 KB: one per CU).
constexpr int conv3h_lds(int BM, int BN, int KC, int AR = 2) { return KC * (AR * BM * 64 + 2 * 3 * 2 * BN * 32); }
constexpr int conv3h_wgs_per_cu(int BM, int BN, int KC, int AR = 2) {
    return 160 * 1024 / conv3h_lds(BM, BN, KC, AR) >= 3 ? 3 : (160 * 1024 / conv3h_lds(BM, BN, KC, AR) >= 2 ? 2 : 1);
}

template <int BM, int BN, int WM, int WN, int KC, int AR>
__device__ __forceinline__ void conv3h_body(const IgemmDesc& d) {
    static_assert(AR == 2 || AR == 3, "activation ring of two or three stages");
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int BME = BM - 2;                            // rows the workgroup owns: the image is BM slots = BME outputs + one halo pixel either side
    constexpr int A_INST = BM / 16;                        // LDS-DMA wave-instructions (1 KiB) of one activation stage: BM slots x 64 B
    constexpr int B_IPT = BN / 16;                         // per tap: 2 planes x BN rows x 32 B
    constexpr int B_INST = 3 * B_IPT;
    constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;
    constexpr int A_BYTES = A_INST * 1024, B_BYTES = B_INST * 1024;
    constexpr int ST_BYTES = KC * (A_BYTES + B_BYTES);     // a stage: the KC activation images, then the KC filter images
    constexpr int NM1 = 3 * MT * NT;                       // MFMAs per tap
    constexpr int NMG = 3 * KC * NM1;
    constexpr int SPT = A_PW + B_PW;                       // DMA slots per wave and chunk
    constexpr int CNT_MAX = KC * SPT;
    static_assert(CNT_MAX <= NMG, "one DMA slot per MFMA slot at most");
    constexpr int NF = 2 * (MT + NT);
    constexpr int SMEM_BYTES = KC * (AR * A_BYTES + 2 * B_BYTES);
    static_assert(SMEM_BYTES == conv3h_lds(BM, BN, KC, AR), "occupancy bound uses the same footprint");
    static_assert(AR == 2 || A_INST % 4 == 0, "counted vmcnt: every wave issues the same number of activation instructions");
    // where stage sa of the activation ring / stage sb of the filter ring start (AR = 2: the interleaved layout [A0 B0 A1 B1])
    auto a_stage = [&](int sa) { return AR == 2 ? sa * ST_BYTES : sa * (KC * A_BYTES); };
    auto b_stage = [&](int sb) { return AR == 2 ? sb * ST_BYTES + KC * A_BYTES : AR * KC * A_BYTES + sb * (KC * B_BYTES); };
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];     // ONE shared object (conv3p.hip)

    const int tid = threadIdx.x;
#ifdef SAGEN_TRACE      // debug builds (tools/trace_conv3h.py): life of every workgroup - entry, K loop entered, K loop left, epilogue done
    const size_t trc_wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // (dh-split / grouped launches: every workgroup its own record)
    unsigned long long* const trc = (d.trace && trc_wg < 8192) ? (unsigned long long*)d.trace + trc_wg * 8 : nullptr;
#define C3H_TRC(k) do { if (trc && tid == 0) trc[k] = __builtin_amdgcn_s_memtime(); } while (0)
    if (trc && tid == 0) {
        trc[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: wave / simd / cu / sh / se
        trc[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
        trc[6] = __builtin_amdgcn_s_memrealtime();                 // 100 MHz, common to the XCDs
    }
    C3H_TRC(0);
    // ... and the phases of every group of two workgroups (an early and a late one): top, tiles landed, barrier passed, first
    // fragments in registers, last MFMA issued - [2][64 groups][8] behind the per-workgroup records
    const int gsel = (blockIdx.y | blockIdx.z) ? -1 : ((int)blockIdx.x == 8 ? 0 : ((int)blockIdx.x == (int)gridDim.x - 64 ? 1 : -1));
    unsigned long long* const gtr = (d.trace && gsel >= 0) ? (unsigned long long*)d.trace + (size_t)8192 * 8 + (size_t)gsel * 64 * 8 : nullptr;
    int g_idx = 0;
#define C3H_GTRC(k) do { if (gtr && tid == 0 && g_idx < 64) gtr[g_idx * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C3H_TRC(k) do { } while (0)
#define C3H_GTRC(k) do { } while (0)
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int W = d.Win, H = d.Hin, Wp = W + 1, NP = d.p3_np;
    const int nchunk = d.Cin >> 4;
    // dh-split (round 6; d.splitk == 3, grid.y = filter row): workgroup (tile, dh) contracts ONE vertical tap - a third of the K loop,
    // three times the workgroups - and writes a raw partial tile [dh][M][N]; splitk_reduce_stats_kernel sums the three and takes the
    // batch-norm statistics.  The grid runs all dh = 0 workgroups first: the third of the filter they stream stays in the XCD's L2.
    const bool dhs = d.splitk == 3;
    const int dh_z = dhs ? (int)blockIdx.y : 0;
    const int G = (dhs ? 1 : 3) * nchunk / KC;             // (nchunk % KC == 0: conv3h_dispatch)
    const int nM = (NP + BME - 1) / BME, nN = (d.N + BN - 1) / BN;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.xp3, 0, d.xp3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.wh2, 0, d.wh2_bytes, 0x00020000);

    const float osc = d.h2_a_inv[0] * d.h2_w_inv[0];      // (read here: two dependent-latency loads the epilogue would otherwise wait for)

    // block -> tile: XCD x owns M tiles [x*per, (x+1)*per) (vertically neighbouring tiles share input rows in one L2)
    // (grouped launch: group g rotates the owner by g, so that the XCD a short tile list leaves without tiles - 14 tiles of 256 rows at
    //  stage 5 are 2 + 2 + .. + 0 - is another one for every group; a workgroup's physical XCD stays blockIdx.x & 7: the grid is a
    //  multiple of 8 wide, and all tiles of one owner still run on one XCD)
    const int xcd = (blockIdx.x + blockIdx.z) & 7;
    const int per = (nM + 7) >> 3;
    const int t_run = blockIdx.x >> 3;
    const int my_tiles = max(min((xcd + 1) * per, nM) - xcd * per, 0) * nN;
    if (t_run >= my_tiles) return;
    const int tq = t_run / nN;
    const int m0 = (xcd * per + tq) * BME;                 // first PADDED pixel of the tile
    const int n0 = (t_run - tq * nN) * BN;

    // ---- per-lane DMA state: activation unit U = inst*64 + lane = (slot, position v); slot <-> padded pixel m0 - 1 + slot ----
    unsigned a_v0[A_PW], a_v1[A_PW], a_v2[A_PW], a_cur[A_PW];      // per vertical tap dh = -1 / 0 / +1; the current one
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int inst = wave + 4 * j;
        const int U = inst * 64 + lane;
        const int slot = U >> 2, v = U & 3;
        const int u = v ^ ((slot >> 2) & 3);               // which (plane, half) of the pixel lands at position v
        const int p = m0 - 1 + slot;
        unsigned bad = 7u;
        if (inst < A_INST && p >= 0 && p < NP) {
            const unsigned row = __umulhi((unsigned)p, d.p3_magic_wp);          // p / Wp  (exact: conv3h_dispatch)
            const int h = (int)(row - __umulhi(row, d.p3_magic_h) * (unsigned)H);
            bad = (h == 0 ? 1u : 0u) | (h == H - 1 ? 4u : 0u);
        }
        const int base = p * 64 + u * 16;
        a_v0[j] = (bad & 1u) ? OOB : (unsigned)(base - Wp * 64);
        a_v1[j] = (bad & 2u) ? OOB : (unsigned)base;
        a_v2[j] = (bad & 4u) ? OOB : (unsigned)(base + Wp * 64);
    }
    // filter DMA lanes: per tap the image is [plane][BN rows][32 B]
    unsigned b_voff[B_PW];
    int b_tapoff[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int inst = wave + 4 * j;
        b_tapoff[j] = __builtin_amdgcn_readfirstlane((inst / B_IPT) * nchunk * d.N * 64);
        const int r = inst % B_IPT;
        const int L = r * 64 + lane;
        const int pl = L / (2 * BN), n = (L >> 1) % BN, half = L & 1;
        b_voff[j] = (inst < B_INST && n0 + n < d.N) ? (unsigned)((pl * d.N + n0 + n) * 32 + 16 * (half ^ ((n >> 3) & 1))) : OOB;
    }

    // issue state (SGPRs): the activation group and the filter group being issued (AR = 2: the same group; AR = 3: the activation
    // tracker runs one group ahead of the filter tracker)
    int qa_dh = dh_z, qa_ch = 0, cur_dh = -1, qb_dh = dh_z, qb_ch = 0;
    unsigned i_asoff = 0, i_bsoff = 0;
    char* i_astage = smem;
    char* i_bstage = smem;
    auto begin_issue_a = [&](int sa) {
        i_astage = smem + a_stage(sa);
        if (qa_dh != cur_dh) {
            cur_dh = qa_dh;
#pragma unroll
            for (int j = 0; j < A_PW; ++j) a_cur[j] = qa_dh == 0 ? a_v0[j] : (qa_dh == 1 ? a_v1[j] : a_v2[j]);
        }
        i_asoff = (unsigned)qa_ch * d.xp3_cstride;
        qa_ch += KC;
        if (qa_ch == nchunk) { qa_ch = 0; ++qa_dh; }
    };
    auto begin_issue_b = [&](int sb) {
        i_bstage = smem + b_stage(sb);
        i_bsoff = (unsigned)((qb_dh * 3) * nchunk + qb_ch) * (unsigned)(d.N * 64);
        qb_ch += KC;
        if (qb_ch == nchunk) { qb_ch = 0; ++qb_dh; }
    };
    auto issue_a = [&](int kc, int s) {         // activation slot s of chunk kc (compile-time indices)
        const int inst = wave + 4 * s;
        if (A_INST % 4 == 0 || inst < A_INST)
            dma16(x_rsrc, (float*)(i_astage + kc * A_BYTES + inst * 1024), a_cur[s], i_asoff + (unsigned)kc * d.xp3_cstride);
    };
    auto issue_b = [&](int kc, int j) {         // filter slot j of chunk kc
        const int inst = wave + 4 * j;
        if (4 * (j + 1) <= B_INST || inst < B_INST)
            dma16(w_rsrc, (float*)(i_bstage + kc * B_BYTES + inst * 1024), b_voff[j], i_bsoff + (unsigned)b_tapoff[j] + (unsigned)(kc * d.N * 64));
    };
    // n = position in the group's issue order.  AR = 2: per chunk (last chunk first) its filter slots, then its activation slots;
    // AR = 3: ALL filter slots, then all activation slots (the counted wait leaves exactly the activation instructions in flight)
    auto issue_nth = [&](int n) {
        if (AR == 2) {
            const int sg = CNT_MAX - 1 - n;
            const int kc = sg / SPT, s = sg - kc * SPT;
            if (s < A_PW) issue_a(kc, s); else issue_b(kc, s - A_PW);
        } else if (n < KC * B_PW) {
            issue_b(n / B_PW, n % B_PW);
        } else {
            const int m = n - KC * B_PW;
            issue_a(m / A_PW, m % A_PW);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing (bytes inside a stage): plane 0 at a_foff, plane 1 at a_foff ^ 32 ----
    const int li = lane & 31, kk = lane >> 5;
    int a_foff[3][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) {
            const int sl = wm * WM + i * 32 + li + dwi;              // output row r sits at slot r + 1; tap dw reads slot r + dw
            a_foff[dwi][i] = sl * 64 + 16 * (kk ^ ((sl >> 2) & 3));
        }
    const int b_foff = (wn * WN + li) * 32 + 16 * (kk ^ ((li >> 3) & 1));     // (inside a filter stage)
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};             // lo*hi, hi*lo, hi*hi

    int sa = 0, sb = 0;                                              // the stages being consumed
    begin_issue_a(0);
    begin_issue_b(0);
#pragma unroll
    for (int n = 0; n < CNT_MAX; ++n) issue_nth(n);                  // filter tiles first
    if (AR == 3) {
        begin_issue_a(1);                                            // (G >= 3: three vertical taps)
#pragma unroll
        for (int n = KC * B_PW; n < CNT_MAX; ++n) issue_nth(n);
    }

    // one group: its 9*MT*NT MFMAs with the DMA of the coming group(s) spread between them; every MFMA slot is fenced
    // (conv3p.hip: the source order IS the schedule).  ISSUE: 0 = nothing, 1 = the filter slots only, 2 = everything
    auto group = [&](auto issue_tag) {
        constexpr int ISSUE = decltype(issue_tag)::value;
        constexpr int NISSUE = ISSUE == 2 ? CNT_MAX : (ISSUE == 1 ? KC * B_PW : 0);
        const char* st_a = smem + a_stage(sa);
        const char* st_b = smem + b_stage(sb);
        f16x8 fq[2][NF];                                             // [buffer][plane * (MT+NT) + (i | MT + j)]
        auto load_frag = [&](int buf, int t, int f) {               // t = kc*3 + dwi: the (chunk, horizontal tap) step inside the group
            const int kc = t / 3, dwi = t - 3 * kc;
            const int pl = f / (MT + NT), r = f - pl * (MT + NT);
            if (r < MT) fq[buf][f] = *reinterpret_cast<const f16x8*>(st_a + kc * A_BYTES + (a_foff[dwi][r] ^ (pl * 32)));
            else fq[buf][f] = *reinterpret_cast<const f16x8*>(st_b + b_foff + kc * B_BYTES + (dwi * 2 + pl) * (BN * 32) + (r - MT) * 32 * 32);
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) load_frag(0, 0, f);
        C3H_GTRC(3);
#pragma unroll
        for (int t = 0; t < 3 * KC; ++t) {
            const int cb = t & 1;
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int k = (tt * MT + i) * NT + j;
                        const int idx = t * NM1 + k;
                        __builtin_amdgcn_sched_barrier(0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[cb][TA[tt] * (MT + NT) + i], fq[cb][TB[tt] * (MT + NT) + MT + j],
                                                                           acc[i][j], 0, 0, 0);
                        if (t + 1 < 3 * KC) {
#pragma unroll
                            for (int f = 0; f < NF; ++f)
                                if (f * NM1 / NF == k) load_frag(cb ^ 1, t + 1, f);
                        }
#pragma unroll
                        for (int n = 0; n < NISSUE; ++n)
                            if (idx == n) issue_nth(n);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        C3H_GTRC(4);
#ifdef SAGEN_TRACE
        ++g_idx;
#endif
        sa = sa + 1 == AR ? 0 : sa + 1;
        sb ^= 1;
    };

    C3H_TRC(1);                                                      // prologue done: index setup, first tiles issued
    if (AR == 2) {
        for (int it = 0; it + 1 < G; ++it) {
            C3H_GTRC(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C3H_GTRC(1);
            lds_barrier();
            C3H_GTRC(2);
            begin_issue_a(sa ^ 1);
            begin_issue_b(sb ^ 1);
            group(std::integral_constant<int, 2>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        group(std::integral_constant<int, 0>{});
    } else {
        // in front of group g the youngest instructions in flight are the KC*A_PW of activation image g+1: everything older - the
        // filter images and the activation image of group g - has landed once at most those are outstanding
        for (int it = 0; it + 2 < G; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KC * A_PW) : "memory");
            lds_barrier();                                           // every wave is done with group it-1: its stages are free
            begin_issue_b(sb ^ 1);                                   // filter images of group it+1
            begin_issue_a(sa == 0 ? 2 : sa - 1);                     // activation image of group it+2 -> the stage group it-1 read
            group(std::integral_constant<int, 2>{});
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KC * A_PW) : "memory");
        lds_barrier();
        begin_issue_b(sb ^ 1);
        group(std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        group(std::integral_constant<int, 0>{});
    }

    C3H_TRC(2);
    // ---- epilogue straight from the accumulators: x 2^-(ka + kw), bias / ReLU, batch-norm statistics.  C/D layout of 32x32: col =
    // lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a store instruction writes two 128-byte row segments.  A workgroup's way
    // out is bound by the NUMBER of instructions a wave has to issue (tools/trace_conv3h.py: 9.4 k cycles alone, 14 k beside the other
    // workgroup of the CU, for the 1 500 instructions of the form staged through LDS or of a branchy direct form - with or without
    // the stores and the atomics), so this form is branch-free: a dropped row (pad pixel, beyond the tile or the tensor) gets the
    // out-of-range buffer offset and the scale 0, which also keeps it out of the sums - 4 instructions per value, 12 per row.
    const int colb = n0 + wn * WN + li;                              // column of this lane in N block j = 0
    const bool plain = d.bias == nullptr && !d.relu_out;             // (the batch-norm convs of the trunk)
    // (dh-split: conv3h_dispatch pointed y at the partials [3][M][N] with ldy = N; partial dh_z starts y_bytes further on)
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)d.y + (size_t)dh_z * d.y_bytes), 0, d.y_bytes, 0x00020000);
    const unsigned ldy4 = (unsigned)d.ldy * 4u;
    const int plim = min(NP, m0 + BME);
    float bias_j[NT], cs[NT], cq[NT];
    unsigned col4[NT];                                               // byte offset of the lane's column, OOB beyond N
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const bool ok = colb + 32 * j < d.N;
        col4[j] = ok ? (unsigned)(colb + 32 * j) * 4u : OOB;
        bias_j[j] = (d.bias != nullptr && ok) ? d.bias[colb + 32 * j] : 0.f;
        cs[j] = 0.f; cq[j] = 0.f;
    }
    // the rows' store offsets (dense pixel x ldy, or the out-of-range offset for a dropped row) once per workgroup: one row per
    // thread into the (idle) ring, read back four rows per ds_read_b128 - 3 instead of 10 instructions per row and lane
    lds_barrier();                                   // every wave is done with the last group's fragments
    unsigned* const s_roff = reinterpret_cast<unsigned*>(smem);      // [BM]
    for (int r = tid; r < BM; r += 256) {
        const int p = m0 + r;
        const unsigned row = __umulhi((unsigned)p, d.p3_magic_wp);   // p / Wp = b*H + h: one pad pixel per preceding row
        s_roff[r] = (p < plim && (unsigned)p - row * (unsigned)Wp < (unsigned)W) ? ((unsigned)p - row) * ldy4 : OOB;
    }
    lds_barrier();
    // FAST (uniform): whole N tile inside N, no bias, no ReLU - the batch-norm convs of the trunk; the other form keeps every case
    auto epilogue = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 ro = *reinterpret_cast<const uint4*>(s_roff + wm * WM + i * 32 + 8 * q + 4 * kk);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = 4 * q + u;
                    const unsigned rraw = u == 0 ? ro.x : (u == 1 ? ro.y : (u == 2 ? ro.z : ro.w));
                    const float sc = (rraw & OOB) ? 0.f : osc;
                    const unsigned roff = FAST ? rraw + (unsigned)(colb * 4) : rraw;       // (out of range stays out of range)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float v = acc[i][j][e] * sc;
                        // the tile's last two rows were contracted from bytes BEHIND the activation image (the filter image, or - three-
                        // deep ring - an image still in flight): whatever bit pattern that was, it must not reach the sums as 0 x NaN
                        if (i == MT - 1 && e >= 14 && (rraw & OOB)) v = 0.f;
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
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, off, 0, 0);
                        }
                    }
                }
            }
    };
    if (plain && n0 + BN <= d.N) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (d.stats != nullptr) {                 // per-channel (sum, sumsq) of the raw output -> fp64 accumulators [2][N]
        float* const red = reinterpret_cast<float*>(smem + BM * 4);          // [WAVES_M * 2 (lane halves)][2][BN], behind the row table
        static_assert(BM * 4 + WAVES_M * 2 * 2 * BN * 4 <= SMEM_BYTES, "statistics staging must fit the ring");
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
    C3H_TRC(3);
#ifdef SAGEN_TRACE
    if (trc && tid == 0) trc[7] = __builtin_amdgcn_s_memrealtime();
#endif
#undef C3H_TRC
#undef C3H_GTRC
}

template <int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(256, conv3h_wgs_per_cu(BM, BN, KC, 2)) void conv3h_kernel(const IgemmDesc d_in) {
    IgemmDesc d = d_in;
    if (d.grp.G > 1) igemm_relocate(d, (int)blockIdx.z);              // grouped launch (common.h)
    conv3h_body<BM, BN, WM, WN, KC, 2>(d);
}
// ... with the three-deep activation ring
template <int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(256, conv3h_wgs_per_cu(BM, BN, KC, 3)) void conv3hr_kernel(const IgemmDesc d_in) {
    IgemmDesc d = d_in;
    if (d.grp.G > 1) igemm_relocate(d, (int)blockIdx.z);              // grouped launch (common.h)
    conv3h_body<BM, BN, WM, WN, KC, 3>(d);
}

template <int BM, int BN, int WM, int WN, int KC, int AR = 2>
static int launch_conv3h(const IgemmDesc& d, hipStream_t s) {
    if ((d.Cin / 16) % KC) return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: %d channel chunks are not a multiple of %d per group", d.Cin / 16, KC);
    const int per = (cdiv(d.p3_np, BM - 2) + 7) / 8;
    const int grid = 8 * per * cdiv(d.N, BN);
    if constexpr (AR == 3) {
        if (d.splitk != 1) return fail(SAGEN_ERR_UNSUPPORTED, "conv3hr: no dh-split (the three-deep ring needs three groups)");
        hipLaunchKernelGGL((conv3hr_kernel<BM, BN, WM, WN, KC>), dim3(grid, 1, d.grp.G), dim3(256), 0, s, d);
    } else hipLaunchKernelGGL((conv3h_kernel<BM, BN, WM, WN, KC>), dim3(grid, d.splitk, d.grp.G), dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

#ifdef SAGEN_TRACE
static void* g_trace_buf = nullptr;
static int g_trace_nth = -1;
// arms the trace of the nth conv3h launch from now (0 = the next one); buf: 8 x 8 bytes per workgroup, zeroed by the caller
extern "C" void sagen_debug_trace_conv3h(void* buf, int nth) { g_trace_buf = buf; g_trace_nth = nth; }
#endif

int conv3h_dispatch(const IgemmDesc& d_in, IgemmTile tile, hipStream_t s) {
    IgemmDesc d = d_in;
#ifdef SAGEN_TRACE
    if (g_trace_nth >= 0 && g_trace_nth-- == 0) d.trace = g_trace_buf;
#endif
    if (!d.xp3 || d.p3_np <= 0 || d.xp3_fmt != 1) return fail(SAGEN_ERR_NULL, "conv3h: the fp16x2 activation planes are missing");
    if (!d.wh2 || !d.h2_a_inv || !d.h2_w_inv) return fail(SAGEN_ERR_NULL, "conv3h: the fp16x2 filter planes / scales are missing");
    if (d.splitk != 1 && d.splitk != 3) return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: split-K only as the dh-split (3), got %d", d.splitk);
    if (d.splitk == 3) {             // partials [dh][M][N] (dense rows): the reducer applies bias / ReLU / statistics
        if (!d.splitk_ws || d.bias || d.relu_out || d.stats) return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: the dh-split writes raw partials (no bias / ReLU / statistics)");
        d.y = d.splitk_ws;
        d.ldy = d.N;
    }
    if ((long)(d.p3_np + 512) * (d.Win + 1) >= (1L << 32) || ((long)(d.p3_np + 512) / (d.Win + 1) + 1) * d.Hin >= (1L << 32))
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: too many pixels for 32-bit index arithmetic");
    if ((long)d.p3_np * 64 >= (1L << 31) || (long)d.xp3_cstride * (d.Cin / 16) >= (1L << 31) || d.xp3_bytes == 0)
        return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: the activation planes exceed 2 GiB buffer addressing (use a smaller batch)");
    const long y_bytes = ((long)(d.M - 1) * d.ldy + d.N) * 4;
    if (y_bytes >= (1L << 31)) return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: the output exceeds 2 GiB buffer addressing (use a smaller batch)");
    d.y_bytes = (unsigned)y_bytes;
    d.p3_magic_wp = (unsigned)((1UL << 32) / (unsigned)(d.Win + 1)) + 1u;
    d.p3_magic_h = (unsigned)((1UL << 32) / (unsigned)d.Hin) + 1u;
    switch (tile) {
        case TILE_P3H_128x64: return launch_conv3h<128, 64, 64, 32, 1>(d, s);
        case TILE_P3H_128x128: return launch_conv3h<128, 128, 64, 64, 1>(d, s);
        case TILE_P3H_64x64: return launch_conv3h<64, 64, 32, 32, 1>(d, s);
        case TILE_P3H_256x64: return launch_conv3h<256, 64, 64, 64, 1>(d, s);
        case TILE_P3H_128x64_C2: return launch_conv3h<128, 64, 64, 32, 2>(d, s);
        case TILE_P3H_64x64_C2: return launch_conv3h<64, 64, 32, 32, 2>(d, s);
        case TILE_P3H_64x64_C4: return launch_conv3h<64, 64, 32, 32, 4>(d, s);
        case TILE_P3HR_256x64: return launch_conv3h<256, 64, 64, 64, 1, 3>(d, s);
        case TILE_P3HR_128x64: return launch_conv3h<128, 64, 64, 32, 1, 3>(d, s);
        case TILE_P3HR_64x64_C2: return launch_conv3h<64, 64, 32, 32, 2, 3>(d, s);
        case TILE_P3HR_128x128: return launch_conv3h<128, 128, 64, 64, 1, 3>(d, s);
        default: return fail(SAGEN_ERR_UNSUPPORTED, "conv3h: bad tile id %d", (int)tile);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// filter planes for conv3h_kernel: fp32 packed filter [N][Kpad] -> max |w| -> 2^kw -> fp16 (hi, lo) of w 2^kw, [Kpad/16][2][N][16]
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h2_absmax_kernel(const float* __restrict__ wp, long total, unsigned* __restrict__ amax_bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(wp[i]));
    __shared__ unsigned s_m;
    if (threadIdx.x == 0) s_m = 0u;
    __syncthreads();
    atomicMax(&s_m, __builtin_bit_cast(unsigned, m));            // non-negative floats order like their bit patterns
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax_bits, s_m);
}

// 2^k with bound * 2^k in [512, 1024) (k from the exponent field; bound == 0 or not finite -> 1)
__device__ __forceinline__ float h2_scale_for(float bound) {
    const unsigned b = __builtin_bit_cast(unsigned, bound);
    const int e = (int)((b >> 23) & 0xff);
    if (e == 0 || e == 255) return 1.f;
    int k = 127 + 9 - e;                                         // bound in [2^(e-127), 2^(e-126)) -> bound 2^k in [512, 1024)
    k = max(-60, min(60, k));
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

__global__ __launch_bounds__(256) void h2_filter_pack_kernel(const float* __restrict__ wp, long total, int N, int Kpad,
                                                             const unsigned* __restrict__ amax_bits, _Float16* __restrict__ w2,
                                                             float* __restrict__ w_inv) {
    const float sc = h2_scale_for(__builtin_bit_cast(float, amax_bits[0]));
    if (blockIdx.x == 0 && threadIdx.x == 0) w_inv[0] = 1.f / sc;                // exact: a power of two
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long n = idx / Kpad;
    const int k = (int)(idx - n * Kpad);
    const float v = wp[idx] * sc;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const long o = ((long)(k >> 4) * 2 * N + n) * 16 + (k & 15);          // plane 0 of K tile k/16
    w2[o] = h;
    w2[o + (long)N * 16] = l;
}

// wp: fp32 packed filter [N][Kpad]; w2: N*Kpad*2 fp16; scratch: one unsigned (zeroed here); w_inv: where 2^-kw is written
int h2_filter_pack_launch(const float* wp, int N, int Kpad, void* w2, unsigned* scratch, float* w_inv, hipStream_t s) {
    if (!wp || !w2 || !scratch || !w_inv) return fail(SAGEN_ERR_NULL, "h2_filter_pack: null argument");
    const long total = (long)N * Kpad;
    SAGEN_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(h2_absmax_kernel, dim3((int)std::min<long>(cdiv(total, 256), 1024)), dim3(256), 0, s, wp, total, scratch);
    hipLaunchKernelGGL(h2_filter_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, wp, total, N, Kpad, scratch,
                       reinterpret_cast<_Float16*>(w2), w_inv);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}


// ---- the same for MANY layers in two launches (bind; and once per training step, after the optimiser rewrote the variables) ----
// max |w| per job WITHOUT atomics: one partial per workgroup, then one workgroup per job folds its partials (11 k workgroups each
// ending in a same-address atomicMax took 109 us for 44 MB in the training step's trace - the L2 serialises them; an atomic-load
// guard in front of the atomicMax made it 188 us)
__global__ __launch_bounds__(256) void h2_absmax_multi_kernel(const H2Job* __restrict__ jobs, int njobs, float* __restrict__ part) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
    const H2Job job = jobs[j];
    const unsigned total = (unsigned)job.N * (unsigned)job.Kpad;            // (a multiple of 16)
    const unsigned i = ((unsigned)((int)blockIdx.x - job.first_block) * 256 + threadIdx.x) * 4;
    float m = 0.f;
    if (i < total) {
        const float4 v = *reinterpret_cast<const float4*>(job.wp + i);
        m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max_f(m);
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}
__global__ __launch_bounds__(256) void h2_absmax_finish_kernel(const H2Job* __restrict__ jobs, int njobs, int nblocks, const float* __restrict__ part,
                                                               unsigned* __restrict__ amax) {
    const int j = blockIdx.x;
    const int b0 = jobs[j].first_block, b1 = j + 1 < njobs ? jobs[j + 1].first_block : nblocks;
    float m = 0.f;
    for (int b = b0 + threadIdx.x; b < b1; b += 256) m = fmaxf(m, part[b]);
    m = wave_max_f(m);
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax[j] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
}

__global__ __launch_bounds__(256) void h2_filter_pack_multi_kernel(const H2Job* __restrict__ jobs, int njobs, const unsigned* __restrict__ amax) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
    const H2Job job = jobs[j];
    const float sc = h2_scale_for(__builtin_bit_cast(float, amax[j]));
    if ((int)blockIdx.x == job.first_block && threadIdx.x == 0) job.w_inv[0] = 1.f / sc;
    const long total = (long)job.N * job.Kpad;
    const long i0 = ((long)blockIdx.x - job.first_block) * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long idx = i0 + 256 * k;
        if (idx >= total) continue;
        const long n = idx / job.Kpad;
        const int kk = (int)(idx - n * job.Kpad);
        const float v = job.wp[idx] * sc;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        const long o = ((long)(kk >> 4) * 2 * job.N + n) * 16 + (kk & 15);
        reinterpret_cast<_Float16*>(job.w2)[o] = h;
        reinterpret_cast<_Float16*>(job.w2)[o + (long)job.N * 16] = l;
    }
}

int h2_filter_pack_multi_launch(const H2Job* jobs_dev, int njobs, int nblocks, unsigned* amax, hipStream_t s) {
    if (njobs <= 0) return SAGEN_OK;
    if (!jobs_dev || !amax) return fail(SAGEN_ERR_NULL, "h2_filter_pack_multi: null argument");
    float* part = reinterpret_cast<float*>(amax + njobs);        // amax holds njobs + nblocks words
    hipLaunchKernelGGL(h2_absmax_multi_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, njobs, part);
    hipLaunchKernelGGL(h2_absmax_finish_kernel, dim3(njobs), dim3(256), 0, s, jobs_dev, njobs, nblocks, (const float*)part, amax);
    hipLaunchKernelGGL(h2_filter_pack_multi_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, njobs, amax);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
