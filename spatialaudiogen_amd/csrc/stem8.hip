// stem8pool_kernel: the ResNet18 stem for frames that arrive as the uint8 the JPEG decoder produced (sagen_forward_u8) -
// 7x7 stride-2 SAME convolution (resnet.py:133) + the batch statistics of its raw output + the 3x3 stride-2 SAME max-pool
// (resnet.py:135), one kernel, with HALF the matrix work of the float-frame stem (stempool.hip / igemm3s2.hip).
//
// Why half: the feeder normalises a decoded frame as x = u / 255 - 0.5 (myutils.py:88-89), u in {0..255}.  With u' = u - 128,
//     x = (u' + 0.5) / 255      and      conv(W, x) = conv(W, u') / 255 + (0.5 / 255) * sum(W),
// where the zero padding of x is the value u' = -0.5.  Every u' (and -0.5) is EXACTLY representable in bf16 (8 significant bits), so
// the activation operand of the contraction is ONE bf16 plane instead of the three planes (hi, mid, lo) a general fp32 value needs,
// and the fp32-equivalent product is three bf16 MFMA products (u' x W_hi, u' x W_mid, u' x W_lo) instead of six.  Nothing is
// approximated: the operand is exact, the filter keeps its full three-plane split, accumulation is fp32 - the result differs from the
// float-frame kernels only by the rounding of x itself (the float path convolves the ROUNDED float32 x) and of the final scale.
// The operand split disappears with it (the fragment of a lane is a 16-byte load of the plane: no VALU conversion at all).
//
// Structure: persistent workgroups of 8 waves, each owning HALF of the 64 output channels - its half of the filter (three planes,
// 43 KB) stays LDS-resident and the raw tile staging is 32 KB, so TWO workgroups share a CU (78 KB each): while one pools its patch
// the other multiplies (the first version - all 64 channels, 150 KB, one workgroup per CU - kept the matrix pipe 27 % busy: 133 us);
// the batch statistics come straight from the accumulators (a 16-bit ownership mask per lane), not from LDS.  A patch is
// 8 x 7 pooled pixels = 17 x 15 raw outputs (one halo row / column recomputed) = 255 of the 256 rows of 8 MFMA row tiles, one per
// wave; K = 7 x 8 x 4 = 224 (seven taps down; seven + one zero tap across; three + one zero channel): 14 K steps of 16; K step ks of
// raw pixel (r, c) is the 8 CONTIGUOUS bf16 of plane row 2r + ks/2 starting at pixel 2c + 4 (ks & 1) + 2g.  The 39 x 36-pixel patch
// of the plane a tile reads (11 KB) is staged through LDS - prefetched into registers a patch ahead, written once, fragments by
// ds_read_b128 - in the space the raw tile occupies afterwards; raw tile -> LDS once, pooled with max or min per channel by the
// sign of gamma (relu(bn(.)) is monotone per channel: stempool.hip), statistics over the 16 x 14 pixels the patch owns.
#include "igemm3_common.h"
#include "h2_planes.h"

namespace sagen {

constexpr int S8_PH = 8, S8_PW = 7;            // pooled rows / cols per patch
constexpr int S8_RH = 2 * S8_PH + 1;           // raw rows per patch (17)
constexpr int S8_RW = 2 * S8_PW + 1;           // raw cols per patch (15)
constexpr int S8_M = S8_RH * S8_RW;            // 255
// where raw pixel m of a patch sits in the staged raw tile (128 B per pixel): the pool reads pixels two apart (256 B = all 64 banks
// once round) from eight lanes each, so a 16-lane ds_read_b128 group met the same 32 banks twice - every second pixel PAIR swaps its
// two members and the group covers all 64 banks (rocprofv3: 7.0e6 bank-conflict cycles per launch = a third of the kernel's LDS cycles)
__device__ __forceinline__ int s8_slot(int m) { return m ^ ((m >> 1) & 1); }
constexpr int S8_UH = 229, S8_UW = 456;        // plane geometry: 2 + 224 + 3 rows, 2 + 448 + 6 pixels per row (row pitch 3648 B = 16 * 228)
constexpr int S8_NH = 32;                      // output channels per workgroup: the two halves of the 64 run as separate workgroups
constexpr int S8_W_BYTES = 14 * 3 * S8_NH * 32;   // this half's filter planes [K/16][plane][n][16] bf16 (43 KB; the fp16x2 variant: two planes, 29 KB)
constexpr int S8_CT_BYTES = 256 * S8_NH * 4;   // raw tile [256][32] fp32
constexpr int S8_THREADS = 512;
constexpr int S8_LDS = S8_W_BYTES + S8_CT_BYTES + 2 * 8 * S8_NH * 4 + S8_NH * 4;   // 77.9 KB: TWO workgroups per CU - one pools while the other multiplies
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));

size_t stem8_plane_bytes(int B) { return (size_t)B * S8_UH * S8_UW * 8; }

// uint8 frames [B,224,448,3] -> the centred plane [B,229,456,4] bf16: u - 128 inside, -0.5 (= x 0) in the border, 0 in channel 3
template <bool HALF>      // HALF: the plane is fp16 (stem8pool_kernel MODE 2) instead of bf16 - the values are exact in both
__global__ __launch_bounds__(256) void stem8_prep_kernel(const unsigned char* __restrict__ x_, u32x2* __restrict__ plane_, int B,
                                                         float* __restrict__ zero_ptr_, long zero_n, const GroupInfo gi) {
    // grouped launch (common.h): the caller's frames hold the groups back to back; the plane and the accumulators are per group
    const unsigned char* __restrict__ x = x_ + (size_t)blockIdx.z * ((size_t)B * 224 * 448 * 3);
    u32x2* __restrict__ plane = SAGEN_GRP(plane_);
    float* __restrict__ zero_ptr = SAGEN_GRP(zero_ptr_);
    const long total = (long)B * S8_UH * S8_UW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long)gridDim.x * 256) zero_ptr[i] = 0.f;     // (instead of a fill launch)
    auto conv = [](unsigned u) -> unsigned {                                  // byte -> u - 128 as fp16 / bf16 bits (exact: |u - 128| <= 128)
        return HALF ? (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(float)((int)u - 128))
                    : __builtin_bit_cast(unsigned, (float)((int)u - 128)) >> 16;
    };
    // FOUR plane pixels per thread (round 6; one pixel per thread was three byte loads and one 8-byte store per lane: 17 us per batch for
    // 23 MB).  The plane row is 456 pixels = 114 quads; quad q covers frame columns 4q - 2 .. 4q + 1: its 12 source bytes start 6 bytes
    // into a 12-byte group, always 2 bytes past a dword boundary: an interior quad is the aligned 16-byte window around them (four dword
    // loads) and static byte extracts; the quads that touch the border (q = 0, q >= 112) and rows outside the frame go pixel by pixel.
    constexpr int QW = S8_UW / 4;                                             // 114
    static_assert(S8_UW % 4 == 0, "plane rows are whole quads");
    const long nquad = (long)B * S8_UH * QW;
    const unsigned border = HALF ? 0xB800u : 0xBF00u;                         // -0.5 as fp16 / bf16
    for (long iq = (long)blockIdx.x * 256 + threadIdx.x; iq < nquad; iq += (long)gridDim.x * 256) {
        long p = iq;
        const int q = (int)(p % QW); p /= QW;
        const int h = (int)(p % S8_UH) - 2;
        const int b = (int)(p / S8_UH);
        const int w0 = 4 * q - 2;
        u32x2 o[4];
        if ((unsigned)h < 224u && w0 >= 0 && w0 + 3 < 448) {
            // first of the quad's 12 source bytes: (row * 448 + 4q - 2) * 3 = 2 (mod 4) for every row and quad (1344 and 12 are multiples of 4)
            const long ob = (((long)b * 224 + h) * 448 + w0) * 3;
            constexpr unsigned sh = 2;
            const unsigned* src = reinterpret_cast<const unsigned*>(x + (ob - sh));          // (4-byte aligned frames: checked by the launcher)
            const unsigned d[4] = {src[0], src[1], src[2], src[3]};
            unsigned by[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) by[k] = (d[(sh + k) >> 2] >> (8 * ((sh + k) & 3))) & 0xffu;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = u32x2{conv(by[3 * k]) | (conv(by[3 * k + 1]) << 16), conv(by[3 * k + 2])};
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = w0 + k;
                unsigned c0 = border, c1 = border, c2 = border;
                if ((unsigned)h < 224u && (unsigned)w < 448u) {
                    const unsigned char* src = x + (((long)b * 224 + h) * 448 + w) * 3;
                    c0 = conv(src[0]); c1 = conv(src[1]); c2 = conv(src[2]);
                }
                o[k] = u32x2{c0 | (c1 << 16), c2};
            }
        }
        u32x4* dst = reinterpret_cast<u32x4*>(plane + iq * 4);               // 32 bytes per quad (the plane is 256-byte aligned, rows are whole quads)
        dst[0] = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
        dst[1] = u32x4{o[2][0], o[2][1], o[3][0], o[3][1]};
    }
    (void)total;
}

// RAW: no pool - the owned 16 x 14 raw outputs of every patch go to y0 [B,112,224,64] (the training step keeps the raw stem output for
// the batch-norm backward and pools it in the pass that also writes the planes of the pooled tensor)
// F16 (round 5): FLOAT frames (the flow encoder's input; video handed over as float32) on the same structure - the frame as TWO fp16
// planes of x * 2^ka (ka from the exact maximum of the batch: stem16_prep), the filter as two fp16 planes of w * 2^kw: three products
// per multiply on v_mfma_f32_32x32x16_f16 like conv3h_kernel, zero padding is a zero in the planes, the tile is scaled back by
// 2^-(ka + kw).  Replaces igemm3s2_kernel (in-loop bf16x3 split, six products, 155 us) + the separate pool pass for such frames.
// MODE 0: uint8 frames, one bf16 plane of u - 128 x three bf16 filter planes (round 4; the training step's raw variant)
// MODE 1: float frames, two fp16 planes x two fp16 filter planes (three products)
// MODE 2: uint8 frames, ONE fp16 plane of u - 128 (|u - 128| <= 128 and -0.5 are exact in fp16 as well) x two fp16 filter planes of
//         w * 2^kw: TWO products per multiply instead of three - the operand is exact, the filter keeps 22 bits + its residual's sign
// OUT 0: pooled raw output; 1: the raw output itself (into `pooled`); 2: both - raw into `raw2`, pooled raw into `pooled` (the training
// step: the backward keeps the raw tensor, and the pool no longer re-reads its 205 MB in a pass of its own)
template <int OUT, int MODE = 0>
__global__ __launch_bounds__(S8_THREADS, 2) void stem8pool_kernel(const char* __restrict__ plane_, const float* __restrict__ wf32,
                                                                  const char* __restrict__ wplanes, const float* __restrict__ gamma,
                                                                  float* __restrict__ pooled_, double* __restrict__ stats_, int B,
                                                                  long plane_stride, const float* __restrict__ a_inv_, const float* __restrict__ w_inv_,
                                                                  float* __restrict__ raw2_, const GroupInfo gi) {
    const char* __restrict__ plane = SAGEN_GRP(plane_);               // grouped launch (common.h): group = blockIdx.z
    float* __restrict__ pooled = SAGEN_GRP(pooled_);
    double* __restrict__ stats = SAGEN_GRP(stats_);
    const float* __restrict__ a_inv = SAGEN_GRP(a_inv_);
    const float* __restrict__ w_inv = SAGEN_GRP(w_inv_);
    float* __restrict__ raw2 = SAGEN_GRP(raw2_);
    constexpr bool F16 = MODE != 0;
    constexpr int NPLA = MODE == 1 ? 2 : 1, NPLW = MODE == 0 ? 3 : 2;             // operand planes: activation, filter
    constexpr int W_BYTES = 14 * NPLW * S8_NH * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;
    char* const pa = smem + S8_W_BYTES;                                                                 // the patch of the plane(s) (K loop) ...
    float* const ct = reinterpret_cast<float*>(smem + S8_W_BYTES);                                      // ... and the raw tile (epilogue) share this space
    float* const red = reinterpret_cast<float*>(smem + S8_W_BYTES + S8_CT_BYTES);                       // [2][8][32]
    float* const cb = reinterpret_cast<float*>(smem + S8_W_BYTES + S8_CT_BYTES + 2 * 8 * S8_NH * 4);    // [32]: (0.5 / 255) * sum_k W[n][k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const int nh = blockIdx.x & 1;                   // which 32 of the 64 channels
    const int wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;

    // this half's filter planes, once: global row ((ks*3 + pl)*64 + nh*32 + n) -> LDS row ((ks*3 + pl)*32 + n), 32 B each
    for (int i = tid; i < W_BYTES / 16; i += S8_THREADS) {
        const int row = i >> 1, hf = i & 1;
        const int kp = row >> 5, n = row & 31;
        // (the two 16-byte halves of a 32-byte row swapped in rows 8-15 / 24-31: lanes li and li + 8 of a fragment read would
        //  otherwise hit the same banks - 49 % of the LDS cycles were conflict cycles, profiles/r04_pmc_per_launch.json)
        reinterpret_cast<f32x4*>(wl)[i] = *reinterpret_cast<const f32x4*>(wplanes + ((long)(kp * 64 + nh * 32 + n) * 32 + 16 * (hf ^ ((n >> 3) & 1))));
    }
    if (tid < S8_NH) {
        double s = 0.0;
        if (MODE != 1)
            for (int k = 0; k < 224; ++k) s += (double)wf32[(nh * S8_NH + tid) * 224 + k];
        cb[tid] = (float)(s * (0.5 / 255.0));       // (float frames: zero padding IS zero, no constant term)
    }
    const float osc = MODE == 1 ? a_inv[0] * w_inv[0] : (MODE == 2 ? w_inv[0] * (1.f / 255.f) : 1.f / 255.f);
    const int pch4 = tid & 7;                        // pooling: this thread's 4 channels, max or min per channel
    bool use_min[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) use_min[k] = OUT == 1 ? false : gamma[nh * S8_NH + 4 * pch4 + k] < 0.f;
    // Round 6 (the kernel's way out was VALU-bound: 11 VALU per MFMA, matrix pipe 39 % busy).  Inference (OUT 0) stages the raw tile
    // SIGN-FLIPPED where gamma < 0: -fma(a, s, t) = fma(a, -s, -t) exactly, min(a, b) = -max(-a, -b) exactly, so the pool is a plain
    // max (one instruction per element instead of min + max + select) and the pooled value is flipped back once; the channel sums
    // are taken of the flipped values and the sum's sign restored at the end (the squares do not care).  Bit-identical results.
    constexpr bool SGN = OUT == 0;
    float sgn4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sgn4[k] = (SGN && use_min[k]) ? -1.f : 1.f;
    const float sgn_l = (SGN && gamma[nh * S8_NH + li] < 0.f) ? -1.f : 1.f;      // of this lane's channel (epilogue)
    // ... and the nine LDS offsets of a pooling thread's window are the same for every patch that does not touch the image's last row /
    // column (pr < 6 and pc < 15): computed once
    int poff[9];
    {
        const int p = tid >> 3;
        const int pr_l = p / S8_PW, pc_l = p - pr_l * S8_PW;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr)
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) poff[dr * 3 + dc] = s8_slot((2 * pr_l + dr) * S8_RW + 2 * pc_l + dc) * S8_NH + 4 * pch4;
    }
    // statistics straight from the accumulators: which of this lane's 16 tile rows are pixels the patch OWNS (16 x 14 of the 17 x 15;
    // the halo row / column belongs to the neighbour) - the same for every patch
    unsigned own = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        const int r = m / S8_RW, c = m - r * S8_RW;
        if (m < S8_M && r < 2 * S8_PH && c < 2 * S8_PW) own |= 1u << e;
    }
    float ssum = 0.f, ssq = 0.f;
    __syncthreads();
    const float cbl = cb[li] * sgn_l;
    const float osc_l = osc * sgn_l;

    // The patch of the plane a tile needs - 39 rows x 36 pixels x 8 B = 11 KB - goes through LDS: every input pixel is fetched from
    // global memory ONCE per workgroup (each is used by ~10 (row, tap) pairs: with per-lane fragment loads straight from global
    // memory the kernel was bound by the vector-memory pipe, 133 / 95 us), as 702 chunks of 16 B, one or two per thread, prefetched
    // into registers a whole patch ahead.
    const int npatch = B * 7 * 16;
    constexpr int PROWS = 2 * (S8_RH - 1) + 7, PCH = (2 * (S8_RW - 1) + 8) / 2;     // 39 rows x 18 chunks (36 pixels)
    constexpr int NCHUNK = PROWS * PCH;                                             // 702
    constexpr int PATCH_BYTES = NCHUNK * 16;
    static_assert(NCHUNK <= 2 * S8_THREADS && NPLA * PATCH_BYTES <= S8_CT_BYTES, "patch staging");
    const int ck0 = tid, ck1 = tid + S8_THREADS;
    const int cr0 = ck0 / PCH, cc0 = ck0 - cr0 * PCH, cr1 = ck1 / PCH, cc1 = ck1 - cr1 * PCH;
    auto patch_src = [&](int patch, int crow, int ccol) {
        const int b = patch / 112, rem = patch - b * 112;
        const int pr = rem >> 4, pc = rem & 15;
        const int row = min(2 * 16 * pr + crow, S8_UH - 1);       // (rows past the plane only feed the raw row below the image, which is never pooled)
        return plane + (((long)b * S8_UH + row) * S8_UW + 2 * 14 * pc) * 8 + ccol * 16;
    };
    f32x4 ld0[NPLA], ld1[NPLA];
#pragma unroll
    for (int p = 0; p < NPLA; ++p) {
        ld0[p] = f32x4{0.f, 0.f, 0.f, 0.f}; ld1[p] = ld0[p];
        if (wg < npatch) {
            ld0[p] = *reinterpret_cast<const f32x4*>(patch_src(wg, cr0, cc0) + p * plane_stride);
            if (ck1 < NCHUNK) ld1[p] = *reinterpret_cast<const f32x4*>(patch_src(wg, cr1, cc1) + p * plane_stride);
        }
    }
    // this lane's raw pixel (row `li` of the wave's MFMA tile): byte offset of its K-step-0 fragment inside the LDS patch
    int aoff;
    {
        const int m = wave * 32 + li;
        int r = m / S8_RW, c = m - r * S8_RW;
        if (m >= S8_M) { r = 0; c = 0; }                          // the 256th row: any valid address, dropped later
        aoff = ((2 * r) * (2 * PCH) + 2 * c + 2 * g) * 8;
    }
    for (int patch = wg; patch < npatch; patch += nwg) {
        const int b = patch / 112, rem = patch - b * 112;
        const int pr = rem >> 4, pc = rem & 15;
        const int R0 = 16 * pr, C0 = 14 * pc;
#pragma unroll
        for (int p = 0; p < NPLA; ++p) {
            reinterpret_cast<f32x4*>(pa + p * PATCH_BYTES)[ck0] = ld0[p];
            if (ck1 < NCHUNK) reinterpret_cast<f32x4*>(pa + p * PATCH_BYTES)[ck1] = ld1[p];
        }
        __syncthreads();
        const int next_patch = patch + nwg;
        if (next_patch < npatch) {                                // flies under this patch's K loop and epilogue
#pragma unroll
            for (int p = 0; p < NPLA; ++p) {
                ld0[p] = *reinterpret_cast<const f32x4*>(patch_src(next_patch, cr0, cc0) + p * plane_stride);
                if (ck1 < NCHUNK) ld1[p] = *reinterpret_cast<const f32x4*>(patch_src(next_patch, cr1, cc1) + p * plane_stride);
            }
        }
        f32x16 acc[2];                               // no accumulator twice in a row
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            // K step ks of raw pixel (r, c): the 8 bf16 at patch row 2r + ks/2, pixels 2c + 4 (ks & 1) + 2g, +1
            bf16x8 fa[NPLA], fb[NPLW];
#pragma unroll
            for (int p = 0; p < NPLA; ++p) fa[p] = *reinterpret_cast<const bf16x8*>(pa + p * PATCH_BYTES + aoff + ((ks >> 1) * (2 * PCH) + (ks & 1) * 4) * 8);
#pragma unroll
            for (int pl = 0; pl < NPLW; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(wl + ((ks * NPLW + pl) * S8_NH + li) * 32 + 16 * (g ^ ((li >> 3) & 1)));
            const int a = ks & 1;                    // u' x W_lo, x W_mid, x W_hi on accumulators a, a^1, a | a^1, a, a^1 | ...
            if constexpr (MODE == 2) {               // u' x W_lo, u' x W_hi: two products, alternating accumulators
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fa[0]), __builtin_bit_cast(f16x8s, fb[1]), acc[a], 0, 0, 0);
                acc[a ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fa[0]), __builtin_bit_cast(f16x8s, fb[0]), acc[a ^ 1], 0, 0, 0);
            } else if constexpr (F16) {              // lo x hi, hi x lo, hi x hi (conv3h.hip)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fa[NPLA - 1]), __builtin_bit_cast(f16x8s, fb[0]), acc[a], 0, 0, 0);
                acc[a ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fa[0]), __builtin_bit_cast(f16x8s, fb[NPLW - 1]), acc[a ^ 1], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fa[0]), __builtin_bit_cast(f16x8s, fb[0]), acc[a], 0, 0, 0);
            } else {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[NPLW - 1], acc[a], 0, 0, 0);
                acc[a ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[a ^ 1], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[a], 0, 0, 0);
            }
        }
        __syncthreads();                       // every wave is done with the patch: its space becomes the raw tile
        // raw output = acc / 255 + (0.5 / 255) sum(W) -> LDS [pixel][32] + this lane's share of the statistics.
        // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = fmaf(acc[0][e] + acc[1][e], osc_l, cbl);       // (sign-flipped where gamma < 0 at inference, see above)
            ct[s8_slot(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g) * S8_NH + li] = v;
            const float vo = ((own >> e) & 1u) ? v : 0.f;
            ssum += vo;
            ssq = fmaf(vo, vo, ssq);
        }
        __syncthreads();
        if (OUT >= 1) {
            // the patch's own pixels, 128 contiguous bytes (this half's 32 channels) per pixel
            float* const raw = OUT == 2 ? raw2 : pooled;
            for (int it = tid; it < 2 * S8_PH * 2 * S8_PW * 8; it += S8_THREADS) {
                const int p = it >> 3, q4 = it & 7;
                const int r = p / (2 * S8_PW), cc = p - r * (2 * S8_PW);
                *reinterpret_cast<float4*>(raw + (((long)b * 112 + R0 + r) * 224 + C0 + cc) * 64 + nh * S8_NH + 4 * q4) =
                    *reinterpret_cast<const float4*>(ct + s8_slot(r * S8_RW + cc) * S8_NH + 4 * q4);
            }
        }
        // pool 3x3 / 2 (TF SAME: nothing before, one row / column after -> clipped at the image edge): one item per thread
        if (OUT != 1 && tid < S8_PH * S8_PW * 8) {
            const int p = tid >> 3;
            const int pr_l = p / S8_PW, pc_l = p - pr_l * S8_PW;
            float4 v[9];
            if (pr < 6 && pc < 15) {                 // (uniform) interior patch: the precomputed window
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] = *reinterpret_cast<const float4*>(ct + poff[k]);
            } else {
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        // (rows / columns past the image edge are clamped to the window's first row / column: max / min is unchanged by a duplicate)
                        const int rr = (R0 + 2 * pr_l + dr >= 112) ? 2 * pr_l : 2 * pr_l + dr;
                        const int cc = (C0 + 2 * pc_l + dc >= 224) ? 2 * pc_l : 2 * pc_l + dc;
                        v[dr * 3 + dc] = *reinterpret_cast<const float4*>(ct + s8_slot(rr * S8_RW + cc) * S8_NH + 4 * pch4);
                    }
            }
            float4 ext = v[0];
            if constexpr (SGN) {                     // the staged values are flipped where the pool is a min: a plain max, flipped back
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    ext.x = fmaxf(ext.x, v[k].x); ext.y = fmaxf(ext.y, v[k].y); ext.z = fmaxf(ext.z, v[k].z); ext.w = fmaxf(ext.w, v[k].w);
                }
                ext.x *= sgn4[0]; ext.y *= sgn4[1]; ext.z *= sgn4[2]; ext.w *= sgn4[3];
            } else {
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    ext.x = use_min[0] ? fminf(ext.x, v[k].x) : fmaxf(ext.x, v[k].x); ext.y = use_min[1] ? fminf(ext.y, v[k].y) : fmaxf(ext.y, v[k].y);
                    ext.z = use_min[2] ? fminf(ext.z, v[k].z) : fmaxf(ext.z, v[k].z); ext.w = use_min[3] ? fminf(ext.w, v[k].w) : fmaxf(ext.w, v[k].w);
                }
            }
            *reinterpret_cast<float4*>(pooled + (((long)b * 56 + 8 * pr + pr_l) * 112 + 7 * pc + pc_l) * 64 + nh * S8_NH + 4 * pch4) = ext;
        }
        __syncthreads();                       // the raw tile is overwritten by the next patch
    }
    // per-channel (sum, sumsq): the two lane halves of a wave hold the same channel, then the eight waves
    ssum *= sgn_l;                                   // (the sums were taken of the sign-flipped values)
    ssum = wave_xor_add<32>(ssum);
    ssq = wave_xor_add<32>(ssq);
    if (g == 0) {
        red[(0 * 8 + wave) * S8_NH + li] = ssum;
        red[(1 * 8 + wave) * S8_NH + li] = ssq;
    }
    __syncthreads();
    if (tid < 2 * S8_NH) {
        const int which = tid >> 5, ch = tid & 31;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[(which * 8 + k) * S8_NH + ch];
        atomicAdd(&stats[which * 64 + nh * S8_NH + ch], (double)s);
    }
}

int stem8_prep_launch(const unsigned char* x, void* plane, int B, hipStream_t s, float* zero_ptr, long zero_n, int half) {
    if (!x || !plane) return fail(SAGEN_ERR_NULL, "stem8_prep: null argument");
    if (((uintptr_t)x % 4) || ((uintptr_t)plane % 16)) return fail(SAGEN_ERR_UNSUPPORTED, "stem8_prep: the frames must be 4-byte aligned, the plane 16-byte aligned");
    const long total = (long)B * S8_UH * S8_UW / 4;       // one thread per four plane pixels
    const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 16);
    const GroupInfo gi = cur_group();
    if (half) hipLaunchKernelGGL(stem8_prep_kernel<true>, dim3(grid, 1, gi.G), dim3(256), 0, s, x, reinterpret_cast<u32x2*>(plane), B, zero_ptr, zero_ptr ? zero_n : 0L, gi);
    else hipLaunchKernelGGL(stem8_prep_kernel<false>, dim3(grid, 1, gi.G), dim3(256), 0, s, x, reinterpret_cast<u32x2*>(plane), B, zero_ptr, zero_ptr ? zero_n : 0L, gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// plane: stem8_prep's output; wp: the stem's packed filter (fp32 [64][224] followed by its bf16x3 planes); pooled [B,56,112,64] RAW
// (max or min per channel by the sign of gamma: BN + ReLU follow on the pooled tensor); stats: fp64 (sum, sumsq) of the raw output
int stem8pool_launch(const void* plane, const float* wp, const float* gamma, float* pooled, double* stats, int B, hipStream_t s) {
    if (!plane || !wp || !gamma || !pooled || !stats) return fail(SAGEN_ERR_NULL, "stem8pool: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const char* planes = reinterpret_cast<const char*>(wp + 64 * 224);
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL(stem8pool_kernel<0>, dim3(2 * std::min(npatch, 256), 1, cur_group().G), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(plane), wp, planes, gamma,
                       pooled, stats, B, 0L, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, cur_group());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// MODE 2: plane = stem8_prep's fp16 plane (half = 1); wp = the packed fp32 filter (for the constant term), wh2 = its two fp16 planes, w_inv = 2^-kw
int stem8pool_h2_launch(const void* plane, const float* wp, const void* wh2, const float* w_inv, const float* gamma, float* pooled, double* stats, int B,
                        hipStream_t s) {
    if (!plane || !wp || !wh2 || !w_inv || !gamma || !pooled || !stats) return fail(SAGEN_ERR_NULL, "stem8pool_h2: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL((stem8pool_kernel<0, 2>), dim3(2 * std::min(npatch, 256), 1, cur_group().G), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(plane), wp,
                       reinterpret_cast<const char*>(wh2), gamma, pooled, stats, B, 0L, (const float*)nullptr, w_inv, (float*)nullptr, cur_group());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// the same contraction without the pool: y0 [B,112,224,64] = the raw stem output (training step), statistics as above
int stem8raw_launch(const void* plane, const float* wp, float* y0, double* stats, int B, hipStream_t s) {
    if (!plane || !wp || !y0 || !stats) return fail(SAGEN_ERR_NULL, "stem8raw: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const char* planes = reinterpret_cast<const char*>(wp + 64 * 224);
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL(stem8pool_kernel<1>, dim3(2 * std::min(npatch, 256), 1, cur_group().G), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(plane), wp, planes,
                       (const float*)nullptr, y0, stats, B, 0L, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, cur_group());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// ... with the pool as well: y0 = the raw stem output, pooled [B,56,112,64] = its 3x3/2 max (min where gamma < 0), one kernel
int stem8rawpool_launch(const void* plane, const float* wp, const float* gamma, float* y0, float* pooled, double* stats, int B, hipStream_t s) {
    if (!plane || !wp || !gamma || !y0 || !pooled || !stats) return fail(SAGEN_ERR_NULL, "stem8rawpool: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const char* planes = reinterpret_cast<const char*>(wp + 64 * 224);
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL(stem8pool_kernel<2>, dim3(2 * std::min(npatch, 256), 1, cur_group().G), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(plane), wp, planes,
                       gamma, pooled, stats, B, 0L, (const float*)nullptr, (const float*)nullptr, y0, cur_group());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// ---- float frames (F16 variant): the batch's exact maximum, then the two fp16 planes ----
constexpr int S16_PARTS = 1024;                   // per-workgroup partial maxima (no atomics, nothing to clear)
__global__ __launch_bounds__(256) void stem16_amax_kernel(const float* __restrict__ x_, long n4, float* __restrict__ part_, float* __restrict__ zero_ptr_, long zero_n, const GroupInfo gi) {
    const float* __restrict__ x = x_ + (size_t)blockIdx.z * (size_t)n4 * 4;        // grouped launch: the caller's frames hold the groups back to back
    float* __restrict__ part = SAGEN_GRP(part_);
    float* __restrict__ zero_ptr = SAGEN_GRP(zero_ptr_);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long)gridDim.x * 256) zero_ptr[i] = 0.f;     // (the trunk's batch-norm accumulators)
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    m = wave_max_f(m);
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}
// float frames [B,224,448,3] -> planes hi / lo [B,229,456,4] fp16 of x * 2^ka (max |x| * 2^ka in [512, 1024): exact bound), zero border,
// zero channel 3; 2^-ka -> a_inv[0]
__global__ __launch_bounds__(256) void stem16_prep_kernel(const float* __restrict__ x_, u32x2* __restrict__ hi_, u32x2* __restrict__ lo_, int B,
                                                          const float* __restrict__ part_, float* __restrict__ a_inv_, const GroupInfo gi) {
    const float* __restrict__ x = x_ + (size_t)blockIdx.z * ((size_t)B * 224 * 448 * 3);
    u32x2* __restrict__ hi = SAGEN_GRP(hi_);
    u32x2* __restrict__ lo = SAGEN_GRP(lo_);
    const float* __restrict__ part = SAGEN_GRP(part_);
    float* __restrict__ a_inv = SAGEN_GRP(a_inv_);
    float m = 0.f;
    for (int i = threadIdx.x & 63; i < S16_PARTS; i += 64) m = fmaxf(m, part[i]);
    m = wave_max_f(m);
    const float sa = h2_scale_of_bound(m);
    if (blockIdx.x == 0 && threadIdx.x == 0) a_inv[0] = 1.f / sa;
    const long total = (long)B * S8_UH * S8_UW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i;
        const int w = (int)(p % S8_UW) - 2; p /= S8_UW;
        const int h = (int)(p % S8_UH) - 2;
        const int b = (int)(p / S8_UH);
        unsigned short hh[3] = {0, 0, 0}, ll[3] = {0, 0, 0};
        if ((unsigned)h < 224u && (unsigned)w < 448u) {
            const float* src = x + (((long)b * 224 + h) * 448 + w) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = src[c] * sa;
                const _Float16 a = __builtin_isfinite(v) ? (_Float16)v : __builtin_bit_cast(_Float16, (unsigned short)0x7e00);    // (a NaN stays one)
                const _Float16 r = __builtin_isfinite(v) ? (_Float16)(v - (float)a) : (_Float16)0.f;
                hh[c] = __builtin_bit_cast(unsigned short, a); ll[c] = __builtin_bit_cast(unsigned short, r);
            }
        }
        hi[i] = u32x2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2]};
        lo[i] = u32x2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2]};
    }
}

// x: float frames [B,224,448,3]; planes: 2 * stem8_plane_bytes(B) bytes; part: S16_PARTS floats; a_inv: where 2^-ka goes;
// zero_ptr / zero_n: a buffer the first launch clears on the way (the batch-norm accumulators)
int stem16_prep_launch(const float* x, void* planes, float* part, float* a_inv, int B, hipStream_t s, float* zero_ptr, long zero_n) {
    if (!x || !planes || !part || !a_inv) return fail(SAGEN_ERR_NULL, "stem16_prep: null argument");
    const long n = (long)B * 224 * 448 * 3;
    if (n % 4 || ((uintptr_t)x % 16)) return fail(SAGEN_ERR_UNSUPPORTED, "stem16_prep: the frames must be 16-byte aligned");
    const GroupInfo gi = cur_group();
    hipLaunchKernelGGL(stem16_amax_kernel, dim3(S16_PARTS, 1, gi.G), dim3(256), 0, s, x, n / 4, part, zero_ptr, zero_ptr ? zero_n : 0L, gi);
    const long total = (long)B * S8_UH * S8_UW;
    char* p = reinterpret_cast<char*>(planes);
    hipLaunchKernelGGL(stem16_prep_kernel, dim3((int)std::min<long>(cdiv(total, 256), 256L * 16), 1, gi.G), dim3(256), 0, s, x, reinterpret_cast<u32x2*>(p),
                       reinterpret_cast<u32x2*>(p + stem8_plane_bytes(B)), B, part, a_inv, gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// planes: stem16_prep's output; wh2: the stem's filter as two fp16 planes [224/16][2][64][16] of w * 2^kw; a_inv / w_inv: the scales
int stem16pool_launch(const void* planes, const void* wh2, const float* gamma, float* pooled, double* stats, const float* a_inv, const float* w_inv,
                      int B, hipStream_t s) {
    if (!planes || !wh2 || !gamma || !pooled || !stats || !a_inv || !w_inv) return fail(SAGEN_ERR_NULL, "stem16pool: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL((stem8pool_kernel<0, 1>), dim3(2 * std::min(npatch, 256), 1, cur_group().G), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(planes),
                       (const float*)nullptr, reinterpret_cast<const char*>(wh2), gamma, pooled, stats, B, (long)stem8_plane_bytes(B), a_inv, w_inv, (float*)nullptr, cur_group());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
