// Producers of "P3" activation tensors: the operand format of conv3p_kernel (conv3p.hip).
//
// A P3 tensor holds an NHWC fp32 activation [B,H,W,C] (C % 16 == 0) as its three bf16 planes hi / mid / lo
// (hi = rne(v), mid = rne(v - hi), lo = rne(v - hi - mid); both residuals exact in fp32; hi + mid + lo carries
// 24 mantissa bits), laid out [C/16][NP][3][16] with NP = B*H*(W+1): every image row is followed by one ZERO pixel,
// which serves as the right-hand SAME padding of its row and the left-hand padding of the next.  The split is done
// here, once per element, by the HBM-bound pass that has to touch the tensor anyway:
//   * p3_pack_kernel     y = [relu]( x*scale + shift [+ residual] )  ->  fp32 NHWC (optional) + P3 (optional)
//                        = tf batch_norm(is_training) + ReLU of conv_1 and the residual merge (resnet.py:215-221, 230-235)
//   * p3_maxpool_kernel  3x3/2 SAME max-pool of relu(bn(x)) (resnet.py:134-135) -> fp32 NHWC + P3
// Each thread owns 8 consecutive channels of one pixel: two 16-byte loads, three 16-byte plane stores.
#include "igemm3_common.h"
#include "h2_planes.h"

namespace sagen {

size_t p3_bytes(int B, int H, int W, int C) { return (size_t)(C / 16) * B * H * (W + 1) * 96; }

// Per-block table of the batch-norm coefficients: scale = gamma / sqrt(var + eps), shift = beta - mean * scale from the
// producer's fp64 (sum, sumsq) accumulators (contrib batch_norm, training mode, core.py:6,209-210; biased variance).
// One fp64 sqrt + divide per CHANNEL and block instead of eight per thread (which made the small late-stage tensors
// latency-bound: 16 MB in 11-14 us).
constexpr int P3_MAX_C = 512;
__device__ __forceinline__ void p3_bn_table(const float* scale, const float* shift, const BnRef& bn, int C, float (*tab)[P3_MAX_C]) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sc = 1.f, sh = 0.f;
        if (bn.acc != nullptr) {
            const double mean = bn.acc[c] * bn.inv_count;
            double var = bn.acc[C + c] * bn.inv_count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double a = (double)bn.gamma[c] / sqrt(var + (double)bn.eps);
            sc = (float)a;
            sh = (float)((double)bn.beta[c] - mean * a);
        } else if (scale != nullptr) {
            sc = scale[c];
            sh = shift[c];
        }
        tab[0][c] = sc;
        tab[1][c] = sh;
    }
    __syncthreads();
}

// 8 fp32 values -> the three bf16 planes (8 bf16 = 16 bytes each)
__device__ __forceinline__ void p3_split8(float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) hi[k] = split_pair(v[2 * k], v[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) mid[k] = split_pair(v[2 * k], v[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) lo[k] = split_pair(v[2 * k], v[2 * k + 1]);
}

__device__ __forceinline__ void p3_store(char* p3, long cstride, long pp, int c8, const u32x4& hi, const u32x4& mid, const u32x4& lo) {
    char* dst = p3 + (long)(c8 >> 1) * cstride + pp * 96 + (c8 & 1) * 16;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 32) = mid;
    *reinterpret_cast<u32x4*>(dst + 64) = lo;
}

// ---- format 1 ("H2", conv3h.hip): two fp16 planes (hi, lo) of v * 2^ka, [C/16][NP][2][16], 64 B per pixel and chunk ----
size_t p3h_bytes(int B, int H, int W, int C) { return (size_t)(C / 16) * B * H * (W + 1) * 64; }

// 2^ka for the tensor this pass writes, from STATISTICS the forward already holds (no extra pass over the data):
//   * the batch-norm branch: channel c of relu(bn(y)) has mean beta_c and standard deviation |gamma_c| sqrt(var_c / (var_c + eps))
//     (NOT |gamma_c|: a batch-norm whose input variance is below eps does not normalise to unit variance) -> bound_c = |beta_c| + 8 std_c;
//   * the residual branch of a block merge: either the bound of the block input, tracked from pass to pass in device memory
//     (`h.res_bound`), or - first block of a stage - the (sum, sumsq) of the 1x1 shortcut conv's output: |mean_c| + 8 std_c.
// bound = max_c(bn) + max_c(residual) is scaled into [512, 1024): fp16 overflows at 65504, i.e. 64 x headroom for what lies beyond
// eight standard deviations.  Every block derives the same value; block 0 publishes 2^-ka (read by conv3h_kernel's
// epilogue) and the bound itself (the residual bound of the next merge).  Without a BnRef the scale is 1.
//
// NOTHING CAN SATURATE (round 5): next to the statistical bound the pass carries a RIGOROUS one.  The batch statistics are those of
// the very tensor being normalised, and no sample of N values lies further than sqrt(N - 1) standard deviations from their mean
// (Samuelson's inequality), so |bn(y)_c| <= |beta_c| + sqrt(N) |gamma_c| sqrt(var_c / (var_c + eps)) for EVERY element, whatever the
// distribution; a 1x1 projection's output obeys |mean_c| + sqrt(N) std_c the same way, and an identity residual inherits the rigorous
// bound of the block input (tracked like the statistical one, H2_RIG_OFF floats behind it).  ka = min(ka of the statistical bound,
// the largest k with rigorous_bound * 2^k <= 64000): wherever sqrt(N) / 8 < 64 - every pass of the network at batch 32 except the
// stem's pool (N = B * 112 * 224) - that is the statistical ka unchanged; the stem's planes give up one bit (of 34) at batch 32.
// The saturation counter is therefore an assertion (0 unless an input was not finite - which p3h_store turns into NaN, not a clamp).

// Both tables in ONE pass over the channels (one round trip to the statistics, two barriers): the batch-norm coefficients and, for
// the fp16x2 format, the activation scale.
template <bool H2>
__device__ __forceinline__ float p3_tables(const float* scale, const float* shift, const BnRef& bn, const P3hScale& h, bool has_res, int C,
                                           float (*tab)[P3_MAX_C], unsigned* s_bits) {
    if (H2 && threadIdx.x < 4) s_bits[threadIdx.x] = 0u;
    if (H2) __syncthreads();
    float m = 0.f, mr = 0.f, mg = 0.f, mrg = 0.f;             // statistical bounds (bn branch, residual branch) and the rigorous ones
    const float sqn = H2 && bn.acc != nullptr ? (float)sqrt(1.0 / bn.inv_count) : 0.f;
    const float sqn_r = H2 && has_res && h.res_acc != nullptr ? (float)sqrt(1.0 / h.res_inv_count) : 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sc = 1.f, sh = 0.f;
        if (bn.acc != nullptr) {
            const double mean = bn.acc[c] * bn.inv_count;
            double var = bn.acc[C + c] * bn.inv_count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double inv = 1.0 / sqrt(var + (double)bn.eps);
            const double a = (double)bn.gamma[c] * inv;
            sc = (float)a;
            sh = (float)((double)bn.beta[c] - mean * a);
            if (H2) {
                const float gs = fabsf(bn.gamma[c]) * (float)(sqrt(var) * inv);
                m = fmaxf(m, fabsf(bn.beta[c]) + 8.f * gs);
                mg = fmaxf(mg, fabsf(bn.beta[c]) + sqn * gs);
                if (has_res && h.res_acc != nullptr) {
                    const double rm = h.res_acc[c] * h.res_inv_count;
                    double rv = h.res_acc[C + c] * h.res_inv_count - rm * rm;
                    rv = rv < 0.0 ? 0.0 : rv;
                    mr = fmaxf(mr, (float)(fabs(rm) + 8.0 * sqrt(rv)));
                    mrg = fmaxf(mrg, (float)(fabs(rm) + (double)sqn_r * sqrt(rv)));
                }
            }
        } else if (scale != nullptr) {
            sc = scale[c];
            sh = shift[c];
        }
        tab[0][c] = sc;
        tab[1][c] = sh;
    }
    if (H2 && bn.acc != nullptr) {
        m = wave_max_f(m); mr = wave_max_f(mr); mg = wave_max_f(mg); mrg = wave_max_f(mrg);
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&s_bits[0], __builtin_bit_cast(unsigned, m));        // non-negative floats order like their bit patterns
            atomicMax(&s_bits[1], __builtin_bit_cast(unsigned, mr));
            atomicMax(&s_bits[2], __builtin_bit_cast(unsigned, mg));
            atomicMax(&s_bits[3], __builtin_bit_cast(unsigned, mrg));
        }
    }
    __syncthreads();
    if (!H2) return 1.f;
    float bound = __builtin_bit_cast(float, s_bits[0]) + __builtin_bit_cast(float, s_bits[1]);
    float rig = __builtin_bit_cast(float, s_bits[2]) + __builtin_bit_cast(float, s_bits[3]);
    if (has_res && h.res_acc == nullptr && h.res_bound != nullptr) { bound += h.res_bound[0]; rig += h.res_bound[H2_RIG_OFF]; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && h.bound_out != nullptr) { h.bound_out[0] = bound; h.bound_out[H2_RIG_OFF] = rig; }
    const unsigned b = __builtin_bit_cast(unsigned, bound);
    const int e = (int)((b >> 23) & 0xff);
    float sa = 1.f;
    if (bn.acc != nullptr && e != 0 && e != 255) {
        int k = 127 + 9 - e;                                                // statistical bound -> [512, 1024)
        const float q = 64000.f / rig;                                      // rigorous bound * 2^k <= 64000 < 65504
        const int eq = (int)((__builtin_bit_cast(unsigned, q) >> 23) & 0xff);
        if (eq != 0 && eq != 255) k = min(k, eq - 127);                     // (floor(log2 q); rig = 0 or not finite: no constraint)
        sa = __builtin_bit_cast(float, (unsigned)(127 + max(-60, min(60, k))) << 23);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && h.a_inv) h.a_inv[0] = 1.f / sa;
    return sa;
}

template <bool H2>
__global__ __launch_bounds__(256) void p3_pack_kernel(const float* __restrict__ x_, const float* __restrict__ scale_,
                                                      const float* __restrict__ shift_, const BnRef bn_,
                                                      const float* __restrict__ res_, int relu, float* __restrict__ y_,
                                                      char* p3_, long nrows, int W, int C, const P3hScale h2_, const GroupInfo gi) {       // (p3 may alias h2.res_planes)
    // grouped launch (common.h): this workgroup's group = blockIdx.z, its tensors / accumulators / scales in that group's region
    const float* __restrict__ x = SAGEN_GRP(x_);
    const float* __restrict__ scale = SAGEN_GRP(scale_);
    const float* __restrict__ shift = SAGEN_GRP(shift_);
    const float* __restrict__ res = SAGEN_GRP(res_);
    float* __restrict__ y = SAGEN_GRP(y_);
    char* p3 = SAGEN_GRP(p3_);
    BnRef bn = bn_;
    bn.acc = SAGEN_GRP(bn.acc);
    P3hScale h2 = h2_;
    h2.a_inv = SAGEN_GRP(h2.a_inv); h2.res_bound = SAGEN_GRP(h2.res_bound); h2.res_acc = SAGEN_GRP(h2.res_acc);
    h2.bound_out = SAGEN_GRP(h2.bound_out); h2.sat_count = SAGEN_GRP(h2.sat_count); h2.relu_bits = SAGEN_GRP(h2.relu_bits);
    h2.res_planes = SAGEN_GRP(h2.res_planes); h2.res_a_inv = SAGEN_GRP(h2.res_a_inv);
    const int C8 = C >> 3;
    const long total = nrows * (W + 1) * C8;             // padded pixels x channel octets (the grid stride is a multiple of C8)
    const long cstride = nrows * (W + 1) * (H2 ? 64 : 96);
    __shared__ unsigned s_bits[4];
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(t0 % C8);
    __shared__ __attribute__((aligned(16))) float tab[2][P3_MAX_C];
    const char* const resp = H2 ? reinterpret_cast<const char*>(h2.res_planes) : nullptr;
    const float sa = p3_tables<H2>(scale, shift, bn, h2, res != nullptr || resp != nullptr, C, tab, s_bits);
    const float r_inv = resp ? h2.res_a_inv[0] : 0.f;
    const float4 sc0 = *reinterpret_cast<const float4*>(&tab[0][8 * c8]), sc1 = *reinterpret_cast<const float4*>(&tab[0][8 * c8 + 4]);
    const float4 sh0 = *reinterpret_cast<const float4*>(&tab[1][8 * c8]), sh1 = *reinterpret_cast<const float4*>(&tab[1][8 * c8 + 4]);
    for (long i = t0; i < total; i += (long)gridDim.x * 256) {
        const long pp = i / C8;
        const long row = pp / (W + 1);
        const int w = (int)(pp - row * (W + 1));
        float v[8];
        if (w < W) {
            const long e = (row * W + w) * C + 8 * c8;
            const float4 a = *reinterpret_cast<const float4*>(x + e), b = *reinterpret_cast<const float4*>(x + e + 4);
            v[0] = fmaf(a.x, sc0.x, sh0.x); v[1] = fmaf(a.y, sc0.y, sh0.y); v[2] = fmaf(a.z, sc0.z, sh0.z); v[3] = fmaf(a.w, sc0.w, sh0.w);
            v[4] = fmaf(b.x, sc1.x, sh1.x); v[5] = fmaf(b.y, sc1.y, sh1.y); v[6] = fmaf(b.z, sc1.z, sh1.z); v[7] = fmaf(b.w, sc1.w, sh1.w);
            if (res) {
                const float4 ra = *reinterpret_cast<const float4*>(res + e), rb = *reinterpret_cast<const float4*>(res + e + 4);
                v[0] += ra.x; v[1] += ra.y; v[2] += ra.z; v[3] += ra.w; v[4] += rb.x; v[5] += rb.y; v[6] += rb.z; v[7] += rb.w;
            }
            if (H2 && resp) {                // the residual from its planes: (hi + lo) * 2^-ka, exact in fp32 (both are multiples of the lo ulp)
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                const char* src = resp + (long)(c8 >> 1) * cstride + pp * 64 + (c8 & 1) * 16;
                const h8 rh = *reinterpret_cast<const h8*>(src), rl = *reinterpret_cast<const h8*>(src + 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += ((float)rh[k] + (float)rl[k]) * r_inv;
            }
            if (relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (y) {
                *reinterpret_cast<float4*>(y + e) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(y + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (h2.relu_bits) {
                unsigned bits = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) bits |= (v[k] > 0.f ? 1u : 0u) << k;
                h2.relu_bits[e >> 3] = (unsigned char)bits;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        if (p3) {
            if (H2) {
                p3h_store(p3, cstride, pp, c8, v, sa, h2.sat_count);
            } else {
                u32x4 hi, mid, lo;
                p3_split8(v, hi, mid, lo);
                p3_store(p3, cstride, pp, c8, hi, mid, lo);
            }
        }
    }
}

// grid whose stride (grid*256 threads) is a multiple of `unit_threads`, so per-thread channel groups are loop-invariant
static int aligned_grid(long total, int unit_threads, int amort = 1) {
    // <= 8 blocks per CU: the coefficient table is amortised over several elements per thread; `amort` elements per thread at least
    // where the table is the larger part of a workgroup's work (deep stages: 256 / 512 channels of fp64 finalisation for 256 items)
    long g = std::min<long>(cdiv(total, 256L * amort), 256L * 8);
    long a = 256, b = unit_threads;
    while (b) { const long t = a % b; a = b; b = t; }
    const long unit = unit_threads / a;                  // smallest g with (g*256) % unit_threads == 0
    g = std::max<long>(unit, g / unit * unit);
    return (int)g;
}

int p3_pack_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, const float* residual, int relu,
                   float* y, void* p3, int B, int H, int W, int C, hipStream_t s, int fmt, const P3hScale* h2) {
    if (C % 16 || C > P3_MAX_C) return fail(SAGEN_ERR_UNSUPPORTED, "p3_pack: C=%d must be a multiple of 16, at most %d", C, P3_MAX_C);
    if (!y && !p3) return fail(SAGEN_ERR_NULL, "p3_pack: no output");
    const long nrows = (long)B * H;
    if (p3_bytes(B, H, W, C) >= (1UL << 31)) return fail(SAGEN_ERR_UNSUPPORTED, "p3_pack: tensor exceeds 2 GiB buffer addressing");
    const long total = nrows * (W + 1) * (C / 8);
    if (fmt == 1 && p3 && !(h2 && h2->a_inv)) return fail(SAGEN_ERR_NULL, "p3_pack: the fp16x2 format needs a slot for its scale");
    static const int amort_env = getenv("SAGEN_P3_AMORT") ? atoi(getenv("SAGEN_P3_AMORT")) : 0;
    // (measured with three batches in flight, same box, tools/ab_p3_amort.sh: 1 / 2 / 4 items per thread = 2 508-2 524 / 2 519-2 529 / 2 536-2 544
    //  ambisonic-s/s - the pass alone is no faster with fewer workgroups, but it leaves the CUs to the other batches' matrix kernels sooner)
    const int amort = amort_env > 0 ? amort_env : 4;
    const GroupInfo gi = cur_group();
    if (fmt == 1)
        hipLaunchKernelGGL(p3_pack_kernel<true>, dim3(aligned_grid(total, C / 8, amort), 1, gi.G), dim3(256), 0, s, x, scale, shift, bn, residual, relu, y,
                           (char*)p3, nrows, W, C, *h2, gi);
    else
        hipLaunchKernelGGL(p3_pack_kernel<false>, dim3(aligned_grid(total, C / 8, amort), 1, gi.G), dim3(256), 0, s, x, scale, shift, bn, residual, relu, y,
                           (char*)p3, nrows, W, C, P3hScale(), gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// ---- a band of rows of a plain fp32 tensor as fp16x2 planes, scaled by its producers' EXACT maximum (the decoder's concat buffers:
//      no batch-norm bounds them, so the contractions that wrote the two halves publish max |y| from their epilogues) ----
__global__ __launch_bounds__(256) void h2_pack_rows_kernel(const float* __restrict__ x_, long x_bstride, long x_rstride, int ldx, int row0, int B, int R,
                                                           int W, int C, const float* __restrict__ amax0_, const float* __restrict__ amax1_,
                                                           char* __restrict__ planes_, float* __restrict__ a_inv_, unsigned* __restrict__ sat_count_, const GroupInfo gi) {
    const float* __restrict__ x = SAGEN_GRP(x_);
    const float* __restrict__ amax0 = SAGEN_GRP(amax0_);
    const float* __restrict__ amax1 = SAGEN_GRP(amax1_);
    char* __restrict__ planes = SAGEN_GRP(planes_);
    float* __restrict__ a_inv = SAGEN_GRP(a_inv_);
    unsigned* __restrict__ sat_count = SAGEN_GRP(sat_count_);
    const int C8 = C >> 3;
    const long NP = (long)B * R * (W + 1);
    const long total = NP * C8;
    // the producers' maxima: H2_AMAX_SLOTS words each (one lane per slot; every wave folds them itself)
    float bound = amax0[H2_AMAX_STRIDE * (threadIdx.x & (H2_AMAX_SLOTS - 1))];
    if (amax1) bound = fmaxf(bound, amax1[H2_AMAX_STRIDE * (threadIdx.x & (H2_AMAX_SLOTS - 1))]);
    bound = wave_max_f(bound);
    const float sa = h2_scale_of_bound(bound);               // bound * sa in [512, 1024): an exact bound, nothing saturates
    if (blockIdx.x == 0 && threadIdx.x == 0) a_inv[0] = 1.f / sa;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pp = i / C8;
        const int c8 = (int)(i - pp * C8);
        const long row = pp / (W + 1);
        const int w = (int)(pp - row * (W + 1));
        float v[8];
        if (w < W) {
            const long b = row / R;
            const int r = (int)(row - b * R);
            const float* src = x + b * x_bstride + (long)(row0 + r) * x_rstride + (long)w * ldx + 8 * c8;
            const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        p3h_store(planes, NP * 64, pp, c8, v, sa, sat_count);
    }
}

int h2_pack_rows_launch(const float* x, long x_bstride, long x_rstride, int ldx, int row0, int B, int R, int W, int C, const float* amax0,
                        const float* amax1, void* planes, float* a_inv, unsigned* sat_count, hipStream_t s) {
    if (!x || !amax0 || !planes || !a_inv) return fail(SAGEN_ERR_NULL, "h2_pack_rows: null argument");
    if (C % 16 || ldx % 4 || ((uintptr_t)x % 16)) return fail(SAGEN_ERR_UNSUPPORTED, "h2_pack_rows: C %% 16, ldx %% 4 and a 16-byte aligned tensor are required");
    const long total = (long)B * R * (W + 1) * (C / 8);
    const GroupInfo gi = cur_group();
    hipLaunchKernelGGL(h2_pack_rows_kernel, dim3((int)std::min<long>(cdiv(total, 256), 2048), 1, gi.G), dim3(256), 0, s, x, x_bstride, x_rstride, ldx, row0,
                       B, R, W, C, amax0, amax1, (char*)planes, a_inv, sat_count, gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// tf.nn.max_pool(x,[1,3,3,1],[1,2,2,1],'SAME') (resnet.py:135) of relu(bn(x)), -inf padding; see elementwise.hip
template <bool H2>
__global__ __launch_bounds__(256) void p3_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const BnRef bn, float* __restrict__ y,
                                                         char* __restrict__ p3, int B, int H, int W, int C, int Ho, int Wo,
                                                         int pt, int pl, const P3hScale h2) {
    const int C8 = C >> 3;
    const long nrows = (long)B * Ho;
    const long total = nrows * (Wo + 1) * C8;
    const long cstride = nrows * (Wo + 1) * (H2 ? 64 : 96);
    __shared__ unsigned s_bits[4];
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(t0 % C8);
    __shared__ __attribute__((aligned(16))) float tab[2][P3_MAX_C];
    const float sa = p3_tables<H2>(scale, shift, bn, h2, false, C, tab, s_bits);
    const float4 sc0 = *reinterpret_cast<const float4*>(&tab[0][8 * c8]), sc1 = *reinterpret_cast<const float4*>(&tab[0][8 * c8 + 4]);
    const float4 sh0 = *reinterpret_cast<const float4*>(&tab[1][8 * c8]), sh1 = *reinterpret_cast<const float4*>(&tab[1][8 * c8 + 4]);
    const bool has_bn = scale != nullptr || bn.acc != nullptr;
    for (long i = t0; i < total; i += (long)gridDim.x * 256) {
        const long pp = i / C8;
        const long row = pp / (Wo + 1);
        const int wo = (int)(pp - row * (Wo + 1));
        const int ho = (int)(row % Ho);
        const long b = row / Ho;
        float v[8];
        if (wo < Wo) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int h = ho * 2 + dy - pt;
                if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int w = wo * 2 + dx - pl;
                    if ((unsigned)w >= (unsigned)W) continue;
                    const long e = ((b * H + h) * W + w) * C + 8 * c8;
                    const float4 a = *reinterpret_cast<const float4*>(x + e), bq = *reinterpret_cast<const float4*>(x + e + 4);
                    v[0] = fmaxf(v[0], fmaf(a.x, sc0.x, sh0.x)); v[1] = fmaxf(v[1], fmaf(a.y, sc0.y, sh0.y));
                    v[2] = fmaxf(v[2], fmaf(a.z, sc0.z, sh0.z)); v[3] = fmaxf(v[3], fmaf(a.w, sc0.w, sh0.w));
                    v[4] = fmaxf(v[4], fmaf(bq.x, sc1.x, sh1.x)); v[5] = fmaxf(v[5], fmaf(bq.y, sc1.y, sh1.y));
                    v[6] = fmaxf(v[6], fmaf(bq.z, sc1.z, sh1.z)); v[7] = fmaxf(v[7], fmaf(bq.w, sc1.w, sh1.w));
                }
            }
            if (has_bn) {           // relu commutes with max
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (y) {
                const long e = (row * Wo + wo) * C + 8 * c8;
                *reinterpret_cast<float4*>(y + e) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(y + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        if (p3) {
            if (H2) {
                p3h_store(p3, cstride, pp, c8, v, sa, h2.sat_count);
            } else {
                u32x4 hi, mid, lo;
                p3_split8(v, hi, mid, lo);
                p3_store(p3, cstride, pp, c8, hi, mid, lo);
            }
        }
    }
}

int p3_maxpool_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, float* y, void* p3, int B, int H,
                      int W, int C, hipStream_t s, int fmt, const P3hScale* h2) {
    if (C % 16 || C > P3_MAX_C) return fail(SAGEN_ERR_UNSUPPORTED, "p3_maxpool: C=%d must be a multiple of 16, at most %d", C, P3_MAX_C);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pth = std::max((Ho - 1) * 2 + 3 - H, 0), ptw = std::max((Wo - 1) * 2 + 3 - W, 0);
    const long total = (long)B * Ho * (Wo + 1) * (C / 8);
    if (fmt == 1 && p3 && !(h2 && h2->a_inv)) return fail(SAGEN_ERR_NULL, "p3_maxpool: the fp16x2 format needs a slot for its scale");
    if (cur_group().G > 1) return fail(SAGEN_ERR_UNSUPPORTED, "p3_maxpool: no grouped launch (the grouped forward runs the fused stem kernels)");
    if (fmt == 1)
        hipLaunchKernelGGL(p3_maxpool_kernel<true>, dim3(aligned_grid(total, C / 8)), dim3(256), 0, s, x, scale, shift, bn, y, (char*)p3, B, H,
                           W, C, Ho, Wo, pth / 2, ptw / 2, *h2);
    else
        hipLaunchKernelGGL(p3_maxpool_kernel<false>, dim3(aligned_grid(total, C / 8)), dim3(256), 0, s, x, scale, shift, bn, y, (char*)p3, B, H,
                           W, C, Ho, Wo, pth / 2, ptw / 2, P3hScale());
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
