// HBM-bound kernels of the network's backward pass (what tf.gradients / opt.minimize build for train.py:147-149,
// myutils.py:220-221): ReLU / bias gradients, training-mode batch-norm backward (core.py:6,209-210), max-pool backward
// (resnet.py:135), fan-in sums of tf.tile / tf.concat (model.py:230-236, 291-297), the BN moving averages (UPDATE_OPS,
// train.py:147-148) and the filter packs of the data-gradient contractions.  All are 16-byte-per-lane coalesced NHWC streams
// (channels innermost, C % 4 == 0); per-channel sums are accumulated in registers, reduced through LDS into one row of partials
// per workgroup, and the rows are added in a fixed order in fp64 by partials_finish_kernel (deterministic; no atomics).
#include "kernels.h"
#include "h2_planes.h"
#include <cstdlib>
#include <algorithm>

namespace sagen {

// grid whose stride (grid*256 threads) is a multiple of C4, so a thread always sees the same 4 channels
static int aligned_grid(long n4, int C4) {
    // SAGEN_BWD_AMORT items per thread at least: fewer, longer workgroups leave CUs to the weight gradients on the second stream
    // (training step on one box, tools/ab_bwd_amort.sh: 1 / 2 / 4 / 8 / 16 / 32 items = 6.09-6.11 / 6.06-6.08 / 6.01 / 5.98-6.01 / 6.07 / 6.28 ms)
    static const int amort = getenv("SAGEN_BWD_AMORT") ? std::max(1, atoi(getenv("SAGEN_BWD_AMORT"))) : 4;
    long g = std::min<long>(cdiv(n4, 256L * amort), 256L * 16);
    if ((g * 256) % C4) {
        long a = 256, b = C4;
        while (b) { const long t = a % b; a = b; b = t; }
        const long unit = C4 / a;                    // smallest g with (g*256) % C4 == 0
        g = std::max<long>(unit, g / unit * unit);
    }
    return (int)g;
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Per-channel sums without atomics: every workgroup reduces its threads' float4 partials through LDS and writes ONE row of
// per-channel partials, part[block][NV][C]; partials_finish_kernel then adds the rows in a fixed order in fp64 (deterministic,
// and no same-address atomic chains: 4096 workgroups x 128 fp64 atomics on 128 addresses took 300 us per batch-norm layer).
// Requires 256 % C4 == 0 (thread t owns channels 4*(t % C4)).
constexpr int RED_MAX_BLOCKS = 1024;              // capacity of the partial rows (scratch, t:mxpart); the launches use red_blocks() of them
// (SAGEN_RED_BLOCKS: A/B of the reductions' parallelism - 512 workgroups = two per CU)
static int red_blocks() {
    static const int n = getenv("SAGEN_RED_BLOCKS") ? std::min(RED_MAX_BLOCKS, std::max(64, atoi(getenv("SAGEN_RED_BLOCKS")))) : 512;
    return n;
}
template <int NV>
__device__ __forceinline__ void channel_partials(const float4 (&v)[NV], int C4, float* part) {
    __shared__ float4 red[NV][256];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k][tid] = v[k];
    __syncthreads();
    if (tid < C4) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float4 s = red[k][tid];
            for (int t = tid + C4; t < 256; t += C4) s = add4(s, red[k][t]);
            *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * NV + k) * (4 * C4) + 4 * tid) = s;
        }
    }
}

// acc[i] = sum_b part[b][i] (fp64), i < n: 8 columns x 32 row slices per workgroup, four rows in flight per thread (a first
// version with 8 slices and one load in flight walked 256 rows per thread: 60-100 us of pure latency per batch-norm layer)
__global__ __launch_bounds__(256) void partials_finish_kernel(const float* __restrict__ part, int nblocks, int n, double* __restrict__ acc,
                                                              float* __restrict__ out_f32) {
    __shared__ double red[32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + cl;
    double s = 0.0;
    if (i < n) {
        int b = sl;
        for (; b + 96 < nblocks; b += 128) {
            const float v0 = part[(size_t)b * n + i], v1 = part[(size_t)(b + 32) * n + i], v2 = part[(size_t)(b + 64) * n + i], v3 = part[(size_t)(b + 96) * n + i];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nblocks; b += 32) s += (double)part[(size_t)b * n + i];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][cl];
        acc[i] = t;
        if (out_f32) out_f32[i] = (float)t;
    }
}

static int reduce_grid(long n4, int C4) {
    long g = std::min<long>(cdiv(n4, 256 * 8), red_blocks());
    g = std::max<long>(g, 1);
    if ((g * 256) % C4) {
        long a = 256, b = C4;
        while (b) { const long t = a % b; a = b; b = t; }
        const long unit = C4 / a;
        g = std::max<long>(unit, g / unit * unit);
    }
    return (int)g;
}
size_t reduce_scratch_floats(int C) { return (size_t)RED_MAX_BLOCKS * 2 * ((C + 3) / 4 * 4); }

// -----------------------------------------------------------------------------------------
// dy = (ga + gb) * (act > 0)  [+ per-channel sum of dy = the bias gradient of tf.nn.bias_add, core.py:28]
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ ga, int lda, const float* __restrict__ gb, int ldb,
                                                       const float* __restrict__ act, int ldact, float* __restrict__ dy, int lddy,
                                                       long n4, int C4, double* __restrict__ colsum, int C, float* __restrict__ part,
                                                       long band_rows, long batch_rows, long row0) {
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);
    float4 sum[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        long row = i / C4;
        if (band_rows) row = row / band_rows * batch_rows + row0 + row % band_rows;      // a band of rows per batch element: the rest is never touched
        float4 v = *reinterpret_cast<const float4*>(ga + row * lda + 4 * c4);
        if (gb) v = add4(v, *reinterpret_cast<const float4*>(gb + row * ldb + 4 * c4));
        if (act) {
            const float4 a = *reinterpret_cast<const float4*>(act + row * ldact + 4 * c4);
            v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
        }
        if (dy) *reinterpret_cast<float4*>(dy + row * lddy + 4 * c4) = v;
        sum[0] = add4(sum[0], v);
    }
    if (part) channel_partials<1>(sum, C4, part);
    else if (colsum) {                       // odd widths (tiny tensors only): one fp64 atomic per thread and channel
        const float vv[4] = {sum[0].x, sum[0].y, sum[0].z, sum[0].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * c4 + k < C) atomicAdd(colsum + 4 * c4 + k, (double)vv[k]);
    }
}

// The same for SMALL tensors (the FC layers: 96 x 512) in one launch: 32 column groups x 8 row slices per workgroup, every column's
// sum finished inside the workgroup in a fixed order (the two-launch form above spends 12 + 5 us on a few KB).
__global__ __launch_bounds__(256) void relu_bwd_small_kernel(const float* __restrict__ ga, int lda, const float* __restrict__ gb, int ldb,
                                                             const float* __restrict__ act, int ldact, float* __restrict__ dy, int lddy,
                                                             int R, int C4, double* __restrict__ colsum, float* __restrict__ colsum_f32, int C,
                                                             long band_rows, long batch_rows, long row0) {
    __shared__ float4 red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c4 = blockIdx.x * 32 + cl;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        for (int r = sl; r < R; r += 8) {
            const long row = band_rows ? (long)r / band_rows * batch_rows + row0 + (long)r % band_rows : (long)r;
            float4 v = *reinterpret_cast<const float4*>(ga + (long)row * lda + 4 * c4);
            if (gb) v = add4(v, *reinterpret_cast<const float4*>(gb + (long)row * ldb + 4 * c4));
            if (act) {
                const float4 a = *reinterpret_cast<const float4*>(act + (long)row * ldact + 4 * c4);
                v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
            }
            if (dy) *reinterpret_cast<float4*>(dy + (long)row * lddy + 4 * c4) = v;
            sum = add4(sum, v);
        }
    }
    red[sl][cl] = sum;
    __syncthreads();
    if (sl == 0 && c4 < C4 && colsum) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float4 r = red[k][cl]; t[0] += r.x; t[1] += r.y; t[2] += r.z; t[3] += r.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            colsum[4 * c4 + k] = t[k];
            if (colsum_f32 && 4 * c4 + k < C) colsum_f32[4 * c4 + k] = (float)t[k];
        }
    }
}

int relu_bwd_launch(const float* ga, int lda, const float* gb, int ldb, const float* act, int ldact, float* dy, int lddy, long R,
                    int C, double* colsum, float* scratch, hipStream_t s, float* colsum_f32, long band_rows, long batch_rows, long row0) {
    if (!ga || (!dy && !colsum)) return fail(SAGEN_ERR_NULL, "relu_bwd: null argument");
    const int Cp = (C + 3) / 4 * 4;
    if (lda % 4 || (gb && ldb % 4) || (act && ldact % 4) || (dy && lddy % 4) || lda < Cp)
        return fail(SAGEN_ERR_UNSUPPORTED, "relu_bwd: row strides must be multiples of 4 floats covering the padded row (C=%d)", C);
    if (((uintptr_t)ga | (uintptr_t)gb | (uintptr_t)act | (uintptr_t)dy) % 16) return fail(SAGEN_ERR_UNSUPPORTED, "relu_bwd: operands must be 16-byte aligned");
    const int C4 = Cp / 4;
    const long n4 = R * C4;
    static const bool no_small = getenv("SAGEN_RELU_BWD_TWO_LAUNCHES") != nullptr;
    if (colsum && !no_small && n4 <= 32768 && R <= (1 << 20)) {
        hipLaunchKernelGGL(relu_bwd_small_kernel, dim3(cdiv(C4, 32)), dim3(256), 0, s, ga, lda, gb, ldb, act, ldact, dy, lddy, (int)R, C4, colsum,
                           colsum_f32, C, band_rows, batch_rows, row0);
        SAGEN_LAUNCH_CHECK();
        return SAGEN_OK;
    }
    const bool fast = colsum && scratch && 256 % C4 == 0;
    if (colsum && !fast) SAGEN_HIP_CHECK(hipMemsetAsync(colsum, 0, (size_t)Cp * sizeof(double), s));
    const int grid = colsum ? reduce_grid(n4, C4) : aligned_grid(n4, C4);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid), dim3(256), 0, s, ga, lda, gb, ldb, act, ldact, dy, lddy, n4, C4, colsum, C, fast ? scratch : nullptr,
                       band_rows, batch_rows, row0);
    SAGEN_LAUNCH_CHECK();
    if (fast) {
        hipLaunchKernelGGL(partials_finish_kernel, dim3(cdiv(Cp, 8)), dim3(256), 0, s, scratch, grid, Cp, colsum, colsum_f32);
        SAGEN_LAUNCH_CHECK();
    } else if (colsum && colsum_f32) {
        return acc_to_f32_launch(colsum, colsum_f32, C, s);
    }
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// training-mode batch-norm backward.  Forward: xhat = (y - mean) * invstd, z = gamma * xhat + beta (contrib batch_norm,
// is_training=True, biased variance, eps 1e-3).  With dz the gradient at z (after the ReLU mask of the consumer):
//   dbeta = sum dz,  dgamma = sum dz * xhat,  dy = gamma * invstd * (dz - dbeta / N - xhat * dgamma / N)
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_moments4(const BnRef& bn, int C, int c4, float4& mean, float4& invstd) {
    float m[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * c4 + k;
        const double mu = bn.acc[c] * bn.inv_count;
        double var = bn.acc[C + c] * bn.inv_count - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        m[k] = (float)mu;
        r[k] = (float)(1.0 / sqrt(var + (double)bn.eps));
    }
    mean = make_float4(m[0], m[1], m[2], m[3]);
    invstd = make_float4(r[0], r[1], r[2], r[3]);
}

// the forward's scale / shift of this layer, bit for bit (bn_coeffs4 of elementwise.hip, p3.hip): with them relu(bn(y)) > 0 can be
// re-derived from the raw conv output y that the backward reads anyway, and the retained activation need not be read for its sign
__device__ __forceinline__ void bn_forward_coeffs4(const BnRef& bn, int C, int c4, float4& sc, float4& sh) {
    float s4[4], h4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * c4 + k;
        const double mean = bn.acc[c] * bn.inv_count;
        double var = bn.acc[C + c] * bn.inv_count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double a = (double)bn.gamma[c] / sqrt(var + (double)bn.eps);
        s4[k] = (float)a;
        h4[k] = (float)((double)bn.beta[c] - mean * a);
    }
    sc = make_float4(s4[0], s4[1], s4[2], s4[3]);
    sh = make_float4(h4[0], h4[1], h4[2], h4[3]);
}
__device__ __forceinline__ float4 self_masked(float4 g, const float4& v, const float4& sc, const float4& sh) {
    g.x = fmaf(v.x, sc.x, sh.x) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
    g.z = fmaf(v.z, sc.z, sh.z) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
    return g;
}

// the ReLU mask as one bit per element (P3hScale::relu_bits): float4 index i covers the low / high nibble of byte i / 2
__device__ __forceinline__ float4 bit_masked(float4 g, const unsigned char* bits, long i) {
    const unsigned b = (unsigned)bits[i >> 1] >> (4 * (int)(i & 1));
    g.x = (b & 1u) ? g.x : 0.f; g.y = (b & 2u) ? g.y : 0.f; g.z = (b & 4u) ? g.z : 0.f; g.w = (b & 8u) ? g.w : 0.f;
    return g;
}
__device__ __forceinline__ float4 masked_sum(const float4* ga, const float4* gb, const float4* act, long i) {
    float4 v = ga[i];
    if (gb) v = add4(v, gb[i]);
    if (act) {
        const float4 a = act[i];
        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
    }
    return v;
}

// MX: also the workgroup's max |dz| and max |xhat| -> mx_part[2 * block + {0, 1}] (the bound of dy, bn_bwd_apply_h2_kernel)
template <bool MX>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float4* __restrict__ ga, const float4* __restrict__ gb,
                                                            const float4* __restrict__ act, const float4* __restrict__ y, const BnRef bn,
                                                            long n4, int C4, float* __restrict__ part, int self_mask,
                                                            float* __restrict__ mx_part, const unsigned char* __restrict__ bits) {
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);
    float4 mean, invstd;
    bn_moments4(bn, 4 * C4, c4, mean, invstd);
    float4 fsc = make_float4(0.f, 0.f, 0.f, 0.f), fsh = fsc;
    if (self_mask) bn_forward_coeffs4(bn, 4 * C4, c4, fsc, fsh);
    float4 sum[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    // two elements per trip: eight 16-byte loads in flight per lane (one element per trip ran at 2.9 TB/s)
    const long stride = (long)gridDim.x * 256;
    float mx_dz = 0.f, mx_xh = 0.f;
    auto accumulate = [&](const float4& dz, const float4& v) {
        sum[0] = add4(sum[0], dz);
        const float4 xh = make_float4((v.x - mean.x) * invstd.x, (v.y - mean.y) * invstd.y, (v.z - mean.z) * invstd.z, (v.w - mean.w) * invstd.w);
        sum[1].x = fmaf(dz.x, xh.x, sum[1].x); sum[1].y = fmaf(dz.y, xh.y, sum[1].y);
        sum[1].z = fmaf(dz.z, xh.z, sum[1].z); sum[1].w = fmaf(dz.w, xh.w, sum[1].w);
        if (MX) {
            mx_dz = fmaxf(mx_dz, fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w))));
            mx_xh = fmaxf(mx_xh, fmaxf(fmaxf(fabsf(xh.x), fabsf(xh.y)), fmaxf(fabsf(xh.z), fabsf(xh.w))));
        }
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 ga0 = ga[i], ga1 = ga[i + stride];
        const float4 v0 = y[i], v1 = y[i + stride];
        float4 dz0 = ga0, dz1 = ga1;
        if (gb) { dz0 = add4(dz0, gb[i]); dz1 = add4(dz1, gb[i + stride]); }
        if (act) {
            const float4 a0 = act[i], a1 = act[i + stride];
            dz0.x = a0.x > 0.f ? dz0.x : 0.f; dz0.y = a0.y > 0.f ? dz0.y : 0.f; dz0.z = a0.z > 0.f ? dz0.z : 0.f; dz0.w = a0.w > 0.f ? dz0.w : 0.f;
            dz1.x = a1.x > 0.f ? dz1.x : 0.f; dz1.y = a1.y > 0.f ? dz1.y : 0.f; dz1.z = a1.z > 0.f ? dz1.z : 0.f; dz1.w = a1.w > 0.f ? dz1.w : 0.f;
        }
        if (bits) { dz0 = bit_masked(dz0, bits, i); dz1 = bit_masked(dz1, bits, i + stride); }
        if (self_mask) { dz0 = self_masked(dz0, v0, fsc, fsh); dz1 = self_masked(dz1, v1, fsc, fsh); }
        accumulate(dz0, v0);
        accumulate(dz1, v1);
    }
    if (i < n4) {
        const float4 v = y[i];
        float4 dz = masked_sum(ga, gb, act, i);
        if (bits) dz = bit_masked(dz, bits, i);
        if (self_mask) dz = self_masked(dz, v, fsc, fsh);
        accumulate(dz, v);
    }
    channel_partials<2>(sum, C4, part);
    if (MX) {
        __shared__ float s_mx[2][4];
        mx_dz = wave_max_f(mx_dz); mx_xh = wave_max_f(mx_xh);
        if ((threadIdx.x & 63) == 0) { s_mx[0][threadIdx.x >> 6] = mx_dz; s_mx[1][threadIdx.x >> 6] = mx_xh; }
        __syncthreads();
        if (threadIdx.x < 2)
            mx_part[2 * blockIdx.x + threadIdx.x] = fmaxf(fmaxf(s_mx[threadIdx.x][0], s_mx[threadIdx.x][1]), fmaxf(s_mx[threadIdx.x][2], s_mx[threadIdx.x][3]));
    }
}

int bn_bwd_reduce_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, long n_pixels, int C,
                         double* acc, float* scratch, hipStream_t s, int self_mask, float* mx_part, int* mx_blocks, const unsigned char* relu_bits) {
    if ((self_mask || relu_bits) && act) return fail(SAGEN_ERR_SHAPE, "bn_bwd_reduce: self_mask / relu_bits replace the activation operand");
    if (relu_bits && C % 8) return fail(SAGEN_ERR_UNSUPPORTED, "bn_bwd_reduce: relu_bits need C %% 8 == 0");
    if (!ga || !y || !bn.acc || !acc || !scratch) return fail(SAGEN_ERR_NULL, "bn_bwd_reduce: null argument");
    if (C % 4 || 256 % (C / 4)) return fail(SAGEN_ERR_UNSUPPORTED, "bn_bwd_reduce: C=%d must be 4 * a divisor of 256", C);
    const long n4 = n_pixels * (C / 4);
    const int grid = reduce_grid(n4, C / 4);
    if (mx_part) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(grid), dim3(256), 0, s, (const float4*)ga, (const float4*)gb,
                           (const float4*)act, (const float4*)y, bn, n4, C / 4, scratch, self_mask, mx_part, relu_bits);
        if (mx_blocks) *mx_blocks = grid;
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(grid), dim3(256), 0, s, (const float4*)ga, (const float4*)gb,
                           (const float4*)act, (const float4*)y, bn, n4, C / 4, scratch, self_mask, (float*)nullptr, relu_bits);
    }
    SAGEN_LAUNCH_CHECK();
    hipLaunchKernelGGL(partials_finish_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, s, scratch, grid, 2 * C, acc, (float*)nullptr);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4* __restrict__ ga, const float4* __restrict__ gb,
                                                           const float4* __restrict__ act, const float4* __restrict__ y, const BnRef bn,
                                                           const double* __restrict__ acc, long n4, int C4, float4* __restrict__ dy,
                                                           float4* __restrict__ dz_out, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int self_mask) {
    const int C = 4 * C4;
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);
    float4 mean, invstd;
    bn_moments4(bn, C, c4, mean, invstd);
    float4 fsc = make_float4(0.f, 0.f, 0.f, 0.f), fsh = fsc;
    if (self_mask) bn_forward_coeffs4(bn, C, c4, fsc, fsh);
    float k0[4], k1[4], gs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * c4 + k;
        k0[k] = (float)(acc[c] * bn.inv_count);
        k1[k] = (float)(acc[C + c] * bn.inv_count);
        gs[k] = bn.gamma[c] * (&invstd.x)[k];
    }
    if (blockIdx.x == 0 && threadIdx.x < C4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * c4 + k;
            if (dbeta) dbeta[c] = (float)acc[c];
            if (dgamma) dgamma[c] = (float)acc[C + c];
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = y[i];
        float4 dz = masked_sum(ga, gb, act, i);
        if (self_mask) dz = self_masked(dz, v, fsc, fsh);
        float4 o;
        o.x = gs[0] * (dz.x - k0[0] - (v.x - mean.x) * invstd.x * k1[0]);
        o.y = gs[1] * (dz.y - k0[1] - (v.y - mean.y) * invstd.y * k1[1]);
        o.z = gs[2] * (dz.z - k0[2] - (v.z - mean.z) * invstd.z * k1[2]);
        o.w = gs[3] * (dz.w - k0[3] - (v.w - mean.w) * invstd.w * k1[3]);
        dy[i] = o;
        if (dz_out) dz_out[i] = dz;
    }
}

int bn_bwd_apply_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, const double* acc,
                        long n_pixels, int C, float* dy, float* dz_out, float* dgamma, float* dbeta, hipStream_t s, int self_mask) {
    if (self_mask && act) return fail(SAGEN_ERR_SHAPE, "bn_bwd_apply: self_mask replaces the activation operand");
    if (!ga || !y || !bn.acc || !acc || !dy) return fail(SAGEN_ERR_NULL, "bn_bwd_apply: null argument");
    if (C % 4 || 256 % (C / 4)) return fail(SAGEN_ERR_UNSUPPORTED, "bn_bwd_apply: C=%d must be 4 * a divisor of 256", C);
    const long n4 = n_pixels * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(aligned_grid(n4, C / 4)), dim3(256), 0, s, (const float4*)ga, (const float4*)gb,
                       (const float4*)act, (const float4*)y, bn, acc, n4, C / 4, (float4*)dy, (float4*)dz_out, dgamma, dbeta, self_mask);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// The same pass for a layer whose dy feeds a stride-1 3x3 data gradient on conv3h_kernel: dy ALSO as two fp16 planes of dy * 2^kd
// over the padded pixel grid [B*H][W+1] (h2_planes.h; the pad column is written as zeros).  2^kd from a bound every workgroup
// derives identically from what the reduce pass left: |dy_c| <= |gamma_c invstd_c| (max|dz| + |k0_c| + max|xhat| |k1_c|), scaled
// into [512, 1024) - an exact bound, so nothing saturates.  Eight channels per thread (one 16-byte store per plane).
__global__ __launch_bounds__(256) void bn_bwd_apply_h2_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                              const float* __restrict__ act, const float* __restrict__ y, const BnRef bn,
                                                              const double* __restrict__ acc, long nrows, int W, int C,
                                                              float* __restrict__ dy, float* __restrict__ dz_out,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int self_mask,
                                                              char* __restrict__ planes, const float* __restrict__ mx_part, int mx_blocks,
                                                              float* __restrict__ a_inv, unsigned* __restrict__ sat_count,
                                                              const unsigned char* __restrict__ bits) {
    const int C8 = C >> 3;
    const long total = nrows * (W + 1) * C8;
    const long cstride = nrows * (W + 1) * 64;
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(t0 % C8);
    __shared__ unsigned s_bits[3];
    if (threadIdx.x < 3) s_bits[threadIdx.x] = 0u;
    __syncthreads();
    {
        float m0 = 0.f, m1 = 0.f;
        for (int b = threadIdx.x; b < mx_blocks; b += 256) { m0 = fmaxf(m0, mx_part[2 * b]); m1 = fmaxf(m1, mx_part[2 * b + 1]); }
        m0 = wave_max_f(m0); m1 = wave_max_f(m1);
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&s_bits[0], __builtin_bit_cast(unsigned, m0));           // non-negative floats order like their bit patterns
            atomicMax(&s_bits[1], __builtin_bit_cast(unsigned, m1));
        }
    }
    __syncthreads();
    const float mx_dz = __builtin_bit_cast(float, s_bits[0]), mx_xh = __builtin_bit_cast(float, s_bits[1]);
    {
        float m = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) {
            const double mu = bn.acc[c] * bn.inv_count;
            double var = bn.acc[C + c] * bn.inv_count - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            const float gsc = bn.gamma[c] * (float)(1.0 / sqrt(var + (double)bn.eps));
            m = fmaxf(m, fabsf(gsc) * (mx_dz + fabsf((float)(acc[c] * bn.inv_count)) + mx_xh * fabsf((float)(acc[C + c] * bn.inv_count))));
        }
        m = wave_max_f(m);
        if ((threadIdx.x & 63) == 0) atomicMax(&s_bits[2], __builtin_bit_cast(unsigned, m));
    }
    __syncthreads();
    const float sa = h2_scale_of_bound(__builtin_bit_cast(float, s_bits[2]));
    if (blockIdx.x == 0 && threadIdx.x == 0) a_inv[0] = 1.f / sa;

    float mean[8], invstd[8], k0[8], k1[8], gs[8], fsc[8], fsh[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 m4, r4, sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
        bn_moments4(bn, C, 2 * c8 + h, m4, r4);
        if (self_mask) bn_forward_coeffs4(bn, C, 2 * c8 + h, sc4, sh4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 8 * c8 + 4 * h + k;
            mean[4 * h + k] = (&m4.x)[k]; invstd[4 * h + k] = (&r4.x)[k];
            fsc[4 * h + k] = (&sc4.x)[k]; fsh[4 * h + k] = (&sh4.x)[k];
            k0[4 * h + k] = (float)(acc[c] * bn.inv_count);
            k1[4 * h + k] = (float)(acc[C + c] * bn.inv_count);
            gs[4 * h + k] = bn.gamma[c] * (&r4.x)[k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < C8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = 8 * c8 + k;
            if (dbeta) dbeta[c] = (float)acc[c];
            if (dgamma) dgamma[c] = (float)acc[C + c];
        }
    }
    for (long i = t0; i < total; i += (long)gridDim.x * 256) {
        const long pp = i / C8;
        const long row = pp / (W + 1);
        const int w = (int)(pp - row * (W + 1));
        float o[8];
        if (w < W) {
            const long e = (row * W + w) * C + 8 * c8;
            float v[8], dz[8];
            *reinterpret_cast<float4*>(&v[0]) = *reinterpret_cast<const float4*>(y + e);
            *reinterpret_cast<float4*>(&v[4]) = *reinterpret_cast<const float4*>(y + e + 4);
            *reinterpret_cast<float4*>(&dz[0]) = *reinterpret_cast<const float4*>(ga + e);
            *reinterpret_cast<float4*>(&dz[4]) = *reinterpret_cast<const float4*>(ga + e + 4);
            if (gb) {
                const float4 b0 = *reinterpret_cast<const float4*>(gb + e), b1 = *reinterpret_cast<const float4*>(gb + e + 4);
                dz[0] += b0.x; dz[1] += b0.y; dz[2] += b0.z; dz[3] += b0.w; dz[4] += b1.x; dz[5] += b1.y; dz[6] += b1.z; dz[7] += b1.w;
            }
            if (act) {
                float a[8];
                *reinterpret_cast<float4*>(&a[0]) = *reinterpret_cast<const float4*>(act + e);
                *reinterpret_cast<float4*>(&a[4]) = *reinterpret_cast<const float4*>(act + e + 4);
#pragma unroll
                for (int k = 0; k < 8; ++k) dz[k] = a[k] > 0.f ? dz[k] : 0.f;
            }
            if (bits) {
                const unsigned bm = bits[e >> 3];
#pragma unroll
                for (int k = 0; k < 8; ++k) dz[k] = (bm >> k) & 1u ? dz[k] : 0.f;
            }
            if (self_mask) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dz[k] = fmaf(v[k], fsc[k], fsh[k]) > 0.f ? dz[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = gs[k] * (dz[k] - k0[k] - (v[k] - mean[k]) * invstd[k] * k1[k]);
            if (dy) {                                          // (null: every consumer of this dy reads the planes)
                *reinterpret_cast<float4*>(dy + e) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(dy + e + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
            if (dz_out) {
                *reinterpret_cast<float4*>(dz_out + e) = make_float4(dz[0], dz[1], dz[2], dz[3]);
                *reinterpret_cast<float4*>(dz_out + e + 4) = make_float4(dz[4], dz[5], dz[6], dz[7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = 0.f;
        }
        p3h_store(planes, cstride, pp, c8, o, sa, sat_count);
    }
}

int bn_bwd_apply_h2_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, const double* acc,
                           int B, int H, int W, int C, float* dy, float* dz_out, float* dgamma, float* dbeta, hipStream_t s, int self_mask,
                           void* planes, const float* mx_part, int mx_blocks, float* a_inv, unsigned* sat_count, const unsigned char* relu_bits) {
    if ((self_mask || relu_bits) && act) return fail(SAGEN_ERR_SHAPE, "bn_bwd_apply_h2: self_mask / relu_bits replace the activation operand");
    if (!ga || !y || !bn.acc || !acc || !planes || !mx_part || !a_inv) return fail(SAGEN_ERR_NULL, "bn_bwd_apply_h2: null argument");
    if (C % 16 || 256 % (C / 8) || mx_blocks < 1) return fail(SAGEN_ERR_UNSUPPORTED, "bn_bwd_apply_h2: C=%d must be 16 * a divisor of 128", C);
    const long total = (long)B * H * (W + 1) * (C / 8);
    hipLaunchKernelGGL(bn_bwd_apply_h2_kernel, dim3(aligned_grid(total, C / 8)), dim3(256), 0, s, ga, gb, act, y, bn, acc, (long)B * H, W, C,
                       dy, dz_out, dgamma, dbeta, self_mask, (char*)planes, mx_part, mx_blocks, a_inv, sat_count, relu_bits);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// max-pool 3x3/2 SAME backward through relu(bn(y0)).  tf.nn.max_pool's gradient routes each window's gradient to its maximum;
// here as a gather: input pixel (i, j) collects from the (at most 2x2) windows that contain it and whose pooled value equals
// its own activation (recomputed with the forward's exact fmaf / fmaxf sequence, so the comparison is exact).  A window whose
// maximum is 0 (all inputs <= 0) routes nowhere that survives the ReLU mask, in TF as here.  Exact float ties between two
// positive activations of one window would be credited twice (TF: first in scan order); not observed, measure-zero.
// -----------------------------------------------------------------------------------------
// MODE 0 writes the gradient at z0 = bn(y0) (after the ReLU mask); MODES 1 and 2 are the two passes of the batch-norm backward of
// the stem with that gradient RECOMPUTED on the fly (1: per-channel sums of dz and dz*xhat; 2: dy0), so dz0 - 205 MB at batch 32 -
// is never written nor read back: 0.93 GB of traffic for pool + batch-norm backward instead of 1.59 GB.
template <int MODE>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float4* __restrict__ y0, const BnRef bn, const float4* __restrict__ pooled,
                                                          const float4* __restrict__ ga, const float4* __restrict__ gb,
                                                          float4* __restrict__ out, int B, int H, int W, int C4, int Ho, int Wo, int pt, int pl,
                                                          float* __restrict__ part, const double* __restrict__ acc, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
    const int C = 4 * C4;
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);
    float4 sc, sh;
    bn_forward_coeffs4(bn, C, c4, sc, sh);              // identical to bn_coeffs4 of the forward pool (elementwise.hip)
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), invstd = mean;
    float k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f}, gs[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 0) bn_moments4(bn, C, c4, mean, invstd);
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * c4 + k;
            k0[k] = (float)(acc[c] * bn.inv_count);
            k1[k] = (float)(acc[C + c] * bn.inv_count);
            gs[k] = bn.gamma[c] * (&invstd.x)[k];
        }
        if (blockIdx.x == 0 && threadIdx.x < C4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * c4 + k;
                if (dbeta) dbeta[c] = (float)acc[c];
                if (dgamma) dgamma[c] = (float)acc[C + c];
            }
        }
    }
    float4 sum[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const long total = (long)B * H * W * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long p = idx / C4;
        const int j = (int)(p % W); p /= W;
        const int i = (int)(p % H);
        const int b = (int)(p / H);
        const float4 v = y0[idx];
        float4 a;
        a.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); a.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        a.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); a.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        const int oi0 = max((i + pt - 1) >> 1, 0), oi1 = min((i + pt) >> 1, Ho - 1);      // windows rows 2*oi - pt .. 2*oi - pt + 2
        const int oj0 = max((j + pl - 1) >> 1, 0), oj1 = min((j + pl) >> 1, Wo - 1);
        for (int oi = oi0; oi <= oi1; ++oi)
            for (int oj = oj0; oj <= oj1; ++oj) {
                const long q = (((long)b * Ho + oi) * Wo + oj) * C4 + c4;
                const float4 pv = pooled[q];
                float4 gv = ga[q];
                if (gb) gv = add4(gv, gb[q]);
                g.x += (a.x == pv.x) ? gv.x : 0.f; g.y += (a.y == pv.y) ? gv.y : 0.f;
                g.z += (a.z == pv.z) ? gv.z : 0.f; g.w += (a.w == pv.w) ? gv.w : 0.f;
            }
        g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
        if (MODE == 0) out[idx] = g;
        else if (MODE == 1) {
            sum[0] = add4(sum[0], g);
            sum[1].x = fmaf(g.x, (v.x - mean.x) * invstd.x, sum[1].x); sum[1].y = fmaf(g.y, (v.y - mean.y) * invstd.y, sum[1].y);
            sum[1].z = fmaf(g.z, (v.z - mean.z) * invstd.z, sum[1].z); sum[1].w = fmaf(g.w, (v.w - mean.w) * invstd.w, sum[1].w);
        } else {
            float4 o;
            o.x = gs[0] * (g.x - k0[0] - (v.x - mean.x) * invstd.x * k1[0]);
            o.y = gs[1] * (g.y - k0[1] - (v.y - mean.y) * invstd.y * k1[1]);
            o.z = gs[2] * (g.z - k0[2] - (v.z - mean.z) * invstd.z * k1[2]);
            o.w = gs[3] * (g.w - k0[3] - (v.w - mean.w) * invstd.w * k1[3]);
            out[idx] = o;
        }
    }
    if (MODE == 1) channel_partials<2>(sum, C4, part);
}

// The first pass of maxpool_bn_bwd on the POOLED grid (round 5): every window routes its gradient to its extremum, whose raw value the
// training forward now keeps (stem8pool_kernel<raw+pool>: `praw` = max, or min where gamma < 0, of the raw stem output over the
// window; relu(bn(praw)) IS the pooled activation) - so sum dz and sum dz * xhat are sums over the windows: 4 x 51 MB read instead of the
// 205 MB raw tensor plus its 2x2 gathers.  (An exact float tie inside a window is credited once here and twice by the gather of the second
// pass - both measure-zero deviations from TF's first-in-scan-order rule, ~1e-6 of the sums.)
__global__ __launch_bounds__(256) void pool_bn_reduce_kernel(const float4* __restrict__ praw, const BnRef bn, const float4* __restrict__ ga,
                                                             const float4* __restrict__ gb, long total, int C4, float* __restrict__ part) {
    const int C = 4 * C4;
    const int c4 = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C4);
    float4 sc, sh, mean, invstd;
    bn_forward_coeffs4(bn, C, c4, sc, sh);
    bn_moments4(bn, C, c4, mean, invstd);
    float4 sum[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const float4 v = praw[idx];
        float4 g = ga[idx];
        if (gb) g = add4(g, gb[idx]);
        g.x = fmaf(v.x, sc.x, sh.x) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
        g.z = fmaf(v.z, sc.z, sh.z) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
        sum[0] = add4(sum[0], g);
        sum[1].x = fmaf(g.x, (v.x - mean.x) * invstd.x, sum[1].x); sum[1].y = fmaf(g.y, (v.y - mean.y) * invstd.y, sum[1].y);
        sum[1].z = fmaf(g.z, (v.z - mean.z) * invstd.z, sum[1].z); sum[1].w = fmaf(g.w, (v.w - mean.w) * invstd.w, sum[1].w);
    }
    channel_partials<2>(sum, C4, part);
}

static int maxpool_bwd_check(const float* y0, const BnRef& bn, const float* pooled, const float* ga, const float* out, int C) {
    if (!y0 || !bn.acc || !pooled || !ga || !out) return fail(SAGEN_ERR_NULL, "maxpool_bwd: null argument");
    if (C % 4) return fail(SAGEN_ERR_UNSUPPORTED, "maxpool_bwd: C=%d must be a multiple of 4", C);
    return SAGEN_OK;
}

int maxpool_bwd_launch(const float* y0, const BnRef& bn, const float* pooled, const float* ga, const float* gb, float* dz, int B,
                       int H, int W, int C, hipStream_t s) {
    if (int rc = maxpool_bwd_check(y0, bn, pooled, ga, dz, C)) return rc;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pth = std::max((Ho - 1) * 2 + 3 - H, 0), ptw = std::max((Wo - 1) * 2 + 3 - W, 0);
    const long total = (long)B * H * W * (C / 4);
    // (i + pt - 1) >> 1 must be ceil((i + pt - 2) / 2): true for i + pt >= 1; i + pt == 0 gives -1 >> 1 = -1 -> clamped to 0
    hipLaunchKernelGGL(maxpool_bwd_kernel<0>, dim3(aligned_grid(total, C / 4)), dim3(256), 0, s, (const float4*)y0, bn, (const float4*)pooled,
                       (const float4*)ga, (const float4*)gb, (float4*)dz, B, H, W, C / 4, Ho, Wo, pth / 2, ptw / 2, (float*)nullptr,
                       (const double*)nullptr, (float*)nullptr, (float*)nullptr);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// x0 = maxpool(relu(bn0(y0))): gradient at y0 and at gamma / beta from the gradient at x0 (ga + gb), dz0 never materialised
int maxpool_bn_bwd_launch(const float* y0, const BnRef& bn, const float* pooled, const float* ga, const float* gb, float* dy0, int B,
                          int H, int W, int C, double* acc, float* scratch, float* dgamma, float* dbeta, hipStream_t s, const float* pooled_raw) {
    if (int rc = maxpool_bwd_check(y0, bn, pooled, ga, dy0, C)) return rc;
    if (!acc || !scratch) return fail(SAGEN_ERR_NULL, "maxpool_bn_bwd: null accumulator / scratch");
    if (256 % (C / 4)) return fail(SAGEN_ERR_UNSUPPORTED, "maxpool_bn_bwd: C=%d must be 4 * a divisor of 256", C);
    if (C > 128) return fail(SAGEN_ERR_UNSUPPORTED, "maxpool_bn_bwd: C=%d > 128 (scratch rows)", C);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pth = std::max((Ho - 1) * 2 + 3 - H, 0), ptw = std::max((Wo - 1) * 2 + 3 - W, 0);
    const long total = (long)B * H * W * (C / 4);
    // (gather-heavy: more, shorter workgroups than the streaming reductions; rows of 2C floats)
    static const int gmax = getenv("SAGEN_POOLBWD_GRID") ? atoi(getenv("SAGEN_POOLBWD_GRID")) : 2048;
    int grid = (int)std::max<long>(1, std::min<long>(cdiv(total, 256 * 4), gmax));
    if (pooled_raw) {                                  // the sums over the pooled grid (the forward kept each window's raw extremum)
        const long ptotal = (long)B * Ho * Wo * (C / 4);
        grid = reduce_grid(ptotal, C / 4);
        hipLaunchKernelGGL(pool_bn_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float4*)pooled_raw, bn, (const float4*)ga, (const float4*)gb, ptotal, C / 4,
                           scratch);
    } else
    hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, (const float4*)y0, bn, (const float4*)pooled, (const float4*)ga,
                       (const float4*)gb, (float4*)nullptr, B, H, W, C / 4, Ho, Wo, pth / 2, ptw / 2, scratch, (const double*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    SAGEN_LAUNCH_CHECK();
    hipLaunchKernelGGL(partials_finish_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, s, scratch, grid, 2 * C, acc, (float*)nullptr);
    SAGEN_LAUNCH_CHECK();
    hipLaunchKernelGGL(maxpool_bwd_kernel<2>, dim3(aligned_grid(total, C / 4)), dim3(256), 0, s, (const float4*)y0, bn, (const float4*)pooled,
                       (const float4*)ga, (const float4*)gb, (float4*)dy0, B, H, W, C / 4, Ho, Wo, pth / 2, ptw / 2, (float*)nullptr,
                       (const double*)acc, dgamma, dbeta);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// out[m][c] = sum_{r<rep} (ina[(m*rep + r)][c] + inb[...][c])
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ ina, int lda, const float* __restrict__ inb, int ldb,
                                                       int rep, long n4, int C4, float* __restrict__ out, int ldo) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long m = i / C4;
        const int c4 = (int)(i - m * C4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < rep; ++r) {
            v = add4(v, *reinterpret_cast<const float4*>(ina + (m * rep + r) * lda + 4 * c4));
            if (inb) v = add4(v, *reinterpret_cast<const float4*>(inb + (m * rep + r) * ldb + 4 * c4));
        }
        *reinterpret_cast<float4*>(out + m * ldo + 4 * c4) = v;
    }
}

int sum_rows_launch(const float* ina, int lda, const float* inb, int ldb, int rep, long M, int C, float* out, int ldo, hipStream_t s) {
    if (!ina || !out) return fail(SAGEN_ERR_NULL, "sum_rows: null argument");
    if (C % 4 || lda % 4 || (inb && ldb % 4) || ldo % 4 || rep < 1) return fail(SAGEN_ERR_UNSUPPORTED, "sum_rows: C / strides must be multiples of 4");
    if (((uintptr_t)ina | (uintptr_t)inb | (uintptr_t)out) % 16) return fail(SAGEN_ERR_UNSUPPORTED, "sum_rows: operands must be 16-byte aligned");
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((int)std::min<long>(cdiv(n4, 256), 4096L)), dim3(256), 0, s, ina, lda, inb, ldb, rep, n4, C / 4, out, ldo);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

__global__ __launch_bounds__(256) void acc_to_f32_kernel(const double* __restrict__ acc, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)acc[i];
}

int acc_to_f32_launch(const double* acc, float* dst, int n, hipStream_t s) {
    if (!acc || !dst) return fail(SAGEN_ERR_NULL, "acc_to_f32: null argument");
    hipLaunchKernelGGL(acc_to_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, acc, dst, n);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// tf.contrib.layers.batch_norm(decay=0.99, is_training=True) update ops (core.py:210; run through UPDATE_OPS, train.py:147-148):
// moving <- decay * moving + (1 - decay) * batch statistic.  TF 1.4's contrib layer takes the fused path for rank-4 inputs
// (fused=None), whose running variance is the UNBIASED batch variance (nn.fused_batch_norm); the normalisation itself uses the
// biased one.  The moving averages are never read by the path (is_training is always True, model.py:197): bookkeeping only.
__global__ __launch_bounds__(256) void bn_moving_update_kernel(const BnRef bn, float* __restrict__ mm, float* __restrict__ mv, int C, float decay) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mean = bn.acc[c] * bn.inv_count;
    double var = bn.acc[C + c] * bn.inv_count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double n = 1.0 / bn.inv_count;
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    mm[c] = decay * mm[c] + (1.f - decay) * (float)mean;
    mv[c] = decay * mv[c] + (1.f - decay) * (float)unbiased;
}

// every batch-norm layer of a step in ONE launch (34 launches of a few hundred threads were ~90 us of launch latency at the end of
// the step): the job table travels as a kernel argument
struct BnMovingJobs { BnMovingJob j[BN_MOVING_MAX_JOBS]; };
__global__ __launch_bounds__(256) void bn_moving_update_multi_kernel(const BnMovingJobs jobs, float decay) {
    const BnMovingJob& j = jobs.j[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= j.C) return;
    const double mean = j.acc[c] * j.inv_count;
    double var = j.acc[j.C + c] * j.inv_count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double n = 1.0 / j.inv_count;
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    j.mm[c] = decay * j.mm[c] + (1.f - decay) * (float)mean;
    j.mv[c] = decay * j.mv[c] + (1.f - decay) * (float)unbiased;
}

int bn_moving_update_multi_launch(const BnMovingJob* jobs, int n, float decay, hipStream_t s) {
    if (n <= 0) return SAGEN_OK;
    if (!jobs || n > BN_MOVING_MAX_JOBS) return fail(SAGEN_ERR_SHAPE, "bn_moving_update_multi: %d jobs (max %d)", n, BN_MOVING_MAX_JOBS);
    BnMovingJobs t;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!jobs[i].acc || !jobs[i].mm || !jobs[i].mv) return fail(SAGEN_ERR_NULL, "bn_moving_update_multi: null argument");
        t.j[i] = jobs[i];
        cmax = std::max(cmax, jobs[i].C);
    }
    hipLaunchKernelGGL(bn_moving_update_multi_kernel, dim3(cdiv(cmax, 256), n), dim3(256), 0, s, t, decay);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int bn_moving_update_launch(const BnRef& bn, float* moving_mean, float* moving_var, int C, float decay, hipStream_t s) {
    if (!bn.acc || !moving_mean || !moving_var) return fail(SAGEN_ERR_NULL, "bn_moving_update: null argument");
    hipLaunchKernelGGL(bn_moving_update_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, bn, moving_mean, moving_var, C, decay);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// filter packs of the data-gradient contractions
// -----------------------------------------------------------------------------------------
// dx of a stride-1 conv = conv of dy with the tap-reversed, channel-transposed filter
__global__ __launch_bounds__(256) void pack_conv_flipT_kernel(const float* __restrict__ w, int ntaps, int cin, int cout,
                                                              float* __restrict__ wp, int Kpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)cin * Kpad) return;
    const int n = (int)(idx / Kpad), k = (int)(idx - (long)n * Kpad);
    float v = 0.f;
    if (k < ntaps * cout) {
        const int tap = k / cout, co = k - tap * cout;
        v = w[((long)(ntaps - 1 - tap) * cin + n) * cout + co];
    }
    wp[idx] = v;
}

int pack_conv_flipT_launch(const float* w_hwio, int ntaps, int cin, int cout, float* wp, int Kpad, hipStream_t s) {
    const long total = (long)cin * Kpad;
    hipLaunchKernelGGL(pack_conv_flipT_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w_hwio, ntaps, cin, cout, wp, Kpad);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wp, int Kpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * Kpad) return;
    const int n = (int)(idx / Kpad), k = (int)(idx - (long)n * Kpad);
    wp[idx] = k < cols ? w[(long)n * cols + k] : 0.f;
}

int pack_rows_launch(const float* w, int rows, int cols, float* wp, int Kpad, hipStream_t s) {
    const long total = (long)rows * Kpad;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, rows, cols, wp, Kpad);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

__global__ __launch_bounds__(256) void stem_wgrad_unpack_kernel(const float* __restrict__ tmp, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // over [7][7][3][64]
    if (idx >= 7 * 7 * 3 * 64) return;
    const int o = idx & 63;
    int r = idx >> 6;
    const int c = r % 3; r /= 3;
    const int tw = r % 7, th = r / 7;
    dw[idx] = tmp[((th * 32) + tw * 4 + c) * 64 + o];
}

int stem_wgrad_unpack_launch(const float* tmp, float* dw, hipStream_t s) {
    hipLaunchKernelGGL(stem_wgrad_unpack_kernel, dim3(cdiv(7 * 7 * 3 * 64, 256)), dim3(256), 0, s, tmp, dw);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
