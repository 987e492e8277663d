// First pieces of the training step (reference train.py:137-236): everything on either side of the network's
// backward pass that is not a contraction -
//   * the loss: losses['stft/mse'] = metrics['stft/avg'] (model.py:156-159), the masked mean of the per-sample stft distance
//     (model.py:62-76, 122-127), and its gradient with respect to the prediction;
//   * the optimiser: tf.train.AdamOptimizer(lr_t) (myutils.py:214-222) as ONE fused launch over a flat parameter bucket.
// The backward of the contraction layers (conv / conv-transpose / FC data and weight gradients, training-mode BN) is not built
// yet; DESIGN.md 7 says what is missing.
//
// Loss algebra.  stft_for_loss (myutils.py:151-178) cuts each 4800-sample channel into three Hann-windowed frames of 2048
// ([0,2048), [2048,4096), [1024,3072)); the per-sample distance is mean_frames mean_bins |FFT(h * (gt - pred))|^2, which by
// Parseval is (1/3) sum_n W2[n] (gt - pred)[n]^2 with W2[n] = sum of h^2 over the frames covering n: no transform, forward or
// backward.  With  L = mean_c( 100 / max(sum_b mask[b,c], 1) * sum_b mask[b,c] * dist[b,c] ):
//     dL/dpred[b,n,c] = (100 / 3) / max(sum_b mask[b,c], 1) * mask[b,c] * (2/3) * W2[n] * (pred - gt)[b,n,c].
#include "kernels.h"

namespace sagen {

constexpr int TR_N = 4800, TR_C = 3, TR_W = 2048;

__device__ __forceinline__ float stft_loss_w2(int n) {
    float w = 0.f;
    if (n < 2 * TR_W) {
        const float h = (float)(0.5 - 0.5 * cospi(2.0 * (double)(n % TR_W) / (double)TR_W));
        w += h * h;
    }
    if (n >= TR_W / 2 && n < TR_W / 2 + TR_W) {
        const float h = (float)(0.5 - 0.5 * cospi(2.0 * (double)(n - TR_W / 2) / (double)TR_W));
        w += h * h;
    }
    return w;
}

// one workgroup per (b, c): partial loss sum_n W2 d^2 -> loss accumulators (fp64 atomics); gradient written in place
__global__ __launch_bounds__(256) void stft_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             const float* __restrict__ mask, int B, float* __restrict__ grad,
                                                             double* __restrict__ loss) {
    const int b = blockIdx.x / TR_C, c = blockIdx.x % TR_C;
    float nm = 0.f;
    for (int i = 0; i < B; ++i) nm += mask ? mask[i * TR_C + c] : 1.f;
    nm = fmaxf(nm, 1.f);
    const float mk = mask ? mask[b * TR_C + c] : 1.f;
    const float gscale = (100.f / 3.f) / nm * mk * (2.f / 3.f);
    const long base = (long)b * TR_N * TR_C + c;
    float acc = 0.f;
    for (int n = threadIdx.x; n < TR_N; n += 256) {
        const float d = pred[base + (long)n * TR_C] - gt[base + (long)n * TR_C];
        const float w2 = stft_loss_w2(n);
        acc += w2 * d * d;
        if (grad) grad[base + (long)n * TR_C] = gscale * w2 * d;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss)
        atomicAdd(loss, (double)(red[0] + red[1] + red[2] + red[3]) / 3.0 * (double)mk * (100.0 / 3.0) / (double)nm);
}

int stft_loss_grad_launch(const float* pred, const float* gt, const float* mask, int B, float* grad, double* loss, hipStream_t s) {
    if (loss) SAGEN_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(double), s));
    hipLaunchKernelGGL(stft_loss_grad_kernel, dim3(B * TR_C), dim3(256), 0, s, pred, gt, mask, B, grad, loss);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// tf.train.AdamOptimizer.apply_dense on a flat bucket (TF 1.4 formulation: epsilon is added to sqrt(v), the bias corrections
// live in lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), computed on the host):
//   m <- beta1 m + (1 - beta1) g ;  v <- beta2 v + (1 - beta2) g^2 ;  p <- p - lr_t * m / (sqrt(v) + eps)
// `gscale` multiplies the raw bucket first (1 / world_size after the sum all-reduce).
__global__ __launch_bounds__(256) void adam_update_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                          float4* __restrict__ v, long n4, float lr_t, float beta1, float beta2,
                                                          float eps, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * gscale;
            M[k] = beta1 * M[k] + (1.f - beta1) * gk;
            V[k] = beta2 * V[k] + (1.f - beta2) * gk * gk;
            P[k] -= lr_t * M[k] / (sqrtf(V[k]) + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

int adam_update_launch(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2, float eps,
                       float gscale, hipStream_t s) {
    if (n % 4) return fail(SAGEN_ERR_SHAPE, "adam_update: bucket length %ld must be a multiple of 4 floats", n);
    const long n4 = n / 4;
    const int grid = (int)std::min<long>(cdiv(n4, 256), 256L * 16);
    hipLaunchKernelGGL(adam_update_kernel, dim3(grid), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n4, lr_t,
                       beta1, beta2, eps, gscale);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
