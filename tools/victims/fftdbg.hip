// Dev helper: the library's 1024-point LDS FFT with a dump of the LDS state after every stage.
#include <hip/hip_runtime.h>
__device__ float2 g_tw[1024];
__global__ void tables() { int n = blockIdx.x * 256 + threadIdx.x; if (n >= 1024) return; double s, c; sincospi(2.0 * n / 1024.0, &s, &c); g_tw[n] = make_float2((float)c, (float)(-s)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// dump: [blocks][6][1024] float2  (slot 0 = input, 1..5 = after stage 0..4)
__global__ __launch_bounds__(256) void fft_dump(const float2* __restrict__ in, float2* __restrict__ dump) {
    __shared__ float2 bufA[1024], bufB[1024];
    const int tid = threadIdx.x;
    float2* d = dump + (size_t)blockIdx.x * 6 * 1024;
    for (int t = 0; t < 4; ++t) { const float2 v = in[(size_t)(blockIdx.x & 63) * 1024 + tid + 256 * t]; bufA[tid + 256 * t] = v; d[tid + 256 * t] = v; }
    __syncthreads();
    float2* src = bufA; float2* dst = bufB;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int Ns = 1 << (2 * s);
        const int k = tid & (Ns - 1);
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (s > 0) {
            const int step = k * (256 / Ns);
            float2 w1 = g_tw[step], w2 = g_tw[2 * step], w3 = g_tw[3 * step];
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
        }
        const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
        const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 dd = make_float2(u1.x - u3.x, u1.y - u3.y);
        const float2 v3 = make_float2(dd.y, -dd.x);
        const int j0 = ((tid - k) << 2) + k;
        dst[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
        dst[j0 + Ns] = make_float2(v1.x + v3.x, v1.y + v3.y);
        dst[j0 + 2 * Ns] = make_float2(v0.x - v2.x, v0.y - v2.y);
        dst[j0 + 3 * Ns] = make_float2(v1.x - v3.x, v1.y - v3.y);
        __syncthreads();
        for (int t = 0; t < 4; ++t) d[(s + 1) * 1024 + tid + 256 * t] = dst[tid + 256 * t];
        __syncthreads();
        float2* tt = src; src = dst; dst = tt;
    }
}
extern "C" {
void dbg_init(void* s) { hipLaunchKernelGGL(tables, dim3(4), dim3(256), 0, (hipStream_t)s); }
void dbg_run(const void* in, void* dump, int blocks, void* s) { hipLaunchKernelGGL(fft_dump, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float2*)in, (float2*)dump); }
}
