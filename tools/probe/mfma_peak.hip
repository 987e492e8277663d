// What does the matrix pipe sustain when EVERY SIMD issues nothing but v_mfma_f32_32x32x16_bf16?  (tools/probe/mfma_peak.py)
// grid = blocks of 256 threads (4 waves); each wave: iters x NACC independent accumulators, operands from registers.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* ticks) {
    bf16x8 a[NACC], b[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int e = 0; e < 8; ++e) {
            a[k][e] = (__bf16)src[(threadIdx.x * 8 + e + 17 * k) & 4095];
            b[k][e] = (__bf16)src[(threadIdx.x * 8 + e + 31 * k + 5) & 4095];
        }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

extern "C" int mfma_run(const float* src, float* out, int blocks, int iters, int nacc, unsigned long long* ticks, void* stream) {
    if (nacc == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, iters, ticks);
    else if (nacc == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, iters, ticks);
    else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, iters, ticks);
    return (int)hipGetLastError();
}
