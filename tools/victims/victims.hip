// Dev helper: tiny "victim" kernels, each exercising one instruction class, to find what a co-resident bf16-MFMA kernel
// disturbs.  Built on the GPU box:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC victims.hip -o /tmp/libvictims.so
#include <hip/hip_runtime.h>
__device__ float2 g_table[1024];

__global__ void fill_table() { int n = blockIdx.x * 256 + threadIdx.x; if (n < 1024) g_table[n] = make_float2(n * 0.5f, -n * 0.25f); }

// V0: read a __device__ table (global loads of 8-B elements), write it out
__global__ void v_table(float2* out, int reps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float2 a = make_float2(0.f, 0.f);
    for (int r = 0; r < reps; ++r) { const float2 w = g_table[(t * 7 + r * 13) & 1023]; a.x += w.x; a.y += w.y; }
    out[t] = a;
}
// V1: pure VALU fma chain
__global__ void v_valu(float* out, int reps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a = t * 1e-3f, b = 1.0001f;
    for (int r = 0; r < reps; ++r) { a = fmaf(a, b, 0.5f); b = fmaf(b, 0.99999f, 1e-6f); }
    out[t] = a + b;
}
// V2: LDS ping-pong with barriers (like an FFT stage, no global reads inside)
__global__ void v_lds(float* out, int reps) {
    __shared__ float2 A[1024], Bf[1024];
    const int tid = threadIdx.x;
    for (int i = 0; i < 4; ++i) A[tid + 256 * i] = make_float2((tid + 256 * i) * 1e-3f + blockIdx.x, 1.f);
    __syncthreads();
    float2* src = A; float2* dst = Bf;
    for (int r = 0; r < reps; ++r) {
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        const int j = (tid * 4 + r) & 1023;
        dst[j] = make_float2(u0.x + u2.x, u0.y - u2.y); dst[(j + 1) & 1023] = make_float2(u1.x + u3.x, u1.y - u3.y);
        dst[(j + 2) & 1023] = make_float2(u0.x - u2.x, u0.y + u2.y); dst[(j + 3) & 1023] = make_float2(u1.x - u3.x, u1.y + u3.y);
        __syncthreads();
        float2* tt = src; src = dst; dst = tt;
    }
    for (int i = 0; i < 4; ++i) { const float2 v = src[tid + 256 * i]; out[(blockIdx.x * 1024 + tid + 256 * i)] = v.x * 1e-3f + v.y * 1e-3f; }
}
// V3: sqrt / transcendental
__global__ void v_sqrt(float* out, int reps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a = t + 1.5f;
    for (int r = 0; r < reps; ++r) a = sqrtf(a * a + 1.0f) + 0.25f;
    out[t] = a;
}
namespace copy {
__device__ float2 g_tw[1024];     // exp(-2 pi i n / 1024)
__device__ float g_hann[1024];    // float32(0.5 - 0.5 cos(2 pi n / 1024))   (myutils.py:134)

__global__ void fft_tables_kernel() {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 1024) return;
    double s, c;
    sincospi(2.0 * (double)n / 1024.0, &s, &c);
    g_tw[n] = make_float2((float)c, (float)(-s));
    g_hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI / 1024.0 * (double)n));
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In: a[1024]; out: returned pointer (a or b). 256 threads. Unnormalised in both directions.
template <bool INV>
__device__ __forceinline__ float2* fft1024(float2* a, float2* b, int tid) {
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int Ns = 1 << (2 * s);
        const int k = tid & (Ns - 1);
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (s > 0) {
            const int step = k * (256 / Ns);
            float2 w1 = g_tw[step], w2 = g_tw[2 * step], w3 = g_tw[3 * step];
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
        }
        const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
        const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
        const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        const int j0 = ((tid - k) << 2) + k;
        dst[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
        dst[j0 + Ns] = make_float2(v1.x + v3.x, v1.y + v3.y);
        dst[j0 + 2 * Ns] = make_float2(v0.x - v2.x, v0.y - v2.y);
        dst[j0 + 3 * Ns] = make_float2(v1.x - v3.x, v1.y - v3.y);
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;
}

// -----------------------------------------------------------------------------------------
// STFT: grid (ceil(nframes/2), B). hop 256, window 1024, periodic Hann.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ audio, int n_samples, int f0, int f1,
                                                   float* __restrict__ mag, int c0, int c1, float2* __restrict__ spec) {
    __shared__ float2 bufA[1024], bufB[1024];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int fa = f0 + 2 * blockIdx.x, fb = fa + 1;
    const bool has_b = fb < f1;
    const float* xa = audio + (long)b * n_samples + 256L * fa;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = tid + 256 * t;
        const float h = g_hann[n];
        const float va = xa[n] * h;
        const float vb = has_b ? xa[256 + n] * h : 0.f;
        bufA[n] = make_float2(va, vb);
    }
    __syncthreads();
    const float2* Z = fft1024<false>(bufA, bufB, tid);
    const int nf = f1 - f0, nc = c1 - c0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = tid + 256 * t;
        const float2 zk = Z[k];
        float2 zn = Z[(1024 - k) & 1023];
        zn.y = -zn.y;
        const float2 xA = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
        const float dx = zk.x - zn.x, dy = zk.y - zn.y;
        const float2 xB = make_float2(0.5f * dy, -0.5f * dx);
        if (mag) {
            mag[((long)b * nf + (fa - f0)) * 1024 + k] = sqrtf(xA.x * xA.x + xA.y * xA.y);
            if (has_b) mag[((long)b * nf + (fb - f0)) * 1024 + k] = sqrtf(xB.x * xB.x + xB.y * xB.y);
        }
        if (spec && k <= 512) {   // bins > 512 are the Hermitian mirror (mag still needs them)
            if (fa >= c0 && fa < c1) spec[((long)b * nc + (fa - c0)) * 513 + k] = xA;
            if (has_b && fb >= c0 && fb < c1) spec[((long)b * nc + (fb - c0)) * 513 + k] = xB;
        }
    }
}

}
namespace copy_lds {
__device__ float2 g_tw[1024];     // exp(-2 pi i n / 1024)
__device__ float g_hann[1024];    // float32(0.5 - 0.5 cos(2 pi n / 1024))   (myutils.py:134)

__global__ void fft_tables_kernel() {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 1024) return;
    double s, c;
    sincospi(2.0 * (double)n / 1024.0, &s, &c);
    g_tw[n] = make_float2((float)c, (float)(-s));
    g_hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI / 1024.0 * (double)n));
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In: a[1024]; out: returned pointer (a or b). 256 threads. Unnormalised in both directions.
template <bool INV>
__device__ __forceinline__ float2* fft1024(float2* a, float2* b, int tid, const float2* twl) {
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int Ns = 1 << (2 * s);
        const int k = tid & (Ns - 1);
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (s > 0) {
            const int step = k * (256 / Ns);
            float2 w1 = twl[step], w2 = twl[2 * step], w3 = twl[3 * step];
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
        }
        const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
        const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
        const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        const int j0 = ((tid - k) << 2) + k;
        dst[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
        dst[j0 + Ns] = make_float2(v1.x + v3.x, v1.y + v3.y);
        dst[j0 + 2 * Ns] = make_float2(v0.x - v2.x, v0.y - v2.y);
        dst[j0 + 3 * Ns] = make_float2(v1.x - v3.x, v1.y - v3.y);
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;
}

// -----------------------------------------------------------------------------------------
// STFT: grid (ceil(nframes/2), B). hop 256, window 1024, periodic Hann.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ audio, int n_samples, int f0, int f1,
                                                   float* __restrict__ mag, int c0, int c1, float2* __restrict__ spec) {
    __shared__ float2 bufA[1024], bufB[1024], twl[1024];
    for (int t = 0; t < 4; ++t) twl[threadIdx.x + 256 * t] = g_tw[threadIdx.x + 256 * t];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int fa = f0 + 2 * blockIdx.x, fb = fa + 1;
    const bool has_b = fb < f1;
    const float* xa = audio + (long)b * n_samples + 256L * fa;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = tid + 256 * t;
        const float h = g_hann[n];
        const float va = xa[n] * h;
        const float vb = has_b ? xa[256 + n] * h : 0.f;
        bufA[n] = make_float2(va, vb);
    }
    __syncthreads();
    const float2* Z = fft1024<false>(bufA, bufB, tid, twl);
    const int nf = f1 - f0, nc = c1 - c0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = tid + 256 * t;
        const float2 zk = Z[k];
        float2 zn = Z[(1024 - k) & 1023];
        zn.y = -zn.y;
        const float2 xA = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
        const float dx = zk.x - zn.x, dy = zk.y - zn.y;
        const float2 xB = make_float2(0.5f * dy, -0.5f * dx);
        if (mag) {
            mag[((long)b * nf + (fa - f0)) * 1024 + k] = sqrtf(xA.x * xA.x + xA.y * xA.y);
            if (has_b) mag[((long)b * nf + (fb - f0)) * 1024 + k] = sqrtf(xB.x * xB.x + xB.y * xB.y);
        }
        if (spec && k <= 512) {   // bins > 512 are the Hermitian mirror (mag still needs them)
            if (fa >= c0 && fa < c1) spec[((long)b * nc + (fa - c0)) * 513 + k] = xA;
            if (has_b && fb >= c0 && fb < c1) spec[((long)b * nc + (fb - c0)) * 513 + k] = xB;
        }
    }
}

}
namespace copy_a16 {
__device__ __attribute__((aligned(16))) float2 g_tw[1024];     // exp(-2 pi i n / 1024)
__device__ float g_hann[1024];    // float32(0.5 - 0.5 cos(2 pi n / 1024))   (myutils.py:134)

__global__ void fft_tables_kernel() {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 1024) return;
    double s, c;
    sincospi(2.0 * (double)n / 1024.0, &s, &c);
    g_tw[n] = make_float2((float)c, (float)(-s));
    g_hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI / 1024.0 * (double)n));
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In: a[1024]; out: returned pointer (a or b). 256 threads. Unnormalised in both directions.
template <bool INV>
__device__ __forceinline__ float2* fft1024(float2* a, float2* b, int tid) {
    float2* src = a;
    float2* dst = b;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int Ns = 1 << (2 * s);
        const int k = tid & (Ns - 1);
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (s > 0) {
            const int step = k * (256 / Ns);
            auto ld = [](int i) { const float4 q = reinterpret_cast<const float4*>(g_tw)[i >> 1]; return (i & 1) ? make_float2(q.z, q.w) : make_float2(q.x, q.y); };
            float2 w1 = ld(step), w2 = ld(2 * step), w3 = ld(3 * step);
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
        }
        const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
        const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
        const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        const int j0 = ((tid - k) << 2) + k;
        dst[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
        dst[j0 + Ns] = make_float2(v1.x + v3.x, v1.y + v3.y);
        dst[j0 + 2 * Ns] = make_float2(v0.x - v2.x, v0.y - v2.y);
        dst[j0 + 3 * Ns] = make_float2(v1.x - v3.x, v1.y - v3.y);
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;
}

// -----------------------------------------------------------------------------------------
// STFT: grid (ceil(nframes/2), B). hop 256, window 1024, periodic Hann.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ audio, int n_samples, int f0, int f1,
                                                   float* __restrict__ mag, int c0, int c1, float2* __restrict__ spec) {
    __shared__ float2 bufA[1024], bufB[1024];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int fa = f0 + 2 * blockIdx.x, fb = fa + 1;
    const bool has_b = fb < f1;
    const float* xa = audio + (long)b * n_samples + 256L * fa;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = tid + 256 * t;
        const float h = g_hann[n];
        const float va = xa[n] * h;
        const float vb = has_b ? xa[256 + n] * h : 0.f;
        bufA[n] = make_float2(va, vb);
    }
    __syncthreads();
    const float2* Z = fft1024<false>(bufA, bufB, tid);
    const int nf = f1 - f0, nc = c1 - c0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = tid + 256 * t;
        const float2 zk = Z[k];
        float2 zn = Z[(1024 - k) & 1023];
        zn.y = -zn.y;
        const float2 xA = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
        const float dx = zk.x - zn.x, dy = zk.y - zn.y;
        const float2 xB = make_float2(0.5f * dy, -0.5f * dx);
        if (mag) {
            mag[((long)b * nf + (fa - f0)) * 1024 + k] = sqrtf(xA.x * xA.x + xA.y * xA.y);
            if (has_b) mag[((long)b * nf + (fb - f0)) * 1024 + k] = sqrtf(xB.x * xB.x + xB.y * xB.y);
        }
        if (spec && k <= 512) {   // bins > 512 are the Hermitian mirror (mag still needs them)
            if (fa >= c0 && fa < c1) spec[((long)b * nc + (fa - c0)) * 513 + k] = xA;
            if (has_b && fb >= c0 && fb < c1) spec[((long)b * nc + (fb - c0)) * 513 + k] = xB;
        }
    }
}

}
extern "C" {
void victims_init(void* s) { hipLaunchKernelGGL(fill_table, dim3(4), dim3(256), 0, (hipStream_t)s); }
void run_table(void* out, int blocks, int reps, void* s) { hipLaunchKernelGGL(v_table, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float2*)out, reps); }
void run_valu(void* out, int blocks, int reps, void* s) { hipLaunchKernelGGL(v_valu, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float*)out, reps); }
void run_lds(void* out, int blocks, int reps, void* s) { hipLaunchKernelGGL(v_lds, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float*)out, reps); }
void stft_init(void* s) { hipLaunchKernelGGL(copy::fft_tables_kernel, dim3(4), dim3(256), 0, (hipStream_t)s); }
void run_stft(const void* audio, int B, int n_samples, void* mag, void* spec, void* s) {
    hipLaunchKernelGGL(copy::stft_kernel, dim3((173 - 46 + 1) / 2, B), dim3(256), 0, (hipStream_t)s, (const float*)audio, n_samples, 46, 173, (float*)mag, 89, 117, (float2*)spec);
}
void stft_init2(void* s) { hipLaunchKernelGGL(copy_lds::fft_tables_kernel, dim3(4), dim3(256), 0, (hipStream_t)s); hipLaunchKernelGGL(copy_a16::fft_tables_kernel, dim3(4), dim3(256), 0, (hipStream_t)s); }
void run_stft_lds(const void* audio, int B, int n_samples, void* mag, void* spec, void* s) {
    hipLaunchKernelGGL(copy_lds::stft_kernel, dim3((173 - 46 + 1) / 2, B), dim3(256), 0, (hipStream_t)s, (const float*)audio, n_samples, 46, 173, (float*)mag, 89, 117, (float2*)spec);
}
void run_stft_a16(const void* audio, int B, int n_samples, void* mag, void* spec, void* s) {
    hipLaunchKernelGGL(copy_a16::stft_kernel, dim3((173 - 46 + 1) / 2, B), dim3(256), 0, (hipStream_t)s, (const float*)audio, n_samples, 46, 173, (float*)mag, 89, 117, (float2*)spec);
}
void run_stft_biglds(const void* audio, int B, int n_samples, void* mag, void* spec, int dyn_bytes, void* s) {
    // same kernel, plus a dynamic-LDS request large enough that no contraction workgroup fits on the same CU
    static int set = 0;
    if (!set) { hipFuncSetAttribute((const void*)copy::stft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); set = 1; }
    hipLaunchKernelGGL(copy::stft_kernel, dim3((173 - 46 + 1) / 2, B), dim3(256), dyn_bytes, (hipStream_t)s, (const float*)audio, n_samples, 46, 173, (float*)mag, 89, 117, (float2*)spec);
}
void run_sqrt(void* out, int blocks, int reps, void* s) { hipLaunchKernelGGL(v_sqrt, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float*)out, reps); }
}
