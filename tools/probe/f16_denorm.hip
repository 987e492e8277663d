// Probe (GPU box): does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL operands, or flush them to zero?
// A = one subnormal value everywhere (2^-20), B = 1.0: every output element should be 16 * 2^-20 = 2^-16 = 1.52588e-05.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    const float vals[] = {9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24 smallest subnormal*/, 6.103515625e-05f /*2^-14 smallest normal*/, 3.0517578125e-05f /*2^-15*/};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, 1.0f);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.9g (as f16 %.9g): mfma -> %.9g, expected %.9g  %s\n", v, h[1], h[0], 16.f * h[1], h[0] == 16.f * h[1] ? "subnormal honoured" : "FLUSHED / wrong");
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0f, v);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("b = %.9g: mfma -> %.9g, expected %.9g  %s\n", v, h[0], 16.f * (float)(_Float16)v, h[0] == 16.f * (float)(_Float16)v ? "subnormal honoured" : "FLUSHED / wrong");
    }
    return 0;
}
