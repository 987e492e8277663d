// Empirical lane mapping of ds_read_b64_tr_b16 (gfx950): LDS element i holds the value i; lane l passes the address of elements
// 4l .. 4l+3; the output shows, per destination lane and register element, which source element arrived.
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr16_probe_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
extern "C" int tr16_probe(short* out, void* stream) {
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
