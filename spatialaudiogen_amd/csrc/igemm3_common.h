// Helpers shared by the bf16x3 contraction kernels (igemm3.hip, igemm3dw.hip): vector types, the 16-byte buffer load
// and the bf16 operand split.
#pragma once
#include "igemm_common.h"
#include <type_traits>

namespace sagen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
#else
    return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}

// (a, b) -> packed bf16 pair (round to nearest even) and the exact fp32 residuals
__device__ __forceinline__ unsigned split_pair(float& a, float& b) {
    const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    a -= __builtin_bit_cast(float, pk << 16);
    b -= __builtin_bit_cast(float, pk & 0xffff0000u);
    return pk;
}

}  // namespace sagen
