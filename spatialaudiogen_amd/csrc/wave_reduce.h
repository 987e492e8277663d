// Wavefront (64-lane) sum reductions on the VALU's cross-lane paths: DPP controls inside a row of 16 lanes, gfx950's
// v_permlane16_swap / v_permlane32_swap across rows.  No LDS crossbar traffic (`__shfl_xor` compiles to ds_bpermute_b32, which
// goes through the LDS unit and waits on lgkmcnt): the per-bin magnitude / energy sums, the evaluation metrics and the ambisonic
// power map (decoder.py:24-28, distance.py:41-52) reduce with these.
//
//   wave_xor_add<M>(v)  = v + v[lane ^ M]                      M in {1, 2, 4, 8, 16, 32}
//   wave_sum<LO, HI>(v) = butterfly over the masks LO <= M < HI (powers of two): every lane ends with the sum of its group
//   wave_sum(v)         = wave_sum<1, 64>
#pragma once
#include <hip/hip_runtime.h>

namespace sagen {

// DPP control words (v_mov_b32_dpp): quad_perm = the permutation itself; row_shl:n lane i reads lane i+n of its row, row_shr:n
// lane i reads lane i-n; a lane whose bank (4 lanes) is not in bank_mask keeps `old`.
template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK_MASK, false));
}

template <int M>
__device__ __forceinline__ float wave_xor_add(float v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane mask must be a power of two below 64");
    if constexpr (M == 1) {
        return v + dpp_mov<0xB1>(v, v);                           // quad_perm:[1,0,3,2]
    } else if constexpr (M == 2) {
        return v + dpp_mov<0x4E>(v, v);                           // quad_perm:[2,3,0,1]
    } else if constexpr (M == 4) {
        float t = dpp_mov<0x104, 0x5>(v, v);                      // row_shl:4 for banks 0, 2 (lanes whose bit 2 is clear read lane + 4)
        t = dpp_mov<0x114, 0xA>(t, v);                            // row_shr:4 for banks 1, 3
        return v + t;
    } else if constexpr (M == 8) {
        float t = dpp_mov<0x108, 0x3>(v, v);                      // row_shl:8 for banks 0, 1
        t = dpp_mov<0x118, 0xC>(t, v);                            // row_shr:8 for banks 2, 3
        return v + t;
    } else if constexpr (M == 16) {
        // swaps the odd rows of a with the even rows of b: a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]  (the clang builtin of this
        // instruction loses its second result with ROCm 7.2's hipcc - both sums came out as a + a - hence the asm)
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        return a + b;
    } else {
        float a = v, b = v;                                       // upper half of a <-> lower half of b: a = [lo lo], b = [hi hi]
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        return a + b;
    }
}

template <int LO = 1, int HI = 64>
__device__ __forceinline__ float wave_sum(float v) {
    if constexpr (LO < HI) {
        return wave_sum<LO * 2, HI>(wave_xor_add<LO>(v));
    } else {
        return v;
    }
}

}  // namespace sagen
