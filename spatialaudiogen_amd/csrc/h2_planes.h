// fp16x2 activation planes ("H2", conv3h.hip): device helpers shared by the passes that write them (p3.hip: the forward's
// batch-norm / merge / pool passes; backward.hip: the batch-norm backward, whose dy feeds the stride-1 data gradients).
// Layout: [C/16][NP][2][16] fp16 of v * 2^ka - 64 B per padded pixel and 16-channel chunk, hi plane then lo plane.
#pragma once
#include "wave_reduce.h"

namespace sagen {

__device__ __forceinline__ float wave_max_f(float v) {              // max over the wavefront (values >= 0), by DPP / permlane swaps
    v = fmaxf(v, dpp_mov<0xB1>(v, v)); v = fmaxf(v, dpp_mov<0x4E>(v, v));
    { float t = dpp_mov<0x104, 0x5>(v, v); t = dpp_mov<0x114, 0xA>(t, v); v = fmaxf(v, t); }
    { float t = dpp_mov<0x108, 0x3>(v, v); t = dpp_mov<0x118, 0xC>(t, v); v = fmaxf(v, t); }
    { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); v = fmaxf(a, b); }
    { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); v = fmaxf(a, b); }
    return v;
}

__device__ __forceinline__ void p3h_store(char* p3, long cstride, long pp, int c8, const float (&v)[8], float sa, unsigned* sat_count) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 hi, lo;
    bool clamped = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float sv = v[k] * sa;
        clamped = clamped || !(fabsf(sv) <= 65000.f);
        const float t = fminf(fmaxf(sv, -65000.f), 65000.f);
        // a NaN / Inf must stay one (fmaxf(NaN, x) = x would launder it into a finite plane value and the loss would stay finite):
        // the hi plane carries an fp16 NaN, every product with it is NaN, and the caller's NaN check (train.py:212) fires
        hi[k] = __builtin_isfinite(sv) ? (_Float16)t : __builtin_bit_cast(_Float16, (unsigned short)0x7e00);
        lo[k] = __builtin_isfinite(sv) ? (_Float16)(t - (float)hi[k]) : (_Float16)0.f;
    }
    char* dst = p3 + (long)(c8 >> 1) * cstride + pp * 64 + (c8 & 1) * 16;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 32) = lo;
    if (clamped && sat_count) atomicAdd(sat_count, 1u);       // (never, unless a value lies 64 x beyond eight standard deviations - or is not finite)
}

// 2^k with bound * 2^k in [512, 1024) (k clamped to +-60); 1 for a zero / non-finite bound
__device__ __forceinline__ float h2_scale_of_bound(float bound) {
    const unsigned b = __builtin_bit_cast(unsigned, bound);
    const int e = (int)((b >> 23) & 0xff);
    if (e == 0 || e == 255) return 1.f;
    return __builtin_bit_cast(float, (unsigned)(127 + max(-60, min(60, 127 + 9 - e))) << 23);
}

}  // namespace sagen
