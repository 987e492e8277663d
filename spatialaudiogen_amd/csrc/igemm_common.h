// Pieces shared by the contraction kernels (igemm.hip: exact fp32 MFMA; igemm3.hip: fp32-equivalent bf16x3 MFMA):
// per-row geometry tables, batch-norm coefficient setup, LDS-DMA helpers and the output epilogue.
#pragma once
#include "kernels.h"
#include "h2_planes.h"

namespace sagen {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAX_TAPS = 128;
constexpr unsigned OOB = 0x80000000u;
constexpr int MAX_BN_C = 512;

struct RowInfo {        // per output-grid row of the tile, shared through LDS
    unsigned boff;      // byte offset of x[b, a*in_sh, bb*in_sw, 0] (mod 2^32)
    unsigned nmlo, nmhi;  // INVERTED tap-validity mask (bit t set = tap t reads padding / row invalid)
    int hrem, wrem;     // valid depth-to-space extents
    int pad;
    long rowoff;        // element offset of the output pixel
};

// one LDS-DMA instruction: 64 lanes x 16 B, global (buffer, bounds-checked) -> LDS at `lds` + lane*16.
// (kept out of the kernel template: the builtin silently blocks host-side stub instantiation otherwise)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SAGEN_ABLATE_DMA)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// workgroup barrier that does NOT drain the LDS-DMA queue (hipcc's __syncthreads() would emit vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// ---- per-row geometry + tap table + producer batch-norm coefficients, once per workgroup (ends with a barrier) ----
template <int BM>
__device__ __forceinline__ void igemm_setup(const IgemmDesc& d, int m0, int tid, bool uni, RowInfo* s_row, int* s_tapb,
                                            float (*s_bn)[MAX_BN_C]) {
    // ---- per-row geometry, once per workgroup ----
    const int HgWg = d.Hg * d.Wg;
    if (!uni)
        for (int t = tid; t < d.ntaps; t += 256) {
            const int th = t / d.TW;
            s_tapb[t] = (((th * d.tap_sh + d.tap_h0) * d.Win + ((t - th * d.TW) * d.tap_sw + d.tap_w0)) * d.ldx) * 4;
        }
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        RowInfo ri;
        ri.boff = 0; ri.nmlo = 0xffffffffu; ri.nmhi = 0xffffffffu; ri.hrem = 0; ri.wrem = 0; ri.pad = 0; ri.rowoff = 0;
        if (m < d.M) {
            const int b = m / HgWg;
            const int rem = m - b * HgWg;
            const int ia = rem / d.Wg;
            const int a = d.g_h0 + ia, bb = d.g_w0 + (rem - ia * d.Wg);
            const int hi0 = a * d.in_sh, wi0 = bb * d.in_sw;
            ri.boff = (unsigned)(((long)b * d.x_bstride + ((long)hi0 * d.Win + wi0) * d.ldx) * 4);
            ri.rowoff = (long)b * d.y_bstride + (long)(a * d.dsh) * d.y_rstride + (long)(bb * d.dsw) * d.ldy;
            ri.hrem = d.Hlim - a * d.dsh;
            ri.wrem = d.Wlim - bb * d.dsw;
            if (d.no_bounds) {
                ri.nmlo = 0; ri.nmhi = 0;
            } else {
                unsigned lo = 0, hi = 0;
                for (int t = 0; t < d.ntaps; ++t) {
                    const int th = t / d.TW;
                    const int hh = hi0 + th * d.tap_sh + d.tap_h0, ww = wi0 + (t - th * d.TW) * d.tap_sw + d.tap_w0;
                    const unsigned bad = ((unsigned)hh < (unsigned)d.Hin && (unsigned)ww < (unsigned)d.Win) ? 0u : 1u;
                    if (t < 32) lo |= bad << t; else hi |= bad << (t - 32);
                }
                ri.nmlo = lo; ri.nmhi = hi;
            }
        }
        s_row[r] = ri;
    }
    if (d.bn_in.acc != nullptr) {          // producer's batch statistics -> scale / shift (replaces a finalize launch)
        for (int ch = tid; ch < d.Cin; ch += 256) {
            const double mean = d.bn_in.acc[ch] * d.bn_in.inv_count;
            double var = d.bn_in.acc[d.Cin + ch] * d.bn_in.inv_count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double sc = (double)d.bn_in.gamma[ch] / sqrt(var + (double)d.bn_in.eps);
            s_bn[0][ch] = (float)sc;
            s_bn[1][ch] = (float)((double)d.bn_in.beta[ch] - mean * sc);
        }
    } else if (d.in_scale != nullptr) {
        for (int ch = tid; ch < d.Cin; ch += 256) { s_bn[0][ch] = d.in_scale[ch]; s_bn[1][ch] = d.in_shift[ch]; }
    }
    __syncthreads();

}

// IgemmDesc::amax_out: the wave's maximum of |stored value| into one of H2_AMAX_SLOTS zeroed words (non-negative floats order like their
// bit patterns); the consumer takes the maximum over the slots.  ONE word would be a same-address atomic chain through the L2: the
// 3 936 waves of the spectrogram conv took 43 us instead of 18 with it.
// (fmaxf drops NaNs: a non-finite output does not show in the maximum - the pack pass that reads the tensor meets the value itself and
//  p3h_store keeps it a NaN in the planes)
__device__ __forceinline__ void igemm_publish_amax(float* amax_out, float v) {
    v = wave_max_f(v);
    if ((threadIdx.x & 63) == 0 && v > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(amax_out) + H2_AMAX_STRIDE * ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (H2_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, v));
}

// ---- epilogue: bias / ReLU / depth-to-space scatter, split-K partials, batch-norm statistics ----
// acc: MFMA 32x32 C/D layout per (i, j) sub-tile; `red` = scratch of >= 2 * (BM / WM) * BN floats (the tile ring)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmDesc& d, f32x16 (&acc)[WM / 32][WN / 32], const RowInfo* s_row,
                                               float* red, int m0, int n0, int z, int tid) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, kk = lane >> 5;
    // ---------------- epilogue ----------------
    // C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int dswC = d.dsw * d.Cout;
    const bool to_ws = d.splitk_ws != nullptr;     // raw partials for the split-K / replicate reduce
    float csum[NT], csq[NT];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        csum[j] = 0.f;
        csq[j] = 0.f;
        const int n = n0 + wn * WN + j * 32 + li;
        const bool nok = n < d.N;
        int ry = 0, rx = 0, o = n;
        if (d.dsh * d.dsw > 1) {
            ry = n / dswC;
            const int rem = n - ry * dswC;
            rx = rem / d.Cout;
            o = rem - rx * d.Cout;
        }
        const long coloff = (long)ry * d.y_rstride + (long)rx * d.ldy + o;
        const float bias = (d.bias && nok && !to_ws) ? d.bias[o] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int m = m0 + row;
                float v = acc[i][j][e];
                if (to_ws) {
                    if (nok && m < d.M) {
                        float* q = &d.splitk_ws[((long)z * d.M + m) * d.N + n];
                        if (d.sk_ticket) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (write-through: visible to the other XCDs without an L2 flush)
                        else *q = v;
                    }
                } else {
                    const bool ok = nok && ry < s_row[row].hrem && rx < s_row[row].wrem;
                    if (ok) {
                        csum[j] += v;
                        csq[j] += v * v;
                        v += bias;
                        if (d.relu_out) v = fmaxf(v, 0.f);
                        d.y[s_row[row].rowoff + coloff] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                }
            }
        }
    }
    if (d.amax_out != nullptr && !to_ws) igemm_publish_amax(d.amax_out, amax);
    if (to_ws && d.sk_ticket != nullptr) {
        // ---- last-arriver combine.  The eight XCDs have their own L2s: the partials are written and read with AGENT-scope accesses
        // (write-through stores, loads that never return a stale line) - a release / acquire FENCE pair instead writes back and
        // invalidates the whole L2 of the XCD, which cost the other batches in flight more than the reducer launches saved
        // (2 030 against 2 240 ambisonic-s/s, audio-only 3 200 against 4 800).
        // MEMORY MODEL: the partial stores, the ticket and the partial loads are all RELAXED agent-scope atomics, i.e. there is no
        // release / acquire edge between them in the HIP / LLVM model - the ordering relied on is gfx950's: agent-scope stores are
        // write-through and `s_waitcnt vmcnt(0)` returns only after they have been acknowledged by memory; agent-scope loads bypass
        // the XCD's L2.  That is a property of THIS hardware, not of the language: the path is an opt-in experiment
        // (SAGEN_SK_FUSED=1, measured no faster) and is not part of any supported configuration; the default is the reducer launch ----
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's agent-scope stores have been acknowledged
        __syncthreads();
        const int tile = (m0 / BM) * ((d.N + BN - 1) / BN) + n0 / BN;
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(&d.sk_ticket[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == d.splitk - 1;
            if (s_last) __hip_atomic_store(&d.sk_ticket[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch on this stream
        }
        __syncthreads();
        if (s_last) {
            constexpr int C4 = BN / 4;
            auto ld4 = [](const float* q) {                       // agent-scope loads: never a stale line of this XCD's L2
                return make_float4(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            };
            const long MN = (long)d.M * d.N;
            for (int idx = tid; idx < BM * C4; idx += 256) {
                const int r = idx / C4, m = m0 + r, n = n0 + 4 * (idx - r * C4);
                if (m >= d.M || n >= d.N) continue;
                const float* p = d.splitk_ws + (long)m * d.N + n;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                int zz = 0;
                for (; zz + 4 <= d.splitk; zz += 4, p += 4 * MN) {      // the reducer's association, four partials in flight
                    const float4 t0 = ld4(p), t1 = ld4(p + MN);
                    const float4 t2 = ld4(p + 2 * MN), t3 = ld4(p + 3 * MN);
                    v.x += (t0.x + t1.x) + (t2.x + t3.x); v.y += (t0.y + t1.y) + (t2.y + t3.y);
                    v.z += (t0.z + t1.z) + (t2.z + t3.z); v.w += (t0.w + t1.w) + (t2.w + t3.w);
                }
                for (; zz < d.splitk; ++zz, p += MN) {
                    const float4 t = ld4(p);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                if (d.bias) { v.x += d.bias[n]; v.y += d.bias[n + 1]; v.z += d.bias[n + 2]; v.w += d.bias[n + 3]; }
                if (d.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                for (int rr = 0; rr < d.sk_rep; ++rr) *reinterpret_cast<float4*>(d.y + ((long)m * d.sk_rep + rr) * d.ldy + n) = v;
            }
        }
    }
    if (d.stats != nullptr) {
        // per-tile per-channel partial sums of the raw conv output (pre-bias; BN convs have none)
        __syncthreads();                       // (the K loop already ended with a barrier; kept for clarity)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = wave_xor_add<32>(csum[j]);
            float q = wave_xor_add<32>(csq[j]);
            if (kk == 0) {
                const int col = wn * WN + j * 32 + li;
                red[(0 * WAVES_M + wm) * BN + col] = s;
                red[(1 * WAVES_M + wm) * BN + col] = q;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 256) {
            const int which = t / BN, col = t - which * BN;
            const int n = n0 + col;
            if (n < d.N) {
                float s = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WAVES_M; ++w2) s += red[(which * WAVES_M + w2) * BN + col];
                atomicAdd(&d.stats[(long)which * d.N + n], (double)s);      // fp64 accumulator [2][N], zeroed per forward
            }
        }
    }
}

// ---- epilogue for the plain dense case (no depth-to-space, no split-K partials): the tile goes through LDS one wave-row at a
// time and every output row leaves as 16-byte row-contiguous stores; bias / ReLU / batch-norm statistics in the same pass.
// The element-wise MFMA-layout epilogue above costs ~15k cycles per 128x64 tile (a third of a workgroup's life at K = 576), this
// one ~5.6k (measured on conv3p.hip, which has the same code).  `red` must hold CAP floats; returns false (nothing done) when the
// problem or the capacity does not qualify.  The caller's K loop must have ended with a barrier.
template <int BM, int BN, int WM, int WN, int CAP>
__device__ __forceinline__ bool igemm_epilogue_rows(const IgemmDesc& d, f32x16 (&acc)[WM / 32][WN / 32], const RowInfo* s_row,
                                                    float* red, int n0, int tid) {
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    constexpr int TPR = BN / 4, RPP = 256 / TPR;
    if constexpr (WM % RPP != 0 || WM * BN + 2 * RPP * BN > CAP) {
        return false;
    } else {
        if (d.dsh * d.dsw != 1 || d.splitk_ws != nullptr) return false;
        constexpr int NPASS = WM / RPP;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int li = lane & 31, kk = lane >> 5;
        float* const tile = red;                           // [WM][BN]
        float* const part_sums = red + WM * BN;            // [2][RPP][BN]
        const int c4 = tid % TPR, rg = tid / TPR;
        const int n = n0 + 4 * c4;
        const bool vec_ok = n + 3 < d.N && (d.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.y) & 15) == 0);
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
            bias.x = n < d.N ? d.bias[n] : 0.f; bias.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
            bias.z = n + 2 < d.N ? d.bias[n + 2] : 0.f; bias.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
        float amax = 0.f;
#pragma unroll
        for (int part = 0; part < WAVES_M; ++part) {
            if (wm == part) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e)      // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
                            tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * BN + wn * WN + j * 32 + li] = acc[i][j][e];
            }
            __syncthreads();
            long ro[NPASS];
            bool ok[NPASS];
            float4 tv[NPASS];
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const RowInfo& ri = s_row[part * WM + rg + k * RPP];
                ok[k] = ri.hrem > 0 && ri.wrem > 0;
                ro[k] = ri.rowoff;
            }
#pragma unroll
            for (int k = 0; k < NPASS; ++k) tv[k] = *reinterpret_cast<const float4*>(tile + (rg + k * RPP) * BN + 4 * c4);
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                if (!ok[k]) continue;
                float4 v = tv[k];
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                if (d.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                float* dst = d.y + ro[k] + n;
                if (vec_ok) *reinterpret_cast<float4*>(dst) = v;
                else {
                    if (n < d.N) dst[0] = v.x;
                    if (n + 1 < d.N) dst[1] = v.y;
                    if (n + 2 < d.N) dst[2] = v.z;
                    if (n + 3 < d.N) dst[3] = v.w;
                }
                amax = fmaxf(amax, fmaxf(fmaxf(n < d.N ? fabsf(v.x) : 0.f, n + 1 < d.N ? fabsf(v.y) : 0.f),
                                         fmaxf(n + 2 < d.N ? fabsf(v.z) : 0.f, n + 3 < d.N ? fabsf(v.w) : 0.f)));
            }
            if (part + 1 < WAVES_M) __syncthreads();
        }
        if (d.amax_out != nullptr) igemm_publish_amax(d.amax_out, amax);
        if (d.stats != nullptr) {
            *reinterpret_cast<float4*>(part_sums + (0 * RPP + rg) * BN + 4 * c4) = cs;
            *reinterpret_cast<float4*>(part_sums + (1 * RPP + rg) * BN + 4 * c4) = cq;
            __syncthreads();
            for (int t = tid; t < 2 * BN; t += 256) {
                const int which = t / BN, col = t - which * BN;
                if (n0 + col < d.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < RPP; ++g) sum += part_sums[(which * RPP + g) * BN + col];
                    atomicAdd(&d.stats[(long)which * d.N + n0 + col], (double)sum);
                }
            }
        }
        return true;
    }
}

// ---- fused decoder tail (IgemmDesc::mm_*): the depth-to-space tile of deconv1 goes through LDS one wave-row at a time; a tile
// holds ONE mask frame of ONE window (BM divides the grid row, BN divides dsw*Cout: igemm_tile_ok), so the six weight rows
// (2 localisation steps x 3 outputs x 32 tracks) are loaded once per workgroup.  Two lanes per output pixel, 16 tracks each.
// Frame <-> sample geometry as in fft.hip (mask_istft_kernel): output sample n = 256 f + p - 1216, localisation step n / 1600.
template <int BM, int BN, int WM, int WN, int CAP>
__device__ __forceinline__ void igemm_epilogue_maskmix(const IgemmDesc& d, f32x16 (&acc)[WM / 32][WN / 32], float* red, int m0, int n0,
                                                       int tid) {
    constexpr int NTR = 32;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    constexpr int SUBS = BN / NTR, ITEMS = WM * SUBS * 2;
    static_assert(WM * BN + 7 * NTR <= CAP, "mask-mix staging must fit the tile ring");
    static_assert(ITEMS % 2 == 0 && BN % NTR == 0, "whole output pixels per tile");
    float* const tile = red;                        // [WM][BN]
    float* const wl = red + WM * BN;                // [6][NTR]
    float* const bs = wl + 6 * NTR;                 // [NTR]
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, kk = lane >> 5;
    const int HgWg = d.Hg * d.Wg;
    const int b = m0 / HgWg;
    const int rem = m0 - b * HgWg;
    const int ia = rem / d.Wg;
    const int q0 = d.g_w0 + rem - ia * d.Wg;
    const int dswC = d.dsw * NTR;
    const int ry = n0 / dswC, rx0 = (n0 - ry * dswC) / NTR;
    const int y = (d.g_h0 + ia) * d.dsh + ry, fi = y - d.mm_row0;
    if (m0 >= d.M || y >= d.Hlim || fi < 0 || fi >= d.mm_nf) return;          // (uniform over the workgroup)
    const int f = d.mm_f_lo + fi;
    const int n_lo = max(256 * f - 1216, 0), n_hi = min(256 * f - 1216 + 1023, 4799);
    const int s_lo = n_lo / 1600, s_hi = n_hi / 1600;
    const bool two = s_hi != s_lo;
    for (int i = tid; i < 6 * NTR; i += 256) {
        const int j = i % NTR, o = (i / NTR) % 3, si = i / (3 * NTR);
        wl[i] = d.mm_coeffs[(((long)b * 3 + (si ? s_hi : s_lo)) * 3 + o) * (NTR + 1) + j];
    }
    if (tid < NTR) bs[tid] = d.bias ? d.bias[tid] : 0.f;
#pragma unroll
    for (int part = 0; part < WAVES_M; ++part) {
        if (wm == part) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)      // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
                        tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * BN + wn * WN + j * 32 + li] = acc[i][j][e];
        }
        __syncthreads();
        for (int item = tid; item < ITEMS; item += 256) {
            const int half = item & 1, sub = (item >> 1) % SUBS, r = (item >> 1) / SUBS;
            float e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = *reinterpret_cast<const float4*>(tile + r * BN + sub * NTR + half * 16 + 4 * q4);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = half * 16 + 4 * q4 + u;
                    // sigmoid on the hardware transcendentals: v_exp_f32 (2^x, 1 ulp) and v_rcp_f32 (1 ulp) - ~2e-7 relative against
                    // expf + IEEE division, which cost ~3x the instructions in a pass of 25 M sigmoids per batch (the tile's critical path)
                    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (vv[u] + bs[j])));
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        e[o] = fmaf(wl[o * NTR + j], sg, e[o]);
                        if (two) e[3 + o] = fmaf(wl[(3 + o) * NTR + j], sg, e[3 + o]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) e[c] = wave_xor_add<1>(e[c]);
            const int row = part * WM + r;
            const int x = (q0 + row) * d.dsw + rx0 + sub;
            if (half == 0 && m0 + row < d.M && x < d.Wlim) {
                float4* dst = reinterpret_cast<float4*>(d.mm_out + (((long)b * d.mm_nf + fi) * d.Wlim + x) * 8);
                dst[0] = make_float4(e[0], e[1], e[2], e[3]);
                dst[1] = make_float4(e[4], e[5], 0.f, 0.f);
            }
        }
        if (part + 1 < WAVES_M) __syncthreads();
    }
}

}  // namespace sagen
