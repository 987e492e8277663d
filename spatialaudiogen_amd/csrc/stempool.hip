// stempool_kernel: the ResNet18 stem of the inference path as ONE kernel - 7x7 stride-2 SAME convolution (resnet.py:133)
// with the batch statistics of its raw output AND the 3x3 stride-2 SAME max-pool (resnet.py:135) - so that the 205 MB raw
// conv output (B = 32) is never written to HBM and read back.
//
// Training-mode batch-norm needs the statistics of the WHOLE batch before any output can be normalised, so the pool cannot be
// applied to relu(bn(y)) inside the conv.  But bn is a per-channel affine map with slope scale = gamma / sqrt(var + eps), whose
// SIGN is the sign of gamma and is known before the conv runs, and relu is monotone:
//     maxpool(relu(scale * y + shift)) = relu(scale * (scale >= 0 ? maxpool(y) : minpool(y)) + shift).
// The kernel therefore pools the RAW output with max or min per channel (by the sign of gamma) and the existing elementwise
// BN + ReLU pass runs on the pooled tensor (1/4 of the pixels).  HBM traffic of stem + pool: 53 (frame) + 51 + 51 + 51 MB instead
// of 53 + 205 + 205 + 51.
//
// Structure (bf16x3 arithmetic as igemm3s2.hip: six bf16 MFMA products per fp32 product, fp32 accumulate):
//  * a workgroup owns 7 x 7 pooled pixels = the 15 x 15 raw outputs they need (one halo row / column is recomputed: 225 / 196);
//    M = 225 raw pixels -> 8 MFMA row tiles of 32, ONE per wave of an 8-wave workgroup (two waves per SIMD: the operand split
//    of one overlaps the MFMAs of the other; with 4 waves x 2 tiles the kernel took 288 us, matrix pipe 28 % busy); N = 64; K = 7 x 8 x 4 = 224 (7 taps down, 7 + 1 zero taps
//    across, 3 + 1 zero channels of the zero-bordered 4-channel frame), 14 K steps of 16;
//  * the WHOLE filter (its three bf16 planes, 86 KB) is resident in LDS, loaded once per workgroup; workgroups are persistent
//    (256 of them, 16 patches each);
//  * the A operand needs no staging at all: K step ks of raw pixel (r, c) is 8 CONTIGUOUS floats of the padded frame - pixels
//    2c + 4(ks&1) + 2g, +1 of row 2r + ks/2 - so every lane loads its own fragment (two 16-byte loads, L1/L2 hits: each input
//    pixel serves ~12 outputs of the patch) and splits it in registers; the loads of step ks+1 fly under the MFMAs of step ks;
//  * epilogue: the raw tile goes to LDS as [pixel][64] fp32; 256 threads pool it (float4 of channels per thread), the owned
//    14 x 14 pixels feed the per-channel (sum, sumsq), accumulated per workgroup in registers and added to the fp64
//    accumulators once at the end.
#include "igemm3_common.h"

namespace sagen {

constexpr int SP_PR = 7;                       // pooled rows / cols per patch
constexpr int SP_CR = 2 * SP_PR + 1;           // raw rows / cols per patch (15)
constexpr int SP_M = SP_CR * SP_CR;            // 225
constexpr int SP_W_BYTES = 14 * 3 * 64 * 32;   // filter planes: [K/16][plane][n][16] bf16
constexpr int SP_CT_BYTES = 256 * 64 * 4;      // raw tile [256][64] fp32
constexpr int SP_THREADS = 512;
constexpr int SP_LDS = SP_W_BYTES + SP_CT_BYTES + 2 * 8 * 64 * 4;

__global__ __launch_bounds__(SP_THREADS, 1) void stempool_kernel(const float* __restrict__ xpad, const __bf16* __restrict__ wplanes,
                                                          const float* __restrict__ gamma, float* __restrict__ pooled,
                                                          double* __restrict__ stats, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;
    float* const ct = reinterpret_cast<float*>(smem + SP_W_BYTES);
    float* const red = reinterpret_cast<float*>(smem + SP_W_BYTES + SP_CT_BYTES);      // [2][8][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;

    // the filter planes, once
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wplanes);
        f32x4* dst = reinterpret_cast<f32x4*>(wl);
        for (int i = tid; i < SP_W_BYTES / 16; i += SP_THREADS) dst[i] = src[i];
    }
    // pooling: this thread's 4 channels, max or min per channel
    const int pch4 = tid & 15;
    bool use_min[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) use_min[k] = gamma[4 * pch4 + k] < 0.f;
    const int sch = tid & 63, spart = tid >> 6;           // statistics: channel, pixel slice
    float ssum = 0.f, ssq = 0.f;
    __syncthreads();

    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};   // hh, hm, mh, hl, lh, mm
    const int npatch = B * 8 * 16;
    // address of this lane's raw pixel (row `li` of the wave's MFMA tile) in patch `patch`
    auto pixel_base = [&](int patch) {
        const int b = patch >> 7, pr = (patch >> 4) & 7, pc = patch & 15;
        const int m = wave * 32 + li;
        int r = m / SP_CR, c = m - r * SP_CR;
        if (m >= SP_M || 14 * pr + r >= 112 || 14 * pc + c >= 224) { r = 0; c = 0; }       // dummy / outside the image: any valid address, dropped later
        return xpad + (((long)b * 229 + 2 * (14 * pr + r)) * 454 + 2 * (14 * pc + c)) * 4 + 8 * g;
    };
    auto kofs = [](int ks) { return ((ks >> 1) * 454 + (ks & 1) * 4) * 4; };
    // operand loads run TWO K steps ahead of the MFMAs (L2 latency under load is longer than one step), and the first two steps
    // of the next patch are issued before the current patch's epilogue
    f32x4 q0[2], q1[2];                          // steps ks, ks+1 in flight
    const float* abase = blockIdx.x < npatch ? pixel_base(blockIdx.x) : xpad;
    q0[0] = *reinterpret_cast<const f32x4*>(abase + kofs(0)); q0[1] = *reinterpret_cast<const f32x4*>(abase + kofs(0) + 4);
    q1[0] = *reinterpret_cast<const f32x4*>(abase + kofs(1)); q1[1] = *reinterpret_cast<const f32x4*>(abase + kofs(1) + 4);
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        const int b = patch >> 7, pr = (patch >> 4) & 7, pc = patch & 15;
        const int R0 = 14 * pr, C0 = 14 * pc;
        const int next_patch = patch + gridDim.x;
        const float* nbase = next_patch < npatch ? pixel_base(next_patch) : abase;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const f32x4 c0 = q0[0], c1 = q0[1];
            q0[0] = q1[0]; q0[1] = q1[1];
            {
                const float* src = ks + 2 < 14 ? abase + kofs(ks + 2) : nbase + kofs(ks + 2 - 14);     // steps 0, 1 of the next patch
                q1[0] = *reinterpret_cast<const f32x4*>(src);
                q1[1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            bf16x8 fa[3], fb[3][2];
            {
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = split_pair(v[2 * k], v[2 * k + 1]);
                    fa[pl] = __builtin_bit_cast(bf16x8, w);
                }
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[pl][j] = *reinterpret_cast<const bf16x8*>(wl + ((ks * 3 + pl) * 64 + j * 32 + li) * 32 + 16 * g);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[tt]], fb[TB[tt]][j], acc[j], 0, 0, 0);
        }
        abase = nbase;
        // raw tile -> LDS [pixel][64].  C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ct[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g) * 64 + j * 32 + li] = acc[j][e];
        __syncthreads();
        // pool 3x3 / 2 (TF SAME: nothing before, one row / column after -> clipped at the image edge)
        for (int p = tid >> 4; p < SP_PR * SP_PR; p += SP_THREADS / 16) {
            const int pr_l = p / SP_PR, pc_l = p - pr_l * SP_PR;
            float4 ext;
            bool first = true;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                if (R0 + 2 * pr_l + dr >= 112) continue;
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    if (C0 + 2 * pc_l + dc >= 224) continue;
                    const float4 v = *reinterpret_cast<const float4*>(ct + ((2 * pr_l + dr) * SP_CR + 2 * pc_l + dc) * 64 + 4 * pch4);
                    if (first) { ext = v; first = false; }
                    else {
                        ext.x = use_min[0] ? fminf(ext.x, v.x) : fmaxf(ext.x, v.x); ext.y = use_min[1] ? fminf(ext.y, v.y) : fmaxf(ext.y, v.y);
                        ext.z = use_min[2] ? fminf(ext.z, v.z) : fmaxf(ext.z, v.z); ext.w = use_min[3] ? fminf(ext.w, v.w) : fmaxf(ext.w, v.w);
                    }
                }
            }
            *reinterpret_cast<float4*>(pooled + (((long)b * 56 + 7 * pr + pr_l) * 112 + 7 * pc + pc_l) * 64 + 4 * pch4) = ext;
        }
        // batch statistics over the 14 x 14 pixels this patch OWNS (the halo row / column belongs to the neighbour)
        for (int q = spart; q < 196; q += SP_THREADS / 64) {
            const int r = q / 14, cc = q - r * 14;
            const float v = ct[(r * SP_CR + cc) * 64 + sch];
            ssum += v;
            ssq = fmaf(v, v, ssq);
        }
        __syncthreads();                       // the tile is rewritten by the next patch
    }
    red[(0 * 8 + spart) * 64 + sch] = ssum;
    red[(1 * 8 + spart) * 64 + sch] = ssq;
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[(which * 8 + k) * 64 + ch];
        atomicAdd(&stats[which * 64 + ch], (double)s);
    }
}

// xpad: zero-bordered [B,229,454,4]; wp: the stem's packed filter (fp32 [64][224] followed by its bf16x3 planes); pooled [B,56,112,64]
int stempool_launch(const float* xpad, const float* wp, const float* gamma, float* pooled, double* stats, int B, hipStream_t s) {
    if (cur_group().G > 1) return fail(SAGEN_ERR_UNSUPPORTED, "%s: no grouped launch (common.h: GroupInfo)", __func__);
    if (!xpad || !wp || !gamma || !pooled || !stats) return fail(SAGEN_ERR_NULL, "stempool: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stempool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
        attr_set = true;
    }
    const __bf16* planes = reinterpret_cast<const __bf16*>(wp + 64 * 224);
    const int npatch = B * 8 * 16;
    hipLaunchKernelGGL(stempool_kernel, dim3(std::min(npatch, 256)), dim3(SP_THREADS), SP_LDS, s, xpad, planes, gamma, pooled, stats, B);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
