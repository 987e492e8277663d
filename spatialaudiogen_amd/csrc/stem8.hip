// stem8pool_kernel: the ResNet18 stem for frames that arrive as the uint8 the JPEG decoder produced (sagen_forward_u8) -
// 7x7 stride-2 SAME convolution (resnet.py:133) + the batch statistics of its raw output + the 3x3 stride-2 SAME max-pool
// (resnet.py:135), one kernel, with HALF the matrix work of the float-frame stem (stempool.hip / igemm3s2.hip).
//
// Why half: the feeder normalises a decoded frame as x = u / 255 - 0.5 (myutils.py:88-89), u in {0..255}.  With u' = u - 128,
//     x = (u' + 0.5) / 255      and      conv(W, x) = conv(W, u') / 255 + (0.5 / 255) * sum(W),
// where the zero padding of x is the value u' = -0.5.  Every u' (and -0.5) is EXACTLY representable in bf16 (8 significant bits), so
// the activation operand of the contraction is ONE bf16 plane instead of the three planes (hi, mid, lo) a general fp32 value needs,
// and the fp32-equivalent product is three bf16 MFMA products (u' x W_hi, u' x W_mid, u' x W_lo) instead of six.  Nothing is
// approximated: the operand is exact, the filter keeps its full three-plane split, accumulation is fp32 - the result differs from the
// float-frame kernels only by the rounding of x itself (the float path convolves the ROUNDED float32 x) and of the final scale.
// The operand split disappears with it (the fragment of a lane is a 16-byte load of the plane: no VALU conversion at all).
//
// Structure (as stempool.hip): persistent workgroups of 8 waves, the whole filter (three planes, 86 KB) LDS-resident; a patch is
// 8 x 7 pooled pixels = 17 x 15 raw outputs (one halo row / column recomputed) = 255 of the 256 rows of 8 MFMA row tiles, one per
// wave; K = 7 x 8 x 4 = 224 (seven taps down; seven + one zero tap across; three + one zero channel): 14 K steps of 16; K step ks of
// raw pixel (r, c) is the 8 CONTIGUOUS bf16 of plane row 2r + ks/2 starting at pixel 2c + 4 (ks & 1) + 2g - every lane loads its own
// fragment, seven steps ahead; raw tile -> LDS once, pooled with max or min per channel by the sign of gamma (relu(bn(.)) is
// monotone per channel: stempool.hip), statistics over the 16 x 14 pixels the patch owns.
#include "igemm3_common.h"

namespace sagen {

constexpr int S8_PH = 8, S8_PW = 7;            // pooled rows / cols per patch
constexpr int S8_RH = 2 * S8_PH + 1;           // raw rows per patch (17)
constexpr int S8_RW = 2 * S8_PW + 1;           // raw cols per patch (15)
constexpr int S8_M = S8_RH * S8_RW;            // 255
constexpr int S8_UH = 229, S8_UW = 456;        // plane geometry: 2 + 224 + 3 rows, 2 + 448 + 6 pixels per row (row pitch 3648 B = 16 * 228)
constexpr int S8_W_BYTES = 14 * 3 * 64 * 32;   // filter planes [K/16][plane][n][16] bf16
constexpr int S8_CT_BYTES = 256 * 64 * 4;      // raw tile [256][64] fp32
constexpr int S8_THREADS = 512;
constexpr int S8_D = 7;                        // operand loads in flight per lane (K steps ahead); must divide 14: the ring carries over to the next patch
constexpr int S8_LDS = S8_W_BYTES + S8_CT_BYTES + 2 * 8 * 64 * 4 + 64 * 4;

size_t stem8_plane_bytes(int B) { return (size_t)B * S8_UH * S8_UW * 8; }

// uint8 frames [B,224,448,3] -> the centred plane [B,229,456,4] bf16: u - 128 inside, -0.5 (= x 0) in the border, 0 in channel 3
__global__ __launch_bounds__(256) void stem8_prep_kernel(const unsigned char* __restrict__ x, u32x2* __restrict__ plane, int B) {
    const long total = (long)B * S8_UH * S8_UW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long p = i;
        const int w = (int)(p % S8_UW) - 2; p /= S8_UW;
        const int h = (int)(p % S8_UH) - 2;
        const int b = (int)(p / S8_UH);
        unsigned c0 = 0xBF00u, c1 = 0xBF00u, c2 = 0xBF00u;                    // bf16(-0.5)
        if ((unsigned)h < 224u && (unsigned)w < 448u) {
            const unsigned char* src = x + (((long)b * 224 + h) * 448 + w) * 3;
            c0 = __builtin_bit_cast(unsigned, (float)((int)src[0] - 128)) >> 16;   // |u - 128| <= 128: 8 significant bits, exact
            c1 = __builtin_bit_cast(unsigned, (float)((int)src[1] - 128)) >> 16;
            c2 = __builtin_bit_cast(unsigned, (float)((int)src[2] - 128)) >> 16;
        }
        plane[i] = u32x2{c0 | (c1 << 16), c2};
    }
}

__global__ __launch_bounds__(S8_THREADS, 1) void stem8pool_kernel(const char* __restrict__ plane, const float* __restrict__ wf32,
                                                                  const __bf16* __restrict__ wplanes, const float* __restrict__ gamma,
                                                                  float* __restrict__ pooled, double* __restrict__ stats, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;
    float* const ct = reinterpret_cast<float*>(smem + S8_W_BYTES);
    float* const red = reinterpret_cast<float*>(smem + S8_W_BYTES + S8_CT_BYTES);                 // [2][8][64]
    float* const cb = reinterpret_cast<float*>(smem + S8_W_BYTES + S8_CT_BYTES + 2 * 8 * 64 * 4); // [64]: (0.5 / 255) * sum_k W[n][k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;

    {   // the filter planes, once
        const f32x4* src = reinterpret_cast<const f32x4*>(wplanes);
        f32x4* dst = reinterpret_cast<f32x4*>(wl);
        for (int i = tid; i < S8_W_BYTES / 16; i += S8_THREADS) dst[i] = src[i];
    }
    if (tid < 64) {
        double s = 0.0;
        for (int k = 0; k < 224; ++k) s += (double)wf32[tid * 224 + k];
        cb[tid] = (float)(s * (0.5 / 255.0));
    }
    const int pch4 = tid & 15;
    bool use_min[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) use_min[k] = gamma[4 * pch4 + k] < 0.f;
    const int sch = tid & 63, spart = tid >> 6;
    float ssum = 0.f, ssq = 0.f;
    __syncthreads();
    const float cb0 = cb[li], cb1 = cb[32 + li];

    const int npatch = B * 7 * 16;
    // byte address of this lane's raw pixel (row `li` of the wave's MFMA tile) in patch `patch`, K step 0
    auto pixel_base = [&](int patch) {
        const int b = patch / 112, rem = patch - b * 112;
        const int pr = rem >> 4, pc = rem & 15;
        const int m = wave * 32 + li;
        int r = m / S8_RW, c = m - r * S8_RW;
        if (m >= S8_M || 16 * pr + r >= 112 || 14 * pc + c >= 224) { r = 0; c = 0; }      // dummy / outside the image: a valid address, dropped later
        return plane + (((long)b * S8_UH + 2 * (16 * pr + r)) * S8_UW + 2 * (14 * pc + c)) * 8 + 16 * g;
    };
    auto kofs = [](int ks) { return ((ks >> 1) * S8_UW + (ks & 1) * 4) * 8; };
    // operand loads run S8_D K steps ahead of the MFMAs (step s of a patch lives in q[s % S8_D]); the first steps of the next patch
    // fly under the current patch's epilogue
    static_assert(14 % S8_D == 0, "the prefetch ring must close over a patch");
    bf16x8 q[S8_D];
    const char* abase = blockIdx.x < npatch ? pixel_base(blockIdx.x) : plane;
#pragma unroll
    for (int k = 0; k < S8_D; ++k) q[k] = *reinterpret_cast<const bf16x8*>(abase + kofs(k));
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        const int b = patch / 112, rem = patch - b * 112;
        const int pr = rem >> 4, pc = rem & 15;
        const int R0 = 16 * pr, C0 = 14 * pc;
        const int next_patch = patch + gridDim.x;
        const char* nbase = next_patch < npatch ? pixel_base(next_patch) : abase;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const bf16x8 fa = q[ks % S8_D];
            q[ks % S8_D] = *reinterpret_cast<const bf16x8*>(ks + S8_D < 14 ? abase + kofs(ks + S8_D) : nbase + kofs(ks + S8_D - 14));
            bf16x8 fb[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[pl][j] = *reinterpret_cast<const bf16x8*>(wl + ((ks * 3 + pl) * 64 + j * 32 + li) * 32 + 16 * g);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)                     // u' x W_lo, x W_mid, x W_hi; the two accumulators alternate
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[pl][j], acc[j], 0, 0, 0);
        }
        abase = nbase;
        // raw output = acc / 255 + (0.5 / 255) sum(W)  -> LDS [pixel][64].  C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* row = ct + (wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g) * 64;
            row[li] = fmaf(acc[0][e], 1.f / 255.f, cb0);
            row[32 + li] = fmaf(acc[1][e], 1.f / 255.f, cb1);
        }
        __syncthreads();
        // pool 3x3 / 2 (TF SAME: nothing before, one row / column after -> clipped at the image edge)
        for (int p = tid >> 4; p < S8_PH * S8_PW; p += S8_THREADS / 16) {
            const int pr_l = p / S8_PW, pc_l = p - pr_l * S8_PW;
            float4 ext;
            bool first = true;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                if (R0 + 2 * pr_l + dr >= 112) continue;
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    if (C0 + 2 * pc_l + dc >= 224) continue;
                    const float4 v = *reinterpret_cast<const float4*>(ct + ((2 * pr_l + dr) * S8_RW + 2 * pc_l + dc) * 64 + 4 * pch4);
                    if (first) { ext = v; first = false; }
                    else {
                        ext.x = use_min[0] ? fminf(ext.x, v.x) : fmaxf(ext.x, v.x); ext.y = use_min[1] ? fminf(ext.y, v.y) : fmaxf(ext.y, v.y);
                        ext.z = use_min[2] ? fminf(ext.z, v.z) : fmaxf(ext.z, v.z); ext.w = use_min[3] ? fminf(ext.w, v.w) : fmaxf(ext.w, v.w);
                    }
                }
            }
            *reinterpret_cast<float4*>(pooled + (((long)b * 56 + 8 * pr + pr_l) * 112 + 7 * pc + pc_l) * 64 + 4 * pch4) = ext;
        }
        // batch statistics over the 16 x 14 pixels this patch OWNS (the halo row / column belongs to the neighbour)
        for (int qq = spart; qq < 16 * 14; qq += S8_THREADS / 64) {
            const int r = qq / 14, cc = qq - r * 14;
            const float v = ct[(r * S8_RW + cc) * 64 + sch];
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

int stem8_prep_launch(const unsigned char* x, void* plane, int B, hipStream_t s) {
    if (!x || !plane) return fail(SAGEN_ERR_NULL, "stem8_prep: null argument");
    const long total = (long)B * S8_UH * S8_UW;
    const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 16);
    hipLaunchKernelGGL(stem8_prep_kernel, dim3(grid), dim3(256), 0, s, x, reinterpret_cast<u32x2*>(plane), B);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// plane: stem8_prep's output; wp: the stem's packed filter (fp32 [64][224] followed by its bf16x3 planes); pooled [B,56,112,64] RAW
// (max or min per channel by the sign of gamma: BN + ReLU follow on the pooled tensor); stats: fp64 (sum, sumsq) of the raw output
int stem8pool_launch(const void* plane, const float* wp, const float* gamma, float* pooled, double* stats, int B, hipStream_t s) {
    if (!plane || !wp || !gamma || !pooled || !stats) return fail(SAGEN_ERR_NULL, "stem8pool: null argument");
    static bool attr_set = false;
    if (!attr_set) {
        SAGEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem8pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS));
        attr_set = true;
    }
    const __bf16* planes = reinterpret_cast<const __bf16*>(wp + 64 * 224);
    const int npatch = B * 7 * 16;
    hipLaunchKernelGGL(stem8pool_kernel, dim3(std::min(npatch, 256)), dim3(S8_THREADS), S8_LDS, s, reinterpret_cast<const char*>(plane), wp, planes, gamma,
                       pooled, stats, B);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
