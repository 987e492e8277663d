// fcm_kernel: skinny fully-connected layers on the exact fp32 matrix instruction, operands straight from memory.
//
// tfw.fully_connected (core.py:43-93) in the bottleneck, the localisation head and fc-feats (model.py:203-256, 287-294): M = batch x 3
// time steps (<= 96 rows) against [K][N] matrices - 6.4 M weights for <video|flow>-fc, 3.1 M for audio-fc, 0.8 M for fc1 / fc-feats.
// These layers are bound by streaming their weights ONCE and by launch latency, not by arithmetic (0.01 - 0.6 GFLOP).  Round 4 ran
// them on the general implicit-GEMM kernels: packed + pre-split filters (1.5 x the bytes), an in-loop operand split, split-K 8..64
// with a reducer launch behind every layer - 14 launches of 6.5 - 17 us for the seven layers of an audio + video forward.  Here:
//   * v_mfma_f32_32x32x2_f32 (exact fp32: no operand split, no planes, no filter pack - the variable is read in its TF layout [K][N]).
//     The B fragment of lane l is W[k + l / 32][n0 + l % 32]: two 128-byte rows per wave load, global -> VGPR, never through LDS;
//   * a workgroup = one 32-column tile x one K slice; its four waves take a quarter of the slice each and are summed through LDS in a
//     fixed order (deterministic); the slices leave as partials [nslices][M][N];
//   * the INPUT rows are assembled from up to eight K ranges (audio-fc reads the six frequency columns of conv5 out of the concat
//     buffer in place); a range may also be a producer's partials, summed + biased + activated by the loader - measured: a layer
//     that re-reduces its input in every column tile reads it 32 times, the reducer launch is cheaper, so the forward does not use it;
//   * two layers reading the same rows share one launch (fc1 + fc-feats).
// LDS: the wave's 32 x M input tile [k][m] (row pitch M + 1: conflict-free column writes, row reads).
#include "fcm.h"
#include "wave_reduce.h"

namespace sagen {

typedef float f32x16m __attribute__((ext_vector_type(16)));

template <int MT>
__global__ __launch_bounds__(256) void fcm_kernel(const FcmDesc d) {
    constexpr int MP = MT * 32, PITCH = MP + 1;
    constexpr int XS = 32 * PITCH;                       // floats per wave
    constexpr int RED = 4 * MP * 33;
    __shared__ float smem[(4 * XS > RED ? 4 * XS : RED)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    // which (job, column tile)
    int j = 0, t = blockIdx.x;
    while (j + 1 < d.njobs && t >= (d.job[j].N + 31) / 32) { t -= (d.job[j].N + 31) / 32; ++j; }
    const FcmJob job = d.job[j];
    const int n0 = t * 32, N = job.N;
    const int ncol = min(n0 + li, N - 1);                // (clamped: columns >= N are computed on a copy of the last one and dropped)
    const int Ks = d.K / d.nslices, Kw = Ks / 4;         // (K % (8 nslices) == 0: fcm_launch)
    const int kbeg = blockIdx.y * Ks + wave * Kw, kend = kbeg + Kw;
    float* const xs = smem + wave * XS;

    f32x16m acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // one tile = 32 k: the 16 weight fragments (k pair kp: rows kt + 2 kp + kk) and this lane's column of the input tile
    // (k = kt + li, rows kk, kk + 2, ..: coalesced along k), ALL issued before anything waits - the loader is latency-bound otherwise
    // (a first version that fetched row after row took 65 us for a 512 x 512 layer)
    auto load_w = [&](int kt, float (&wf)[16]) {
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
            const int k = min(kt + 2 * kp + kk, d.K - 1);          // rows past the wave's range multiply zeros
            wf[kp] = job.w[(long)k * N + ncol];
        }
    };
    auto load_x = [&](int kt, float (&xr)[MP / 2]) {
        const int k = kt + li;
        int si = -1;
        if (k < kend)
            for (int q = 0; q < d.nseg; ++q) si = (k >= d.seg[q].k0 && k < d.seg[q].k1) ? q : si;
        const FcmSeg sg = d.seg[max(si, 0)];
        // every address is clamped into the range and the value selected afterwards: a conditional load would put each of the MP / 2
        // loads into its own basic block, one memory round trip after the other (45 us for a 512 x 512 layer)
        const int kr = min(max(k - sg.k0, 0), sg.k1 - sg.k0 - 1);
        const float* const base = sg.p + kr;
        const bool kok = si >= 0;
        if (sg.nsplit == 0) {
            float t[MP / 2];
#pragma unroll
            for (int r = 0; r < MP / 2; ++r) t[r] = base[(long)(min(kk + 2 * r, d.M - 1) / sg.row_div) * sg.ld];
#pragma unroll
            for (int r = 0; r < MP / 2; ++r) xr[r] = (kok && kk + 2 * r < d.M) ? t[r] : 0.f;
        } else {                                         // a producer's partials: summed, biased, activated here
            const float bias = sg.bias ? sg.bias[kr] : 0.f;
#pragma unroll
            for (int r = 0; r < MP / 2; ++r) xr[r] = bias;
            for (int z = 0; z < sg.nsplit; ++z) {
                float t[MP / 2];
#pragma unroll
                for (int r = 0; r < MP / 2; ++r) t[r] = base[z * sg.zstride + (long)(min(kk + 2 * r, d.M - 1) / sg.row_div) * sg.ld];
#pragma unroll
                for (int r = 0; r < MP / 2; ++r) xr[r] += t[r];
            }
#pragma unroll
            for (int r = 0; r < MP / 2; ++r) xr[r] = (kok && kk + 2 * r < d.M) ? (sg.relu ? fmaxf(xr[r], 0.f) : xr[r]) : 0.f;
        }
    };
    float wf[16], xr[MP / 2];
    load_w(kbeg, wf);
    load_x(kbeg, xr);
    for (int kt = kbeg; kt < kend; kt += 32) {
#pragma unroll
        for (int r = 0; r < MP / 2; ++r) xs[li * PITCH + kk + 2 * r] = xr[r];
        __syncthreads();
        float wn[16], xn[MP / 2];
        const bool more = kt + 32 < kend;                // (uniform over the workgroup: every wave's range has the same length)
        if (more) {                                      // the next tile travels under this tile's matrix instructions
            load_w(kt + 32, wn);
            load_x(kt + 32, xn);
        }
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float a = xs[(2 * kp + kk) * PITCH + i * 32 + li];           // A[m = li][k = kk] of this pair
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[kp], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) wf[kp] = wn[kp];
#pragma unroll
            for (int r = 0; r < MP / 2; ++r) xr[r] = xn[r];
        }
    }
    // ---- the four waves' sums, in a fixed order, as this slice's partial ----
    float* const red = smem;                             // [4][MP][33]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)      // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
            red[(wave * MP + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 33 + li] = acc[i][e];
    __syncthreads();
    float* const out = job.out + (long)blockIdx.y * d.M * N;
    for (int idx = tid; idx < d.M * 32; idx += 256) {
        const int m = idx >> 5, c = idx & 31;
        if (n0 + c < N)
            out[(long)m * N + n0 + c] = ((red[(0 * MP + m) * 33 + c] + red[(1 * MP + m) * 33 + c]) + red[(2 * MP + m) * 33 + c]) + red[(3 * MP + m) * 33 + c];
    }
}

// slices of K: enough workgroups to pull the weights from HBM in a few microseconds (one workgroup streams ~100 - 400 KB), few enough
// that the consumer's loader sums a handful of partials
int fcm_pick_slices(int K, long weights) {
    (void)weights;
    if (K % 8) return 0;
    // a wave's share of K, K / (4 slices), should be at most two tiles of 32 (one memory round trip under the other's matrix
    // instructions): the smallest slice count that gets there, else the largest that divides
    int best = 1;
    for (int ns = 1; ns <= 64; ++ns) {
        if (K % (8 * ns)) continue;
        best = ns;
        if (K / (4 * ns) <= 64) break;
    }
    return best;
}

int fcm_launch(const FcmDesc& d, hipStream_t s) {
    if (cur_group().G > 1) return fail(SAGEN_ERR_UNSUPPORTED, "%s: no grouped launch (common.h: GroupInfo)", __func__);
    if (d.M <= 0 || d.M > FCM_MAX_M || d.njobs < 1 || d.njobs > FCM_MAX_JOBS || d.nseg < 1 || d.nseg > FCM_MAX_SEG || d.nslices < 1)
        return fail(SAGEN_ERR_SHAPE, "fcm: M=%d jobs=%d segments=%d slices=%d", d.M, d.njobs, d.nseg, d.nslices);
    if (d.K % (8 * d.nslices)) return fail(SAGEN_ERR_UNSUPPORTED, "fcm: K=%d must be a multiple of 8 x %d slices", d.K, d.nslices);
    int tiles = 0, cover = 0;
    for (int j = 0; j < d.njobs; ++j) {
        if (!d.job[j].w || !d.job[j].out || d.job[j].N <= 0) return fail(SAGEN_ERR_NULL, "fcm: job %d is incomplete", j);
        tiles += (d.job[j].N + 31) / 32;
    }
    for (int q = 0; q < d.nseg; ++q) {
        const FcmSeg& g = d.seg[q];
        if (!g.p || g.k1 <= g.k0 || g.k0 != cover || g.row_div < 1 || (g.nsplit > 0 && g.zstride <= 0))
            return fail(SAGEN_ERR_SHAPE, "fcm: input range %d is malformed (the ranges must tile [0, K) in order)", q);
        cover = g.k1;
    }
    if (cover != d.K) return fail(SAGEN_ERR_SHAPE, "fcm: the input ranges cover %d of K=%d", cover, d.K);
    const dim3 grid(tiles, d.nslices);
    if (d.M <= 32) hipLaunchKernelGGL(fcm_kernel<1>, grid, dim3(256), 0, s, d);
    else if (d.M <= 64) hipLaunchKernelGGL(fcm_kernel<2>, grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL(fcm_kernel<3>, grid, dim3(256), 0, s, d);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
