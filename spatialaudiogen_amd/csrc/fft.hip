// FFT family: framed STFT + magnitude, and the fused mask -> iSTFT -> overlap-add -> ambisonic mix.
//
// Replaces myutils.stft (myutils.py:119-147) + tf.abs (model.py:178), and the separation tail
// tf.sigmoid / complex mask / myutils.istft (myutils.py:181-211) / crop / decoder einsum
// (model.py:326-347, 421-434) of the reference.
//
// 1024-point complex FFT: Stockham radix-4 autosort, 5 stages, one 256-thread workgroup, LDS
// ping-pong (2 x 8 KB), twiddles from a device table computed once in fp64.
//  * forward: two real frames are packed as re/im of one complex transform and separated with
//    the Hermitian identities (Xa = (Z[k]+conj Z[N-k])/2, Xb = -i (Z[k]-conj Z[N-k])/2).
//  * inverse: real(ifft(M X)) with X Hermitian equals ifft of ((M[k]+M[N-k])/2) X[k]; the decoder
//    sum over tracks is linear, so  sum_j w[o,j] real(ifft(sigmoid(m_j) X)) is computed as ONE
//    real-output transform of (sum_j w[o,j] sigmoid(m_j)) X per (output channel, localisation
//    step) — two of them packed per complex transform — instead of 32 complex iFFTs per frame.
//    (Changes rounding order only; bounded far below the 1e-4 parity budget, tests/test_gpu_*.)
#include "kernels.h"
#include <mutex>

namespace sagen {

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

// Tables are filled once per process+device; the first caller pays one stream synchronisation
// (sagen_create does it, so captured/async forward calls never do).
int fft_tables_ensure(hipStream_t s) {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    SAGEN_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 64 && done[dev]) return SAGEN_OK;
    hipLaunchKernelGGL(fft_tables_kernel, dim3(4), dim3(256), 0, s);
    SAGEN_LAUNCH_CHECK();
    SAGEN_HIP_CHECK(hipStreamSynchronize(s));
    if (dev < 64) done[dev] = true;
    return SAGEN_OK;
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
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ audio_, int n_samples, int f0, int f1,
                                                   float* __restrict__ mag_, int c0, int c1, float2* __restrict__ spec_,
                                                   float* __restrict__ zero_ptr_, int zero_n, const GroupInfo gi) {
    // grouped launch (common.h): the caller's audio holds the groups back to back (gridDim.y windows each); outputs per group
    const float* __restrict__ audio = audio_ + (size_t)blockIdx.z * ((size_t)gridDim.y * n_samples);
    float* __restrict__ mag = SAGEN_GRP(mag_);
    float2* __restrict__ spec = SAGEN_GRP(spec_);
    float* __restrict__ zero_ptr = SAGEN_GRP(zero_ptr_);
    __shared__ float2 bufA[1024], bufB[1024];
    const int tid = threadIdx.x;
    // (the forward's first kernel on this stream also clears a small buffer for it - the maxima words of the decoder's concat
    //  buffers - instead of a fill launch of its own)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < zero_n; i += gridDim.x * gridDim.y * 256) zero_ptr[i] = 0.f;
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

int stft_launch(const float* audio, int B, int n_samples, int f0, int f1, float* mag, int c0, int c1, float* spec,
                hipStream_t s, float* zero_ptr, int zero_n) {
    if (f1 <= f0 || 256L * (f1 - 1) + 1024 > n_samples)
        return fail(SAGEN_ERR_SHAPE, "stft: frames [%d,%d) do not fit %d samples", f0, f1, n_samples);
    if (spec && (c0 < f0 || c1 > f1 || c1 <= c0)) return fail(SAGEN_ERR_SHAPE, "stft: spec frames must lie in [f0,f1)");
    int rc = fft_tables_ensure(s);
    if (rc) return rc;
    const GroupInfo gi = cur_group();
    hipLaunchKernelGGL(stft_kernel, dim3(cdiv(f1 - f0, 2), B, gi.G), dim3(256), 0, s, audio, n_samples, f0, f1, mag, c0, c1,
                       (float2*)spec, zero_ptr, zero_ptr ? zero_n : 0, gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// mask -> iSTFT (per frame) : grid (NF=23 live frames, B).
// Output sample n in [0,4800) of the window <-> frame f, offset p:  n = 256 f + p - 1216
// (myutils.py:196-205 drops 3*256 leading samples; model.py:344-347 crops 448 more).  Only frames
// 1..23 of the 28 masked frames reach the cropped output.  Localisation step of n: n / 1600.
// -----------------------------------------------------------------------------------------
constexpr int MASK_F_LO = 1, MASK_F_HI = 24, MASK_NF = MASK_F_HI - MASK_F_LO;   // frames [1,24)
constexpr int OUT_SHIFT = 1216;   // 768 + 448

template <int NTR>
__global__ __launch_bounds__(256) void mask_istft_kernel(const float* __restrict__ dmask_, long dmask_bstride, int dmask_f0,
                                                         const float2* __restrict__ spec_, const float* __restrict__ coeffs_,
                                                         float* __restrict__ frames_, const float* __restrict__ ebuf_, const GroupInfo gi) {
    const float* __restrict__ dmask = SAGEN_GRP(dmask_);              // grouped launch (common.h): group = blockIdx.z
    const float2* __restrict__ spec = SAGEN_GRP(spec_);
    const float* __restrict__ coeffs = SAGEN_GRP(coeffs_);
    float* __restrict__ frames = SAGEN_GRP(frames_);
    const float* __restrict__ ebuf = SAGEN_GRP(ebuf_);
    __shared__ __attribute__((aligned(16))) float2 fftbuf[2048];      // FFT ping-pong; before that: staging of the mask rows
    float2* const bufA = fftbuf;
    float2* const bufB = fftbuf + 1024;
    __shared__ float E[6][1024];
    __shared__ float2 X[513];
    __shared__ float wl[2][3][NTR];
    const int tid = threadIdx.x;
    const int fi = blockIdx.x, f = MASK_F_LO + fi, b = blockIdx.y;
    const int n_lo = max(256 * f - OUT_SHIFT, 0), n_hi = min(256 * f - OUT_SHIFT + 1023, 4799);
    const int s_lo = n_lo / 1600, s_hi = n_hi / 1600;
    const bool two = s_hi != s_lo;

    for (int i = tid; i < 2 * 3 * NTR; i += 256) {
        const int j = i % NTR, o = (i / NTR) % 3, si = i / (3 * NTR);
        const int st = si ? s_hi : s_lo;
        wl[si][o][j] = coeffs[(((long)b * 3 + st) * 3 + o) * (NTR + 1) + j];
    }
    for (int k = tid; k < 513; k += 256) X[k] = spec[((long)b * 28 + f) * 513 + k];

    // ---- E[c][k] = sum_j w_c[j] * sigmoid(mask[k][j]): the 128 KiB of mask rows of this frame are the kernel's HBM traffic.
    // They are read as ONE contiguous stream (16 bytes per lane, 1 KiB per wave-instruction), parked in LDS (16-byte chunks
    // XOR-swizzled by the row so the row-wise reads below spread over the banks) and consumed by NTR/16 lanes per bin; the next
    // pass's loads are in flight while the current one is reduced.  (The first version let every lane walk its own 128-byte row:
    // 64 cache lines per load instruction, 1.07 TB/s.)
    if (ebuf != nullptr) {
        // the sums were formed in the epilogue of the deconvolution that produced the logits (igemm_epilogue_maskmix): 32 bytes per bin
        const float4* src = reinterpret_cast<const float4*>(ebuf + ((long)b * MASK_NF + fi) * 1024 * 8);
        __syncthreads();                         // X visible
        for (int k = tid; k < 1024; k += 256) {
            const float4 lo = src[2 * k], hi = src[2 * k + 1];
            E[0][k] = lo.x; E[1][k] = lo.y; E[2][k] = lo.z; E[3][k] = lo.w; E[4][k] = hi.x; E[5][k] = hi.y;
        }
        __syncthreads();
    } else {
    constexpr int NC = NTR / 4;                  // 16-byte chunks per mask row
    constexpr int RPP = 4096 / NTR;              // rows per pass (16 KiB of staging)
    constexpr int TPRW = 256 / RPP;              // lanes per row: each takes 16 tracks
    constexpr int NPASS = 1024 / RPP;
    float4* const stage = reinterpret_cast<float4*>(fftbuf);
    const float4* src = reinterpret_cast<const float4*>(dmask + (long)b * dmask_bstride + (long)(f - dmask_f0) * 1024 * NTR);
    float4 nx0 = src[tid], nx1 = src[tid + 256], nx2 = src[tid + 512], nx3 = src[tid + 768];      // (named: an array here ended up in scratch)
    __syncthreads();                             // wl / X visible
    const int row = tid / TPRW, part = tid % TPRW;
    auto park = [&](int i, const float4& v) {
        const int idx = tid + 256 * i, r = idx / NC, ch = idx % NC;
        stage[r * NC + (ch ^ (r & (NC - 1)))] = v;
    };
#pragma unroll 1
    for (int ps = 0; ps < NPASS; ++ps) {
        park(0, nx0); park(1, nx1); park(2, nx2); park(3, nx3);
        __syncthreads();
        const int pn = ps + 1 < NPASS ? ps + 1 : ps;         // (the last pass re-reads its own rows: keeps the loop body branch-free)
        nx0 = src[pn * 1024 + tid]; nx1 = src[pn * 1024 + tid + 256]; nx2 = src[pn * 1024 + tid + 512]; nx3 = src[pn * 1024 + tid + 768];
        float e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = stage[row * NC + ((part * 4 + q) ^ (row & (NC - 1)))];
            const float sg[4] = {1.f / (1.f + expf(-v.x)), 1.f / (1.f + expf(-v.y)), 1.f / (1.f + expf(-v.z)),
                                 1.f / (1.f + expf(-v.w))};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (part * 4 + q) * 4 + u;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    e[o] = fmaf(wl[0][o][j], sg[u], e[o]);
                    if (two) e[3 + o] = fmaf(wl[1][o][j], sg[u], e[3 + o]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            e[c] = wave_sum<1, TPRW>(e[c]);
            if (part == 0) E[c][ps * RPP + row] = e[c];
        }
        __syncthreads();                         // the staging area is rewritten by the next pass (and then by the FFT)
    }
    }

    float* fout = frames + ((long)b * MASK_NF + fi) * 3 * 1024;
    const int npack = two ? 3 : 2;
    for (int pk = 0; pk < npack; ++pk) {
        // pack (ca -> real part, cb -> imaginary part); cb < 0: empty
        int ca, cb;
        if (pk == 0) { ca = 0; cb = 1; }
        else if (pk == 1) { ca = 2; cb = two ? 5 : -1; }
        else { ca = 3; cb = 4; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = tid + 256 * t;
            const int km = (1024 - k) & 1023;
            float2 xk = (k <= 512) ? X[k] : X[km];
            if (k > 512) xk.y = -xk.y;
            const float ma = 0.5f * (E[ca][k] + E[ca][km]);
            const float mb = cb >= 0 ? 0.5f * (E[cb][k] + E[cb][km]) : 0.f;
            // z = ma*xk + i*mb*xk
            bufA[k] = make_float2(ma * xk.x - mb * xk.y, ma * xk.y + mb * xk.x);
        }
        __syncthreads();
        const float2* Y = fft1024<true>(bufA, bufB, tid);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = tid + 256 * t;
            const int n = 256 * f + p - OUT_SHIFT;
            if (n >= 0 && n < 4800) {
                const int st = n / 1600;
                const float2 y = Y[p];
                const int oa = ca % 3, sa = (ca / 3) ? s_hi : s_lo;
                if (st == sa) fout[oa * 1024 + p] = y.x * (1.f / 1024.f);
                if (cb >= 0) {
                    const int ob = cb % 3, sb = (cb / 3) ? s_hi : s_lo;
                    if (st == sb) fout[ob * 1024 + p] = y.y * (1.f / 1024.f);
                }
            }
        }
        __syncthreads();
    }
}

// overlap-add of the 4 covering frames (average, no synthesis window: myutils.py:205) + bias
__global__ __launch_bounds__(256) void ola_mix_kernel(const float* __restrict__ frames_, const float* __restrict__ coeffs_,
                                                      int ntr, float* __restrict__ out_, int B, const GroupInfo gi) {
    const float* __restrict__ frames = SAGEN_GRP(frames_);
    const float* __restrict__ coeffs = SAGEN_GRP(coeffs_);
    float* __restrict__ out = out_ + (size_t)blockIdx.z * ((size_t)B * 4800 * 3);      // the caller's output: the groups back to back
    const long total = (long)B * 4800 * 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % 3);
        long r = i / 3;
        const int n = (int)(r % 4800);
        const int b = (int)(r / 4800);
        const int st = n / 1600;
        float acc = 0.f;
        const int fl = max((n + 193 + 255) / 256, MASK_F_LO), fh = min((n + OUT_SHIFT) / 256, MASK_F_HI - 1);
        for (int f = fl; f <= fh; ++f)
            acc += frames[(((long)b * MASK_NF + (f - MASK_F_LO)) * 3 + o) * 1024 + (n + OUT_SHIFT - 256 * f)];
        out[i] = fmaf(acc, 0.25f, coeffs[(((long)b * 3 + st) * 3 + o) * (ntr + 1) + ntr]);
    }
}

size_t mask_istft_scratch_bytes(int B) { return (size_t)B * MASK_NF * 3 * 1024 * sizeof(float); }

int mask_istft_mix_launch(const float* dmask, long dmask_bstride, int dmask_f0, const float* spec, const float* coeffs,
                          int B, int ntracks, float* out, float* scratch, hipStream_t s, const float* ebuf) {
    int rc = fft_tables_ensure(s);
    if (rc) return rc;
    if (dmask_f0 > MASK_F_LO) return fail(SAGEN_ERR_SHAPE, "mask_istft: dmask must start at frame <= %d", MASK_F_LO);
    const GroupInfo gi = cur_group();
    dim3 grid(MASK_NF, B, gi.G);
    if (ntracks == 32)
        hipLaunchKernelGGL(mask_istft_kernel<32>, grid, dim3(256), 0, s, dmask, dmask_bstride, dmask_f0, (const float2*)spec, coeffs, scratch, ebuf, gi);
    else if (ntracks == 64)
        hipLaunchKernelGGL(mask_istft_kernel<64>, grid, dim3(256), 0, s, dmask, dmask_bstride, dmask_f0, (const float2*)spec, coeffs, scratch, ebuf, gi);
    else if (ntracks == 16)
        hipLaunchKernelGGL(mask_istft_kernel<16>, grid, dim3(256), 0, s, dmask, dmask_bstride, dmask_f0, (const float2*)spec, coeffs, scratch, ebuf, gi);
    else
        return fail(SAGEN_ERR_UNSUPPORTED, "mask_istft: num_sep_tracks=%d (supported: 16, 32, 64)", ntracks);
    SAGEN_LAUNCH_CHECK();
    const long total = (long)B * 4800 * 3;
    hipLaunchKernelGGL(ola_mix_kernel, dim3((int)std::min<long>(cdiv(total, 256), 4096L), 1, gi.G), dim3(256), 0, s, scratch, coeffs,
                       ntracks, out, B, gi);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

// -----------------------------------------------------------------------------------------
// Adjoint of mask -> iSTFT -> overlap-add -> mix (backward of model.py:326-347, myutils.py:181-211, model.py:421-434).
// Forward per (window b, frame f):  out[n,o] = 1/4 sum_f y_f^{o,step(n)}[p] + bias,  y^{o,s} = real(ifft((sum_j w[s,o,j] sigma(m_j)) X)),
// n = 256 f + p - 1216.  With g^{o,s}[p] = 1/4 dL/dout[n,o] [step(n) = s] and G = FFT(g):
//     Q^{o,s}[k]      = 1/N Re(X[k] conj(G^{o,s}[k]))                       (the gradient at the summed mask of (o, s))
//     dL/dm_j[k]      = sigma'(m_j[k]) sum_{o,s} w[s,o,j] Q^{o,s}[k]
//     dL/dw[s,o,j]    = sum_{f,k} sigma(m_j[k]) Q^{o,s}[k]                   (per-frame partials, summed by mask_bwd_finalize)
//     dL/dbias[s,o]   = sum_{n in step s} dL/dout[n,o]
// so the adjoint needs <= 3 packed 1024-point FFTs per frame (as the forward) instead of 2 x 32, and one pass over the mask.
// -----------------------------------------------------------------------------------------
template <int NTR>
__global__ __launch_bounds__(256) void mask_istft_bwd_kernel(const float* __restrict__ dmask, long dmask_bstride, int dmask_f0,
                                                             const float2* __restrict__ spec, const float* __restrict__ coeffs,
                                                             const float* __restrict__ dpred, float* __restrict__ d_dmask,
                                                             long dd_bstride, int dd_f0, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float2 fftbuf[2048];
    float2* const bufA = fftbuf;
    float2* const bufB = fftbuf + 1024;
    __shared__ float Q[6][1024];
    __shared__ float2 X[513];
    __shared__ float wl[6][NTR];
    __shared__ float red[4][6][NTR];
    const int tid = threadIdx.x;
    const int fi = blockIdx.x, f = MASK_F_LO + fi, b = blockIdx.y;
    const int n_lo = max(256 * f - OUT_SHIFT, 0), n_hi = min(256 * f - OUT_SHIFT + 1023, 4799);
    const int s_lo = n_lo / 1600, s_hi = n_hi / 1600;
    const bool two = s_hi != s_lo;

    for (int i = tid; i < 6 * NTR; i += 256) {
        const int j = i % NTR, o = (i / NTR) % 3, si = i / (3 * NTR);
        const int st = si ? s_hi : s_lo;
        wl[si * 3 + o][j] = coeffs[(((long)b * 3 + st) * 3 + o) * (NTR + 1) + j];
    }
    for (int k = tid; k < 513; k += 256) X[k] = spec[((long)b * 28 + f) * 513 + k];
    for (int i = tid; i < 3 * 1024; i += 256) Q[3 + i / 1024][i % 1024] = 0.f;
    __syncthreads();

    const int npack = two ? 3 : 2;
    for (int pk = 0; pk < npack; ++pk) {
        int ca, cb;                                   // same packing as the forward
        if (pk == 0) { ca = 0; cb = 1; }
        else if (pk == 1) { ca = 2; cb = two ? 5 : -1; }
        else { ca = 3; cb = 4; }
        const int oa = ca % 3, sa = (ca / 3) ? s_hi : s_lo;
        const int ob = cb >= 0 ? cb % 3 : 0, sb = (cb >= 0 && cb / 3) ? s_hi : s_lo;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = tid + 256 * t;
            const int n = 256 * f + p - OUT_SHIFT;
            float va = 0.f, vb = 0.f;
            if (n >= 0 && n < 4800) {
                const int st = n / 1600;
                const float* g = dpred + ((long)b * 4800 + n) * 3;
                if (st == sa) va = 0.25f * g[oa];
                if (cb >= 0 && st == sb) vb = 0.25f * g[ob];
            }
            bufA[p] = make_float2(va, vb);
        }
        __syncthreads();
        const float2* Z = fft1024<false>(bufA, bufB, tid);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = tid + 256 * t;
            const int km = (1024 - k) & 1023;
            const float2 zk = Z[k];
            float2 zn = Z[km];
            zn.y = -zn.y;
            const float2 Ga = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
            const float ddx = zk.x - zn.x, ddy = zk.y - zn.y;
            const float2 Gb = make_float2(0.5f * ddy, -0.5f * ddx);
            float2 xk = (k <= 512) ? X[k] : X[km];
            if (k > 512) xk.y = -xk.y;
            Q[ca][k] = (xk.x * Ga.x + xk.y * Ga.y) * (1.f / 1024.f);
            if (cb >= 0) Q[cb][k] = (xk.x * Gb.x + xk.y * Gb.y) * (1.f / 1024.f);
        }
        __syncthreads();
    }

    // one pass over the mask rows of this frame: thread = (16-byte chunk of 4 tracks, row group)
    constexpr int NC = NTR / 4, RG = 256 / NC;
    const int ch = tid % NC, rg = tid / NC;
    float w[6][4], acc[6][4];
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) { w[c][u] = wl[c][4 * ch + u]; acc[c][u] = 0.f; }
    const float4* src = reinterpret_cast<const float4*>(dmask + (long)b * dmask_bstride + (long)(f - dmask_f0) * 1024 * NTR);
    float4* dst = reinterpret_cast<float4*>(d_dmask + (long)b * dd_bstride + (long)(f - dd_f0) * 1024 * NTR);
#pragma unroll 4
    for (int k = rg; k < 1024; k += RG) {
        const float4 v = src[k * NC + ch];
        float q[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) q[c] = Q[c][k];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float sg = 1.f / (1.f + expf(-vv[u]));
            float dm = 0.f;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                dm = fmaf(w[c][u], q[c], dm);
                acc[c][u] = fmaf(sg, q[c], acc[c][u]);
            }
            o[u] = dm * sg * (1.f - sg);
        }
        dst[k * NC + ch] = make_float4(o[0], o[1], o[2], o[3]);
    }
    // dL/dw partials of this frame: reduce over the row groups (lanes sharing a chunk, then the four waves)
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a = acc[c][u];
            a = wave_sum<NC, 64>(a);
            acc[c][u] = a;
        }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < NC) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[wave][c][4 * lane + u] = acc[c][u];
    }
    __syncthreads();
    for (int i = tid; i < 6 * NTR; i += 256) {
        const int c = i / NTR, j = i % NTR;
        partial[(((long)b * MASK_NF + fi) * 6 + c) * NTR + j] = (red[0][c][j] + red[1][c][j]) + (red[2][c][j] + red[3][c][j]);
    }
}

// grid (B, 9): (step s, output channel o) of one window: sums the per-frame partials in frame order (deterministic) and the
// bias gradient; dcoeffs [B*3][ldc], element (b*3 + s)*ldc + o*(ntr+1) + j
__global__ __launch_bounds__(256) void mask_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ dpred,
                                                                int ntr, float* __restrict__ dcoeffs, int ldc) {
    const int b = blockIdx.x, s = blockIdx.y / 3, o = blockIdx.y % 3;
    const int tid = threadIdx.x;
    float a = 0.f;
    for (int n = 1600 * s + tid; n < 1600 * (s + 1); n += 256) a += dpred[((long)b * 4800 + n) * 3 + o];
    __shared__ float red[4];
    a = wave_sum(a);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float* out = dcoeffs + ((long)b * 3 + s) * ldc + o * (ntr + 1);
    if (tid == 0) out[ntr] = (red[0] + red[1]) + (red[2] + red[3]);
    if (tid < ntr) {
        float v = 0.f;
        for (int fi = 0; fi < MASK_NF; ++fi) {
            const int f = MASK_F_LO + fi;
            const int n_lo = max(256 * f - OUT_SHIFT, 0), n_hi = min(256 * f - OUT_SHIFT + 1023, 4799);
            const int s_lo = n_lo / 1600, s_hi = n_hi / 1600;
            const float* pf = partial + (((long)b * MASK_NF + fi) * 6) * ntr;
            if (s_lo == s) v += pf[o * ntr + tid];
            if (s_hi != s_lo && s_hi == s) v += pf[(3 + o) * ntr + tid];
        }
        out[tid] = v;
    }
}

size_t mask_istft_bwd_scratch_floats(int B, int ntracks) { return (size_t)B * MASK_NF * 6 * ntracks; }

int mask_istft_mix_bwd_launch(const float* dmask, long dmask_bstride, int dmask_f0, const float* spec, const float* coeffs,
                              const float* dpred, int B, int ntracks, float* d_dmask, long dd_bstride, int dd_f0, float* dcoeffs,
                              int ldc, float* scratch, hipStream_t s) {
    int rc = fft_tables_ensure(s);
    if (rc) return rc;
    if (dmask_f0 > MASK_F_LO || dd_f0 > MASK_F_LO) return fail(SAGEN_ERR_SHAPE, "mask_istft_bwd: mask buffers must start at frame <= %d", MASK_F_LO);
    if (ldc < 3 * (ntracks + 1)) return fail(SAGEN_ERR_SHAPE, "mask_istft_bwd: ldc=%d too small", ldc);
    dim3 grid(MASK_NF, B);
#define SAGEN_MB(N) hipLaunchKernelGGL(mask_istft_bwd_kernel<N>, grid, dim3(256), 0, s, dmask, dmask_bstride, dmask_f0, (const float2*)spec, \
                                       coeffs, dpred, d_dmask, dd_bstride, dd_f0, scratch)
    if (ntracks == 32) SAGEN_MB(32);
    else if (ntracks == 64) SAGEN_MB(64);
    else if (ntracks == 16) SAGEN_MB(16);
    else return fail(SAGEN_ERR_UNSUPPORTED, "mask_istft_bwd: num_sep_tracks=%d (supported: 16, 32, 64)", ntracks);
#undef SAGEN_MB
    SAGEN_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_bwd_finalize_kernel, dim3(B, 9), dim3(256), 0, s, scratch, dpred, ntracks, dcoeffs, ldc);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
