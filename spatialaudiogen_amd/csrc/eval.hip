// On-graph evaluation metrics of the reference (SptAudioGen.evaluation_ops, model.py:110-154):
// per sample and predicted channel  stft-distance (model.py:62-76, myutils.stft_for_loss myutils.py:151-178),
// log-spectral distance on a 1200-point STFT (model.py:78-94, myutils.stft), temporal MSE and SNR
// (model.py:96-108), and the prediction / target power sums.
//
//  * the stft distance needs only mean_f |FFT(hann * (gt - pred))|^2, which by Parseval is
//    sum_n (hann[n] * (gt - pred)[n])^2: no transform (the parity tests compare against the FFT form);
//  * the LSD needs magnitude spectra of gt and pred on 1200-point frames (not a power of two): framed +
//    Hann-windowed rows x a [1200 x 2*601] cos/sin matrix on the igemm kernel (DFT as a contraction), then a
//    log-magnitude reduction that uses the Hermitian symmetry (bins 1..599 count twice).
#include "kernels.h"

namespace sagen {

constexpr int EV_N = 4800;          // samples per window
constexpr int EV_C = 3;             // predicted channels (Y, Z, X)
constexpr int LSD_W = 1200, LSD_HOP = 600, LSD_T = 6, LSD_BINS = 601;
constexpr int SL_W = 2048;          // stft_for_loss window: 2^ceil(log2(1200))

// Wp[n][k], n = part*601 + bin (part 0 = cos, 1 = -sin), k = sample: the packed [N][K] filter layout of igemm
__global__ __launch_bounds__(256) void eval_dft_matrix_kernel(float* __restrict__ wp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2L * LSD_BINS * LSD_W) return;
    const int n = (int)(idx / LSD_W), k = (int)(idx - (long)n * LSD_W);
    const int part = n / LSD_BINS, bin = n - part * LSD_BINS;
    const long prod = ((long)bin * k) % LSD_W;                 // exact angle reduction
    double s, c;
    sincospi(2.0 * (double)prod / (double)LSD_W, &s, &c);
    wp[idx] = part == 0 ? (float)c : (float)(-s);
}

// time-domain metrics: one 256-thread workgroup per (b, c).  ps = [4][B][3]: stft, lsd (filled later), mse, snr;
// pw[2] += per-sample power sums (atomics)
__global__ __launch_bounds__(256) void eval_time_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B,
                                                        float* __restrict__ ps, double* __restrict__ pw) {
    const int b = blockIdx.x / EV_C, c = blockIdx.x % EV_C;
    const float* p = pred + (long)b * EV_N * EV_C + c;
    const float* g = gt + (long)b * EV_N * EV_C + c;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // sum d^2, sum g^2, sum p^2, sum_w (hann d)^2 (3 windows), unused
    for (int n = threadIdx.x; n < EV_N; n += 256) {
        const float pv = p[(long)n * EV_C], gv = g[(long)n * EV_C];
        const float d = gv - pv;
        acc[0] += d * d; acc[1] += gv * gv; acc[2] += pv * pv;
        // stft_for_loss windows: [0,2048), [2048,4096) (overlap 0) and [1024,3072) (overlap 1)  (myutils.py:166-172)
        float w = 0.f;
        if (n < 2 * SL_W) {
            const float h = (float)(0.5 - 0.5 * cospi(2.0 * (double)(n % SL_W) / (double)SL_W));
            w += h * h;
        }
        if (n >= SL_W / 2 && n < SL_W / 2 + SL_W) {
            const float h = (float)(0.5 - 0.5 * cospi(2.0 * (double)(n - SL_W / 2) / (double)SL_W));
            w += h * h;
        }
        acc[3] += w * d * d;
    }
    __shared__ float red[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = acc[k];
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[4];
        for (int k = 0; k < 4; ++k) s[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        const int o = b * EV_C + c;
        ps[0 * B * EV_C + o] = s[3] / 3.f;                                            // mean over the 3 windows of mean_f |X|^2
        ps[2 * B * EV_C + o] = s[0] / (float)EV_N;                                    // temporal mse
        ps[3 * B * EV_C + o] = 10.f * logf((s[1] + 1e-1f) / (s[0] + 1e-1f)) / logf(10.f);   // snr
        atomicAdd(&pw[0], (double)s[2]);
        atomicAdd(&pw[1], (double)s[1]);
    }
}

// rows r = ((src*B + b)*3 + c)*6 + t, src 0 = gt, 1 = pred: frame t of channel c, periodic Hann(1200) in float32
__global__ __launch_bounds__(256) void eval_lsd_frames_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B,
                                                              float* __restrict__ frames) {
    const int r = blockIdx.x;
    const int t = r % LSD_T, c = (r / LSD_T) % EV_C, b = (r / (LSD_T * EV_C)) % B, src = r / (LSD_T * EV_C * B);
    const float* x = (src ? pred : gt) + (long)b * EV_N * EV_C + c;
    for (int n = threadIdx.x; n < LSD_W; n += 256) {
        const float h = (float)(0.5 - 0.5 * cospi(2.0 * (double)n / (double)LSD_W));
        frames[(long)r * LSD_W + n] = x[(long)(t * LSD_HOP + n) * EV_C] * h;
    }
}

// spec [2*B*18][1202] (re | im).  lsd[b,c] = mean_t sqrt(mean_f (P_gt - P_pred)^2), P = 10 log10(|X| + 0.01)
__global__ __launch_bounds__(256) void eval_lsd_reduce_kernel(const float* __restrict__ spec, int B, float* __restrict__ ps) {
    const int b = blockIdx.x / EV_C, c = blockIdx.x % EV_C;
    __shared__ float red[4];
    float lsd = 0.f;
    for (int t = 0; t < LSD_T; ++t) {
        const long rg = (((long)0 * B + b) * EV_C + c) * LSD_T + t, rp = (((long)1 * B + b) * EV_C + c) * LSD_T + t;
        const float* sg = spec + rg * (2 * LSD_BINS);
        const float* sp = spec + rp * (2 * LSD_BINS);
        float acc = 0.f;
        for (int k = threadIdx.x; k < LSD_BINS; k += 256) {
            const float mg = sqrtf(sg[k] * sg[k] + sg[LSD_BINS + k] * sg[LSD_BINS + k]);
            const float mp = sqrtf(sp[k] * sp[k] + sp[LSD_BINS + k] * sp[LSD_BINS + k]);
            const float d = 10.f * (logf(mg + 1e-2f) - logf(mp + 1e-2f)) / logf(10.f);
            acc += ((k == 0 || k == LSD_W / 2) ? 1.f : 2.f) * d * d;           // bins k and 1200-k are equal
        }
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) lsd += sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)LSD_W);
        __syncthreads();
    }
    if (threadIdx.x == 0) ps[1 * B * EV_C + b * EV_C + c] = lsd / (float)LSD_T;
}

// scratch layout (floats): [DFT matrix 1202*1200][frames 2*B*18*1200][spec 2*B*18*1202][pw: 2 doubles]
size_t eval_scratch_floats(int B) {
    return (size_t)2 * LSD_BINS * LSD_W + (size_t)2 * B * EV_C * LSD_T * LSD_W + (size_t)2 * B * EV_C * LSD_T * 2 * LSD_BINS + 16;
}

int eval_init_launch(float* scratch, hipStream_t s) {
    const long total = 2L * LSD_BINS * LSD_W;
    hipLaunchKernelGGL(eval_dft_matrix_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, scratch);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

int eval_metrics_launch(const float* pred, const float* gt, int B, float* ps, double* pw, float* scratch, hipStream_t s) {
    float* wp = scratch;
    float* frames = wp + (size_t)2 * LSD_BINS * LSD_W;
    float* spec = frames + (size_t)2 * B * EV_C * LSD_T * LSD_W;
    SAGEN_HIP_CHECK(hipMemsetAsync(pw, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(eval_time_kernel, dim3(B * EV_C), dim3(256), 0, s, pred, gt, B, ps, pw);
    SAGEN_LAUNCH_CHECK();
    const int rows = 2 * B * EV_C * LSD_T;
    hipLaunchKernelGGL(eval_lsd_frames_kernel, dim3(rows), dim3(256), 0, s, pred, gt, B, frames);
    SAGEN_LAUNCH_CHECK();
    IgemmDesc d;                                   // spec[rows, 1202] = frames[rows, 1200] x DFT
    d.x = frames; d.w = wp; d.y = spec;
    d.M = rows; d.N = 2 * LSD_BINS; d.K = LSD_W; d.Kpad = LSD_W; d.Cin = LSD_W; d.ldx = LSD_W; d.x_bstride = LSD_W;
    d.Cout = d.N; d.ldy = d.N; d.y_rstride = d.N; d.y_bstride = d.N;
    int rc = igemm_launch(d, TILE_AUTO, s);
    if (rc) return rc;
    hipLaunchKernelGGL(eval_lsd_reduce_kernel, dim3(B * EV_C), dim3(256), 0, s, spec, B, ps);
    SAGEN_LAUNCH_CHECK();
    return SAGEN_OK;
}

}  // namespace sagen
