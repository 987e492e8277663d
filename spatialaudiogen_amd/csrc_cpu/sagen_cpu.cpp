// libsagen_cpu.so - the CPU twin of the OP LEVEL of include/sagen.h ("the same header is implemented twice").
//
// Plain C++, one thread, straightforward loops with double accumulation: every op-level entry point of the HIP library
// (sagen_stft_mag, sagen_conv2d, sagen_bn_finalize, sagen_bn_apply_relu, sagen_maxpool3x3s2, sagen_fc, sagen_deconv2d,
// sagen_mask_istft_mix, sagen_power_map[_batched], sagen_assemble_wyzx and their scratch-size queries) with the same
// signatures, layouts and argument meaning, on HOST pointers (the `stream` argument is ignored).  What it is for: the
// op-level parity cases of tests/test_gpu_ops.py run in a container WITHOUT a GPU against the same fp64 checker
// (tests/test_cpu_twin_ops.py), i.e. the header's semantics are pinned independently of the device code.
// What it is NOT: a fallback.  It is never loaded unless SAGEN_LIB names it explicitly, the context / forward / training
// entry points do not exist here (the Python binding refuses them), and it shares no code with the test checker.
// Each function cites the reference it follows, like the header.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sagen.h"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

typedef std::complex<double> cd;

// in-place radix-2 FFT of 1024 points (sign -1 forward, +1 inverse; unnormalised)
void fft1024(cd* a, int sign) {
    const int N = 1024;
    for (int i = 1, j = 0; i < N; ++i) {
        int bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (int len = 2; len <= N; len <<= 1) {
        const double ang = sign * 2.0 * M_PI / len;
        const cd wl(std::cos(ang), std::sin(ang));
        for (int i = 0; i < N; i += len) {
            cd w(1.0, 0.0);
            for (int k = 0; k < len / 2; ++k) {
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
                w *= wl;
            }
        }
    }
}

// TF 'SAME': pad_total = max((ceil(in / s) - 1) * s + k - in, 0), pad_before = pad_total / 2
void same_pad(int in, int k, int s, int* out, int* before) {
    *out = (in + s - 1) / s;
    const int total = std::max((*out - 1) * s + k - in, 0);
    *before = total / 2;
}

}  // namespace

extern "C" {

int sagen_version(void) { return 100; }
const char* sagen_build_info(void) { return "cpu-twin (op level only; plain C++, double accumulation)"; }
const char* sagen_source_digest(void) { return "cpu-twin"; }
const char* sagen_last_error(void) { return g_err; }

/* myutils.stft (myutils.py:119-147) + crop + tf.abs (model.py:166-178): frame t = samples [256 t, 256 t + 1024) x periodic Hann */
int sagen_stft_mag(const float* audio, int batch, int n_samples, int f0, int f1, float* mag, int c0, int c1, float* spec, void*) {
    if (!audio) return fail(SAGEN_ERR_NULL, "sagen_stft_mag: null audio");
    const int nframes = (n_samples / 1024 - 1) * 4;
    if (f0 < 0 || f1 > nframes || (spec && (c0 < 0 || c1 > nframes))) return fail(SAGEN_ERR_SHAPE, "sagen_stft_mag: frames out of range (%d available)", nframes);
    std::vector<double> hann(1024);
    for (int n = 0; n < 1024; ++n) hann[n] = (double)(float)(0.5 - 0.5 * std::cos(2.0 * M_PI / 1024 * n));     // (the reference rounds the window to float32, myutils.py:134)
    std::vector<cd> buf(1024);
    for (int b = 0; b < batch; ++b) {
        const int lo = std::min(mag ? f0 : c0, spec ? c0 : f0), hi = std::max(mag ? f1 : c1, spec ? c1 : f1);
        for (int t = lo; t < hi; ++t) {
            const bool want_m = mag && t >= f0 && t < f1, want_s = spec && t >= c0 && t < c1;
            if (!want_m && !want_s) continue;
            for (int n = 0; n < 1024; ++n) buf[n] = cd((double)audio[(size_t)b * n_samples + 256 * t + n] * hann[n], 0.0);
            fft1024(buf.data(), -1);
            if (want_m)
                for (int k = 0; k < 1024; ++k) mag[((size_t)b * (f1 - f0) + (t - f0)) * 1024 + k] = (float)std::abs(buf[k]);
            if (want_s)
                for (int k = 0; k <= 512; ++k) {
                    float* o = spec + (((size_t)b * (c1 - c0) + (t - c0)) * 513 + k) * 2;
                    o[0] = (float)buf[k].real(); o[1] = (float)buf[k].imag();
                }
        }
    }
    return SAGEN_OK;
}

size_t sagen_conv2d_scratch_bytes(int, int, int, int, int, int, int) { return 256; }
size_t sagen_bn_stats_floats(int, int, int, int cout) { return (size_t)4 * cout; }     /* 2 * cout doubles */

/* tfw.conv_2d (core.py:156-220) = tf.nn.convolution NHWC / HWIO + bias + optional ReLU; input prologue relu(x * scale + shift) */
int sagen_conv2d(const float* x, int batch, int h, int w, int cin, const float* w_hwio, int kh, int kw, int cout, int sh, int sw,
                 int padding, const float* bias, int relu, const float* in_scale, const float* in_shift, float* y, float* bn_stats,
                 void*, size_t, void*) {
    if (!x || !w_hwio || !y) return fail(SAGEN_ERR_NULL, "sagen_conv2d: null argument");
    int ho, wo, pt = 0, pl = 0;
    if (padding) { same_pad(h, kh, sh, &ho, &pt); same_pad(w, kw, sw, &wo, &pl); }
    else { ho = (h - kh) / sh + 1; wo = (w - kw) / sw + 1; }
    double* st = reinterpret_cast<double*>(bn_stats);
    if (st) std::fill(st, st + 2 * cout, 0.0);
    std::vector<double> acc(cout);
    std::vector<float> xin(cin);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < ho; ++i)
            for (int j = 0; j < wo; ++j) {
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int p = 0; p < kh; ++p) {
                    const int hi = i * sh + p - pt;
                    if (hi < 0 || hi >= h) continue;
                    for (int q = 0; q < kw; ++q) {
                        const int wi = j * sw + q - pl;
                        if (wi < 0 || wi >= w) continue;
                        const float* px = x + (((size_t)b * h + hi) * w + wi) * cin;
                        for (int c = 0; c < cin; ++c) xin[c] = in_scale ? std::max(px[c] * in_scale[c] + in_shift[c], 0.f) : px[c];
                        const float* pw = w_hwio + ((size_t)(p * kw + q) * cin) * cout;
                        for (int c = 0; c < cin; ++c) {
                            const double xv = xin[c];
                            if (xv == 0.0) continue;
                            const float* row = pw + (size_t)c * cout;
                            for (int o = 0; o < cout; ++o) acc[o] += xv * (double)row[o];
                        }
                    }
                }
                float* py = y + (((size_t)b * ho + i) * wo + j) * cout;
                for (int o = 0; o < cout; ++o) {
                    if (st) { st[o] += acc[o]; st[cout + o] += acc[o] * acc[o]; }      /* statistics of the RAW output */
                    double v = acc[o] + (bias ? (double)bias[o] : 0.0);
                    if (relu) v = std::max(v, 0.0);
                    py[o] = (float)v;
                }
            }
    return SAGEN_OK;
}

/* contrib batch_norm, is_training (core.py:6,209-210): biased variance, eps as given */
int sagen_bn_finalize(const float* bn_stats, int batch, int hout, int wout, int cout, const float* gamma, const float* beta, float eps,
                      float* scale, float* shift, void*) {
    if (!bn_stats || !gamma || !beta || !scale || !shift) return fail(SAGEN_ERR_NULL, "sagen_bn_finalize: null argument");
    const double* st = reinterpret_cast<const double*>(bn_stats);
    const double n = (double)batch * hout * wout;
    for (int c = 0; c < cout; ++c) {
        const double mean = st[c] / n, var = std::max(st[cout + c] / n - mean * mean, 0.0);
        const double s = (double)gamma[c] / std::sqrt(var + (double)eps);
        scale[c] = (float)s;
        shift[c] = (float)((double)beta[c] - mean * s);
    }
    return SAGEN_OK;
}

/* y = relu(x * scale + shift (+ residual)) (resnet.py:221,235) */
int sagen_bn_apply_relu(const float* x, const float* scale, const float* shift, const float* residual, float* y, int64_t n_pixels, int c, void*) {
    if (!x || !y) return fail(SAGEN_ERR_NULL, "sagen_bn_apply_relu: null argument");
    for (int64_t i = 0; i < n_pixels; ++i)
        for (int k = 0; k < c; ++k) {
            float v = scale ? std::fmaf(x[i * c + k], scale[k], shift[k]) : x[i * c + k];
            if (residual) v += residual[i * c + k];
            y[i * c + k] = std::max(v, 0.f);
        }
    return SAGEN_OK;
}

/* tf.nn.max_pool 3x3 s2 'SAME' (resnet.py:135) of relu(x * scale + shift); scale NULL: of x itself; -inf padding */
int sagen_maxpool3x3s2(const float* x, const float* scale, const float* shift, float* y, int batch, int h, int w, int c, void*) {
    if (!x || !y) return fail(SAGEN_ERR_NULL, "sagen_maxpool3x3s2: null argument");
    int ho, wo, pt, pl;
    same_pad(h, 3, 2, &ho, &pt); same_pad(w, 3, 2, &wo, &pl);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < ho; ++i)
            for (int j = 0; j < wo; ++j)
                for (int k = 0; k < c; ++k) {
                    float m = -INFINITY;
                    for (int p = 0; p < 3; ++p) {
                        const int hi = 2 * i + p - pt;
                        if (hi < 0 || hi >= h) continue;
                        for (int q = 0; q < 3; ++q) {
                            const int wi = 2 * j + q - pl;
                            if (wi < 0 || wi >= w) continue;
                            float v = x[(((size_t)b * h + hi) * w + wi) * c + k];
                            if (scale) v = std::max(std::fmaf(v, scale[k], shift[k]), 0.f);
                            m = std::max(m, v);
                        }
                    }
                    y[(((size_t)b * ho + i) * wo + j) * c + k] = m;
                }
    return SAGEN_OK;
}

size_t sagen_fc_scratch_bytes(int, int, int) { return 256; }
/* tfw.fully_connected (core.py:43-93) */
int sagen_fc(const float* x, int m, int k, const float* w_kn, int n, const float* bias, int relu, float* y, void*, size_t, void*) {
    if (!x || !w_kn || !y) return fail(SAGEN_ERR_NULL, "sagen_fc: null argument");
    std::vector<double> acc(n);
    for (int i = 0; i < m; ++i) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int kk = 0; kk < k; ++kk) {
            const double xv = x[(size_t)i * k + kk];
            const float* row = w_kn + (size_t)kk * n;
            for (int j = 0; j < n; ++j) acc[j] += xv * (double)row[j];
        }
        for (int j = 0; j < n; ++j) {
            double v = acc[j] + (bias ? (double)bias[j] : 0.0);
            y[(size_t)i * n + j] = (float)(relu ? std::max(v, 0.0) : v);
        }
    }
    return SAGEN_OK;
}

size_t sagen_deconv2d_scratch_bytes(int, int, int, int, int, int) { return 256; }
/* tfw.deconv_2d (core.py:96-153) = tf.nn.conv2d_transpose VALID: out[b, i sh + p, j sw + q, o] += x[b, i, j, c] w[p, q, o, c] */
int sagen_deconv2d(const float* x, int batch, int h, int w, int cin, const float* w_hwoi, int kh, int kw, int cout, int sh, int sw,
                   const float* bias, int relu, float* y, void*, size_t, void*) {
    if (!x || !w_hwoi || !y) return fail(SAGEN_ERR_NULL, "sagen_deconv2d: null argument");
    const int ho = h * sh + kh - sh, wo = w * sw + kw - sw;
    std::vector<double> acc((size_t)ho * wo * cout);
    for (int b = 0; b < batch; ++b) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const float* px = x + (((size_t)b * h + i) * w + j) * cin;
                for (int p = 0; p < kh; ++p)
                    for (int q = 0; q < kw; ++q) {
                        double* po = acc.data() + ((size_t)(i * sh + p) * wo + (j * sw + q)) * cout;
                        const float* pw = w_hwoi + (size_t)(p * kw + q) * cout * cin;
                        for (int o = 0; o < cout; ++o) {
                            double s = 0.0;
                            for (int c = 0; c < cin; ++c) s += (double)px[c] * (double)pw[(size_t)o * cin + c];
                            po[o] += s;
                        }
                    }
            }
        for (size_t e = 0; e < acc.size(); ++e) {
            double v = acc[e] + (bias ? (double)bias[e % cout] : 0.0);
            y[(size_t)b * acc.size() + e] = (float)(relu ? std::max(v, 0.0) : v);
        }
    }
    return SAGEN_OK;
}

size_t sagen_mask_istft_mix_scratch_bytes(int) { return 256; }
/* model.py:326-347 (sigmoid mask x STFT), myutils.istft (myutils.py:181-211: plain average of the four overlaps, no synthesis
 * window), crop [448, 5248), decoder sum (model.py:421-434): out[b, n, o] = sum_k w[b, n / 1600, o, k] s_k[n] + bias */
int sagen_mask_istft_mix(const float* dmask, const float* spec, const float* coeffs, int batch, int ntracks, float* ambi_yzx, void*, size_t, void*) {
    if (!dmask || !spec || !coeffs || !ambi_yzx) return fail(SAGEN_ERR_NULL, "sagen_mask_istft_mix: null argument");
    const int NF = 28;
    std::vector<cd> buf(1024);
    std::vector<double> frames((size_t)NF * 1024), sep(4800);
    std::vector<double> out((size_t)4800 * 3);
    for (int b = 0; b < batch; ++b) {
        const float* cf = coeffs + (size_t)b * 3 * 3 * (ntracks + 1);
        for (int n = 0; n < 4800; ++n)
            for (int o = 0; o < 3; ++o) out[(size_t)n * 3 + o] = cf[((n / 1600) * 3 + o) * (ntracks + 1) + ntracks];
        for (int k = 0; k < ntracks; ++k) {
            for (int f = 0; f < NF; ++f) {
                const float* ps = spec + ((size_t)b * NF + f) * 513 * 2;
                const float* pm = dmask + (((size_t)b * NF + f) * 1024) * ntracks + k;
                for (int bin = 0; bin < 1024; ++bin) {
                    const int kb = bin <= 512 ? bin : 1024 - bin;
                    const cd X((double)ps[2 * kb], bin <= 512 ? (double)ps[2 * kb + 1] : -(double)ps[2 * kb + 1]);   // Hermitian mirror
                    const double m = 1.0 / (1.0 + std::exp(-(double)pm[(size_t)bin * ntracks]));
                    buf[bin] = X * m;
                }
                fft1024(buf.data(), +1);
                for (int n = 0; n < 1024; ++n) frames[(size_t)f * 1024 + n] = buf[n].real() / 1024.0;
            }
            /* istft sample q <-> time q + 768 from the start of frame 0; the crop keeps q in [448, 5248) */
            for (int n = 0; n < 4800; ++n) {
                const int t = n + 448 + 768;
                double s = 0.0;
                for (int f = 0; f < NF; ++f) {
                    const int p = t - 256 * f;
                    if (p >= 0 && p < 1024) s += frames[(size_t)f * 1024 + p];
                }
                sep[n] = s / 4.0;
            }
            for (int n = 0; n < 4800; ++n)
                for (int o = 0; o < 3; ++o) out[(size_t)n * 3 + o] += (double)cf[((n / 1600) * 3 + o) * (ntracks + 1) + k] * sep[n];
        }
        for (size_t e = 0; e < out.size(); ++e) ambi_yzx[(size_t)b * out.size() + e] = (float)out[e];
    }
    return SAGEN_OK;
}

/* AmbiDecoder.decode('projection') + RMS (decoder.py:24-28, distance.py:41-52) */
int sagen_power_map(const float* ambi_wyzx, int64_t t, const float* sh, int p, float* rms, void*) {
    if (!ambi_wyzx || !sh || !rms) return fail(SAGEN_ERR_NULL, "sagen_power_map: null argument");
    for (int d = 0; d < p; ++d) {
        double s = 0.0;
        for (int64_t i = 0; i < t; ++i) {
            double v = 0.0;
            for (int c = 0; c < 4; ++c) v += (double)ambi_wyzx[i * 4 + c] * (double)sh[d * 4 + c];
            s += v * v;
        }
        rms[d] = (float)std::sqrt(s / (double)t);
    }
    return SAGEN_OK;
}
int sagen_power_map_batched(const float* ambi_wyzx, int nchunks, int64_t t, const float* sh, int p, float* rms, double*, void* stream) {
    for (int c = 0; c < nchunks; ++c) {
        const int rc = sagen_power_map(ambi_wyzx + (size_t)c * t * 4, t, sh, p, rms + (size_t)c * p, stream);
        if (rc) return rc;
    }
    return SAGEN_OK;
}

/* deploy.py:143-152: W = mono[snd_contx / 2 : snd_contx / 2 + snd_dur], then Y, Z, X */
int sagen_assemble_wyzx(const float* audio, const float* ambi_yzx, float* out_wyzx, int batch, int snd_size, int snd_contx, int snd_dur, void*) {
    if (!audio || !ambi_yzx || !out_wyzx) return fail(SAGEN_ERR_NULL, "sagen_assemble_wyzx: null argument");
    for (int b = 0; b < batch; ++b)
        for (int n = 0; n < snd_dur; ++n) {
            float* o = out_wyzx + ((size_t)b * snd_dur + n) * 4;
            o[0] = audio[(size_t)b * snd_size + snd_contx / 2 + n];
            for (int c = 0; c < 3; ++c) o[1 + c] = ambi_yzx[((size_t)b * snd_dur + n) * 3 + c];
        }
    return SAGEN_OK;
}

}  // extern "C"
