// extern "C" surface of libsagen_hip.so (include/sagen.h).  Thin: argument checking, error codes,
// no exceptions across the boundary.
#include "kernels.h"
#include <new>
#include <cstdlib>
#include <algorithm>

struct sagen_ctx;
int sagen_create_impl(sagen_ctx** out, const sagen_config* cfg, int groups = 1);
int sagen_groups_impl(const sagen_ctx* c);
int sagen_bind_impl(sagen_ctx* c, const sagen_tensor* tensors, int n, void* workspace, size_t workspace_bytes, hipStream_t s);
int sagen_forward_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s);
int sagen_forward_u8_impl(sagen_ctx* c, const float* audio, const uint8_t* video_u8, const float* flow, float* out, hipStream_t s);
void sagen_destroy_impl(sagen_ctx* c);
size_t sagen_workspace_bytes_impl(const sagen_ctx* c);
int sagen_num_variables_impl(const sagen_ctx* c);
int sagen_variable_spec_impl(const sagen_ctx* c, int i, const char** name, int32_t* ndim, int64_t shape[4]);
int sagen_profile_enable_impl(sagen_ctx* c, int on);
int sagen_autotune_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s);
int sagen_plan_describe_impl(sagen_ctx* c, char* buf, size_t buflen);
int sagen_plan_set_impl(sagen_ctx* c, const char* layer, int tile, int splitk);
int sagen_profile_report_impl(sagen_ctx* c, char* buf, size_t buflen);
int sagen_get_intermediate_impl(const sagen_ctx* c, const char* name, const float** data, int32_t* ndim, int64_t shape[4],
                                int64_t* pixel_stride);
int sagen_set_option_impl(sagen_ctx* c, const char* name, int value);
int sagen_counter_impl(sagen_ctx* c, const char* name, uint64_t* value, hipStream_t s);
size_t sagen_train_workspace_bytes_impl(sagen_ctx* c);
int sagen_train_bind_impl(sagen_ctx* c, const sagen_tensor* grads, int n_grads, const sagen_tensor* moving, int n_moving, void* tws,
                          size_t tws_bytes, hipStream_t s);
int sagen_train_step_u8_impl(sagen_ctx* c, const float* audio, const uint8_t* video_u8, const float* flow, const float* target,
                             const float* mask, float* pred_out, double* loss_out, int update_moving, hipStream_t s);
int sagen_train_step_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, const float* target,
                          const float* mask, float* pred_out, double* loss_out, int update_moving, hipStream_t s);
int sagen_train_get_buffer_impl(const sagen_ctx* c, const char* name, const float** data, size_t* n);
int sagen_train_set_grad_events_impl(sagen_ctx* c, const char* const* names, const int32_t* bucket, int n, void* const* events, int n_buckets);
int sagen_train_autotune_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, const float* target,
                              const float* mask, hipStream_t s);

namespace sagen {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

GroupInfo& cur_group() {
    static thread_local GroupInfo gi;
    return gi;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(SAGEN_ERR_WORKSPACE, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(SAGEN_ERR_SHAPE, "internal error: %s", e.what());
    } catch (...) {
        return fail(SAGEN_ERR_SHAPE, "internal error");
    }
}

// packed fp32 filter + its bf16x3 planes (igemm3.hip)
static size_t pk_bytes(long N, long K) { return align_up(packed_split_floats((size_t)N * ((K + 15) / 16 * 16)) * sizeof(float), 256); }

}  // namespace sagen

using namespace sagen;

extern "C" {

int sagen_version(void) { return SAGEN_VERSION; }
#ifndef SAGEN_BUILD_FLAGS
#define SAGEN_BUILD_FLAGS "unknown"
#endif
const char* sagen_build_info(void) { return SAGEN_BUILD_FLAGS; }
const char* sagen_last_error(void) { return err_buf(); }

int sagen_create(sagen_ctx** out, const sagen_config* cfg) {
    return guarded([&] { return sagen_create_impl(out, cfg); });
}
void sagen_destroy(sagen_ctx* ctx) { sagen_destroy_impl(ctx); }
size_t sagen_workspace_bytes(const sagen_ctx* ctx) { return ctx ? sagen_workspace_bytes_impl(ctx) : 0; }
int sagen_num_variables(const sagen_ctx* ctx) { return ctx ? sagen_num_variables_impl(ctx) : fail(SAGEN_ERR_NULL, "null ctx"); }
int sagen_variable_spec(const sagen_ctx* ctx, int i, const char** name, int32_t* ndim, int64_t shape[4]) {
    if (!ctx || !name || !ndim || !shape) return fail(SAGEN_ERR_NULL, "sagen_variable_spec: null argument");
    return sagen_variable_spec_impl(ctx, i, name, ndim, shape);
}
int sagen_bind_weights(sagen_ctx* ctx, const sagen_tensor* tensors, int n, void* workspace, size_t workspace_bytes,
                       void* stream) {
    return guarded([&] { return sagen_bind_impl(ctx, tensors, n, workspace, workspace_bytes, (hipStream_t)stream); });
}
int sagen_create_grouped(sagen_ctx** out, const sagen_config* cfg, int groups) {
    return guarded([&] { return sagen_create_impl(out, cfg, groups); });
}
static int check_groups(const sagen_ctx* ctx, int groups, const char* fn) {
    if (!ctx) return fail(SAGEN_ERR_NULL, "%s: null ctx", fn);
    if (sagen_groups_impl(ctx) != groups)
        return fail(SAGEN_ERR_SHAPE, "%s: %d group(s) asked of a context created for %d (sagen_create_grouped)", fn, groups, sagen_groups_impl(ctx));
    return SAGEN_OK;
}
int sagen_forward(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, float* ambi_yzx, void* stream) {
    return guarded([&] { const int rc = check_groups(ctx, 1, "sagen_forward"); return rc ? rc : sagen_forward_impl(ctx, audio, video, flow, ambi_yzx, (hipStream_t)stream); });
}
int sagen_forward_u8(sagen_ctx* ctx, const float* audio, const uint8_t* video_u8, const float* flow, float* ambi_yzx, void* stream) {
    return guarded([&] { const int rc = check_groups(ctx, 1, "sagen_forward_u8"); return rc ? rc : sagen_forward_u8_impl(ctx, audio, video_u8, flow, ambi_yzx, (hipStream_t)stream); });
}
int sagen_forward_grouped(sagen_ctx* ctx, int groups, const float* audio, const float* video, const float* flow, float* ambi_yzx, void* stream) {
    return guarded([&] { const int rc = check_groups(ctx, groups, "sagen_forward_grouped"); return rc ? rc : sagen_forward_impl(ctx, audio, video, flow, ambi_yzx, (hipStream_t)stream); });
}
int sagen_forward_grouped_u8(sagen_ctx* ctx, int groups, const float* audio, const uint8_t* video_u8, const float* flow, float* ambi_yzx, void* stream) {
    return guarded([&] { const int rc = check_groups(ctx, groups, "sagen_forward_grouped_u8"); return rc ? rc : sagen_forward_u8_impl(ctx, audio, video_u8, flow, ambi_yzx, (hipStream_t)stream); });
}
int sagen_get_intermediate(const sagen_ctx* ctx, const char* name, const float** data, int32_t* ndim, int64_t shape[4],
                           int64_t* pixel_stride) {
    if (!ctx || !name || !data || !ndim || !shape || !pixel_stride) return fail(SAGEN_ERR_NULL, "sagen_get_intermediate: null argument");
    return guarded([&] { return sagen_get_intermediate_impl(ctx, name, data, ndim, shape, pixel_stride); });
}

int sagen_counter(sagen_ctx* ctx, const char* name, uint64_t* value, void* stream) {
    if (!ctx || !name || !value) return fail(SAGEN_ERR_NULL, "sagen_counter: null argument");
    return guarded([&] { return sagen_counter_impl(ctx, name, value, (hipStream_t)stream); });
}
int sagen_set_option(sagen_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return fail(SAGEN_ERR_NULL, "sagen_set_option: null argument");
    return guarded([&] { return sagen_set_option_impl(ctx, name, value); });
}

int sagen_autotune(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, float* ambi_yzx, void* stream) {
    return guarded([&] { return sagen_autotune_impl(ctx, audio, video, flow, ambi_yzx, (hipStream_t)stream); });
}
int sagen_num_tiles(void) { return (int)TILE_AUTO; }
const char* sagen_tile_name(int tile) { return (tile >= 0 && tile < (int)TILE_AUTO) ? igemm_tile_name((IgemmTile)tile) : nullptr; }

int sagen_plan_set(sagen_ctx* ctx, const char* layer, int tile, int splitk) {
    if (!ctx || !layer) return fail(SAGEN_ERR_NULL, "sagen_plan_set: null argument");
    return guarded([&] { return sagen_plan_set_impl(ctx, layer, tile, splitk); });
}
int sagen_plan_describe(sagen_ctx* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf) return fail(SAGEN_ERR_NULL, "sagen_plan_describe: null argument");
    return guarded([&] { return sagen_plan_describe_impl(ctx, buf, buflen); });
}
int sagen_profile_enable(sagen_ctx* ctx, int on) {
    if (!ctx) return fail(SAGEN_ERR_NULL, "sagen_profile_enable: null ctx");
    return sagen_profile_enable_impl(ctx, on);
}
int sagen_profile_report(sagen_ctx* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf) return fail(SAGEN_ERR_NULL, "sagen_profile_report: null argument");
    return guarded([&] { return sagen_profile_report_impl(ctx, buf, buflen); });
}

int sagen_assemble_wyzx(const float* audio, const float* ambi_yzx, float* out_wyzx, int batch, int snd_size, int snd_contx,
                        int snd_dur, void* stream) {
    if (!audio || !ambi_yzx || !out_wyzx) return fail(SAGEN_ERR_NULL, "sagen_assemble_wyzx: null argument");
    if (batch <= 0 || snd_contx / 2 + snd_dur > snd_size) return fail(SAGEN_ERR_SHAPE, "sagen_assemble_wyzx: bad sizes");
    return assemble_wyzx_launch(audio, ambi_yzx, out_wyzx, batch, snd_size, snd_contx, snd_dur, (hipStream_t)stream);
}

int sagen_stft_mag(const float* audio, int batch, int n_samples, int f0, int f1, float* mag, int c0, int c1, float* spec,
                   void* stream) {
    if (!audio || (!mag && !spec)) return fail(SAGEN_ERR_NULL, "sagen_stft_mag: null argument");
    if (batch <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_stft_mag: batch=%d", batch);
    return stft_launch(audio, batch, n_samples, f0, f1, mag, c0, c1, spec, (hipStream_t)stream);
}

size_t sagen_conv2d_scratch_bytes(int batch, int h, int w, int kh, int kw, int cin, int cout) {
    if (cin == 3) return pk_bytes(cout, (long)kh * kw * 4) + align_up((size_t)batch * (h + kh) * (w + kw) * 4 * sizeof(float), 256);
    // 3x3 convs may run on pre-split activation planes (conv3p.hip): room for them behind the packed filter
    // (only where that path can run: below the 2 GiB buffer-addressing limit.  The query cannot see stride / padding, so a caller
    // that sizes its scratch with pk_bytes alone - the pre-P3 formula - is still served, by the fp32-activation kernels.)
    const size_t pb = (kh == 3 && kw == 3 && cin % 16 == 0) ? p3_bytes(batch, h, w, cin) : 0;
    const size_t planes = (pb > 0 && pb < (1UL << 31)) ? align_up(pb, 256) : 0;
    return pk_bytes(cout, (long)kh * kw * cin) + planes;
}

size_t sagen_bn_stats_floats(int batch, int hout, int wout, int cout) {
    (void)batch; (void)hout; (void)wout;
    return (size_t)4 * cout;                  // two fp64 accumulators per channel
}

int sagen_conv2d(const float* x, int batch, int h, int w, int cin, const float* w_hwio, int kh, int kw, int cout, int sh,
                 int sw, int padding, const float* bias, int relu, const float* in_scale, const float* in_shift, float* y,
                 float* bn_stats, void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!x || !w_hwio || !y || !scratch) return fail(SAGEN_ERR_NULL, "sagen_conv2d: null argument");
        if (batch <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || cout <= 0)
            return fail(SAGEN_ERR_SHAPE, "sagen_conv2d: bad dimensions");
        if ((in_scale == nullptr) != (in_shift == nullptr)) return fail(SAGEN_ERR_NULL, "sagen_conv2d: in_scale/in_shift must come together");
        const size_t need_min = cin == 3 ? sagen_conv2d_scratch_bytes(batch, h, w, kh, kw, cin, cout) : pk_bytes(cout, (long)kh * kw * (cin == 1 ? 1 : cin));
        if (scratch_bytes < need_min) return fail(SAGEN_ERR_WORKSPACE, "sagen_conv2d: scratch too small");
        const bool room_for_planes = scratch_bytes >= sagen_conv2d_scratch_bytes(batch, h, w, kh, kw, cin, cout);
        int Hout, Wout, pt = 0, pb = 0, pl = 0, pr = 0;
        if (padding == 1) {
            Hout = cdiv(h, sh); Wout = cdiv(w, sw);
            const int th = std::max((Hout - 1) * sh + kh - h, 0), tw = std::max((Wout - 1) * sw + kw - w, 0);
            pt = th / 2; pb = th - pt; pl = tw / 2; pr = tw - pl;
        } else if (padding == 0) {
            if (h < kh || w < kw) return fail(SAGEN_ERR_SHAPE, "sagen_conv2d: VALID conv larger than input");
            Hout = (h - kh) / sh + 1; Wout = (w - kw) / sw + 1;
        } else {
            return fail(SAGEN_ERR_SHAPE, "sagen_conv2d: padding must be 0 (VALID) or 1 (SAME)");
        }
        float* wp = (float*)scratch;
        IgemmDesc d;
        d.y = y; d.bias = bias; d.relu_out = relu; d.in_scale = in_scale; d.in_shift = in_shift; d.stats = (double*)bn_stats;
        d.M = batch * Hout * Wout; d.N = cout; d.Cout = cout;
        d.Hg = Hout; d.Wg = Wout; d.Hlim = Hout; d.Wlim = Wout;
        d.ldy = cout; d.y_rstride = (long)Wout * cout; d.y_bstride = (long)Hout * Wout * cout;
        d.in_sh = sh; d.in_sw = sw; d.w = wp;
        int rc;
        if (cin == 1) {
            if (padding != 0 || kw % 4 || sw % 4 || in_scale)
                return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d: cin=1 needs VALID padding, kw%%4==0, sw%%4==0, no input BN");
            d.x = x; d.Hin = h; d.Win = w; d.Cin = kw; d.ldx = 1; d.x_bstride = (long)h * w;
            d.ntaps = kh; d.TW = 1; d.tap_sh = 1; d.tap_sw = 0; d.log2Cin = ilog2_exact(kw);
            if (kh > 1 && d.log2Cin < 0) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d: cin=1 needs power-of-two kw");
            d.K = kh * kw; d.Kpad = (d.K + 15) / 16 * 16;
            rc = pack_conv_launch(w_hwio, kh, kw, kw, cout, wp, cout, d.Kpad, s);
        } else if (cin == 3) {
            if (in_scale) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d: cin=3 does not take an input BN");
            float* xp = (float*)((char*)scratch + pk_bytes(cout, (long)kh * kw * 4));
            rc = pad_nhwc3to4_launch(x, xp, batch, h, w, pt, pb, pl, pr, s);
            if (rc) return rc;
            d.x = xp; d.Hin = h + pt + pb; d.Win = w + pl + pr; d.Cin = 4; d.ldx = 4; d.x_bstride = (long)d.Hin * d.Win * 4;
            d.ntaps = kh * kw; d.TW = kw; d.log2Cin = 2;
            d.K = kh * kw * 4; d.Kpad = (d.K + 15) / 16 * 16;
            rc = pack_conv_launch(w_hwio, kh * kw, 3, 4, cout, wp, cout, d.Kpad, s);
        } else {
            if (cin % 4 || (kh * kw > 1 && ilog2_exact(cin) < 2))
                return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d: cin=%d (supported: 1, 3, powers of two >= 4; any multiple of 4 for 1x1)", cin);
            d.x = x; d.Hin = h; d.Win = w; d.Cin = cin; d.ldx = cin; d.x_bstride = (long)h * w * cin;
            d.ntaps = kh * kw; d.TW = kw; d.tap_h0 = -pt; d.tap_w0 = -pl; d.log2Cin = ilog2_exact(cin);
            d.K = kh * kw * cin; d.Kpad = (d.K + 15) / 16 * 16;
            rc = pack_conv_launch(w_hwio, kh * kw, cin, cin, cout, wp, cout, d.Kpad, s);
        }
        if (!rc) rc = pack_split_launch(wp, cout, d.Kpad, s);
        if (rc) return rc;
        d.w_split = 1;
        if (bn_stats) SAGEN_HIP_CHECK(hipMemsetAsync(bn_stats, 0, (size_t)2 * cout * sizeof(double), s));
        static const bool no_p3 = getenv("SAGEN_NO_P3") != nullptr || getenv("SAGEN_FP32_ONLY") != nullptr;
        if (!no_p3 && room_for_planes && sh == 1 && sw == 1 && padding == 1 && cin % 16 == 0 && igemm_p3_eligible(d) &&
            p3_bytes(batch, h, w, cin) < (1UL << 31)) {
            // dense 3x3 stride-1 SAME: one elementwise pass applies the input BN+ReLU (if any) and writes the three bf16
            // planes; the contraction then runs LDS-DMA -> MFMA only (conv3p.hip)
            IgemmDesc e = d;
            e.xp3 = (char*)scratch + pk_bytes(cout, (long)kh * kw * cin);
            e.p3_np = batch * h * (w + 1);
            e.xp3_cstride = (unsigned)((size_t)e.p3_np * 96);
            e.xp3_bytes = (unsigned)p3_bytes(batch, h, w, cin);
            const IgemmTile t = igemm_pick_tile(e);
            if (igemm_tile_p3(t)) {
                rc = p3_pack_launch(x, in_scale, in_shift, BnRef(), nullptr, in_scale ? 1 : 0, nullptr, (void*)e.xp3, batch, h, w, cin, s);
                if (rc) return rc;
                return igemm_launch(e, t, s);
            }
        }
        return igemm_launch(d, TILE_AUTO, s);
    });
}

int sagen_bn_finalize(const float* bn_stats, int batch, int hout, int wout, int cout, const float* gamma, const float* beta,
                      float eps, float* scale, float* shift, void* stream) {
    if (!bn_stats || !gamma || !beta || !scale || !shift) return fail(SAGEN_ERR_NULL, "sagen_bn_finalize: null argument");
    if (((uintptr_t)bn_stats) % 8) return fail(SAGEN_ERR_SHAPE, "sagen_bn_finalize: bn_stats must be 8-byte aligned");
    return bn_finalize_launch((const double*)bn_stats, (long)batch * hout * wout, cout, gamma, beta, eps, scale, shift,
                              (hipStream_t)stream);
}

int sagen_bn_apply_relu(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                        int64_t n_pixels, int c, void* stream) {
    if (!x || !y) return fail(SAGEN_ERR_NULL, "sagen_bn_apply_relu: null argument");
    if ((scale == nullptr) != (shift == nullptr)) return fail(SAGEN_ERR_NULL, "sagen_bn_apply_relu: scale/shift must come together");
    if (n_pixels <= 0 || c <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_bn_apply_relu: bad sizes");
    return bn_apply_relu_launch(x, scale, shift, BnRef(), residual, y, n_pixels, c, (hipStream_t)stream);
}

int sagen_maxpool3x3s2(const float* x, const float* scale, const float* shift, float* y, int batch, int h, int w, int c,
                       void* stream) {
    if (!x || !y) return fail(SAGEN_ERR_NULL, "sagen_maxpool3x3s2: null argument");
    if (batch <= 0 || h <= 0 || w <= 0 || c <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_maxpool3x3s2: bad sizes");
    return maxpool3x3s2_launch(x, scale, shift, BnRef(), y, batch, h, w, c, (hipStream_t)stream);
}

size_t sagen_fc_scratch_bytes(int m, int k, int n) {
    return pk_bytes(n, k) + align_up((size_t)64 * m * n * sizeof(float), 256);
}

int sagen_fc(const float* x, int m, int k, const float* w_kn, int n, const float* bias, int relu, float* y, void* scratch,
             size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!x || !w_kn || !y || !scratch) return fail(SAGEN_ERR_NULL, "sagen_fc: null argument");
        if (m <= 0 || k <= 0 || n <= 0 || k % 4) return fail(SAGEN_ERR_SHAPE, "sagen_fc: bad sizes (k must be a multiple of 4)");
        if (scratch_bytes < sagen_fc_scratch_bytes(m, k, n)) return fail(SAGEN_ERR_WORKSPACE, "sagen_fc: scratch too small");
        float* wp = (float*)scratch;
        float* ws = (float*)((char*)scratch + pk_bytes(n, k));
        IgemmDesc d;
        d.x = x; d.w = wp; d.y = y;
        d.M = m; d.N = n; d.K = k; d.Kpad = (k + 15) / 16 * 16; d.Cin = k; d.ldx = k; d.x_bstride = k;
        d.Cout = n; d.ldy = n; d.y_rstride = n; d.y_bstride = n;
        int rc = pack_conv_launch(w_kn, 1, k, k, n, wp, n, d.Kpad, s);
        if (rc) return rc;
        // low-parallelism rows: split K so the weight stream is spread over the chip
        IgemmTile tile = igemm_pick_tile(d);
        const int bm = tile == TILE_32x128 ? 32 : 64, bn = tile == TILE_32x128 ? 128 : 64;
        const long blocks = (long)cdiv(m, bm) * cdiv(n, bn);
        int sk = 1;
        if (blocks < 384 && d.Kpad / 16 >= 16) sk = (int)std::min<long>({(512 + blocks - 1) / blocks, (long)d.Kpad / 16 / 8, 64L});
        if (sk > 1) {
            d.splitk = sk; d.splitk_ws = ws;
            rc = igemm_launch(d, tile, s);
            if (rc) return rc;
            return splitk_reduce_launch(ws, sk, m, n, bias, relu, y, n, 1, nullptr, s);
        }
        d.bias = bias; d.relu_out = relu;
        return igemm_launch(d, tile, s);
    });
}

size_t sagen_deconv2d_scratch_bytes(int kh, int kw, int cin, int cout, int sh, int sw) {
    return pk_bytes((long)sh * sw * cout, (long)cdiv(kh, sh) * cdiv(kw, sw) * cin);
}

int sagen_deconv2d(const float* x, int batch, int h, int w, int cin, const float* w_hwoi, int kh, int kw, int cout, int sh,
                   int sw, const float* bias, int relu, float* y, void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!x || !w_hwoi || !y || !scratch) return fail(SAGEN_ERR_NULL, "sagen_deconv2d: null argument");
        if (batch <= 0 || h <= 0 || w <= 0 || kh < sh || kw < sw || sh <= 0 || sw <= 0 || cout <= 0)
            return fail(SAGEN_ERR_SHAPE, "sagen_deconv2d: bad dimensions (kernel must be >= stride)");
        if (ilog2_exact(cin) < 2) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_deconv2d: cin=%d must be a power of two >= 4", cin);
        if (scratch_bytes < sagen_deconv2d_scratch_bytes(kh, kw, cin, cout, sh, sw))
            return fail(SAGEN_ERR_WORKSPACE, "sagen_deconv2d: scratch too small");
        const int Hout = h * sh + kh - sh, Wout = w * sw + kw - sw;
        const int nth = cdiv(kh, sh), ntw = cdiv(kw, sw);
        IgemmDesc d;
        d.x = x; d.w = (float*)scratch; d.y = y; d.bias = bias; d.relu_out = relu;
        d.Hg = cdiv(Hout, sh); d.Wg = cdiv(Wout, sw);
        d.M = batch * d.Hg * d.Wg; d.N = sh * sw * cout; d.K = nth * ntw * cin; d.Kpad = (d.K + 15) / 16 * 16;
        d.Hin = h; d.Win = w; d.Cin = cin; d.ldx = cin; d.x_bstride = (long)h * w * cin;
        d.ntaps = nth * ntw; d.TW = ntw; d.tap_sh = -1; d.tap_sw = -1; d.log2Cin = ilog2_exact(cin);
        d.dsh = sh; d.dsw = sw; d.Cout = cout; d.Hlim = Hout; d.Wlim = Wout;
        d.ldy = cout; d.y_rstride = (long)Wout * cout; d.y_bstride = (long)Hout * Wout * cout;
        int rc = pack_deconv_launch(w_hwoi, kh, kw, cout, cin, sh, sw, (float*)scratch, d.N, d.Kpad, s);
        if (!rc) rc = pack_split_launch((float*)scratch, d.N, d.Kpad, s);
        if (rc) return rc;
        d.w_split = 1;
        return igemm_launch(d, TILE_AUTO, s);
    });
}

size_t sagen_mask_istft_mix_scratch_bytes(int batch) { return mask_istft_scratch_bytes(batch); }

int sagen_mask_istft_mix(const float* dmask, const float* spec, const float* coeffs, int batch, int ntracks, float* ambi_yzx,
                         void* scratch, size_t scratch_bytes, void* stream) {
    if (!dmask || !spec || !coeffs || !ambi_yzx || !scratch) return fail(SAGEN_ERR_NULL, "sagen_mask_istft_mix: null argument");
    if (batch <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_mask_istft_mix: batch=%d", batch);
    if (scratch_bytes < mask_istft_scratch_bytes(batch)) return fail(SAGEN_ERR_WORKSPACE, "sagen_mask_istft_mix: scratch too small");
    return mask_istft_mix_launch(dmask, 28L * 1024 * ntracks, 0, spec, coeffs, batch, ntracks, ambi_yzx, (float*)scratch,
                                 (hipStream_t)stream);
}

size_t sagen_eval_scratch_bytes(int batch) { return batch > 0 ? eval_scratch_floats(batch) * sizeof(float) : 0; }

int sagen_eval_init(void* scratch, size_t scratch_bytes, int batch, void* stream) {
    if (!scratch) return fail(SAGEN_ERR_NULL, "sagen_eval_init: null scratch");
    if (batch <= 0 || scratch_bytes < sagen_eval_scratch_bytes(batch)) return fail(SAGEN_ERR_WORKSPACE, "sagen_eval_init: scratch too small");
    return eval_init_launch((float*)scratch, (hipStream_t)stream);
}

int sagen_eval_metrics(const float* pred_yzx, const float* target_yzx, int batch, float* per_sample, double* power_sums,
                       void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        if (!pred_yzx || !target_yzx || !per_sample || !power_sums || !scratch) return fail(SAGEN_ERR_NULL, "sagen_eval_metrics: null argument");
        if (batch <= 0 || scratch_bytes < sagen_eval_scratch_bytes(batch)) return fail(SAGEN_ERR_WORKSPACE, "sagen_eval_metrics: scratch too small");
        if (((uintptr_t)power_sums) % 8) return fail(SAGEN_ERR_SHAPE, "sagen_eval_metrics: power_sums must be 8-byte aligned");
        return eval_metrics_launch(pred_yzx, target_yzx, batch, per_sample, power_sums, (float*)scratch, (hipStream_t)stream);
    });
}

int sagen_power_map(const float* ambi_wyzx, int64_t t, const float* sh, int p, float* rms, void* stream) {
    if (!ambi_wyzx || !sh || !rms) return fail(SAGEN_ERR_NULL, "sagen_power_map: null argument");
    if (t <= 0 || p <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_power_map: bad sizes");
    return power_map_launch(ambi_wyzx, t, sh, p, rms, (hipStream_t)stream);
}

int sagen_power_map_batched(const float* ambi_wyzx, int nchunks, int64_t t, const float* sh, int p, float* rms, double* moments,
                            void* stream) {
    if (!ambi_wyzx || !sh || !rms || !moments) return fail(SAGEN_ERR_NULL, "sagen_power_map_batched: null argument");
    if (nchunks <= 0 || t <= 0 || p <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_power_map_batched: bad sizes");
    if (((uintptr_t)moments) % 8 || ((uintptr_t)ambi_wyzx) % 16 || ((uintptr_t)sh) % 16)
        return fail(SAGEN_ERR_SHAPE, "sagen_power_map_batched: ambi / sh must be 16-byte aligned, moments 8-byte aligned");
    return power_map_batched_launch(ambi_wyzx, nchunks, t, sh, p, rms, moments, (hipStream_t)stream);
}

int sagen_stft_loss_grad(const float* pred_yzx, const float* target_yzx, const float* mask, int batch, float* grad, double* loss,
                         void* stream) {
    if (!pred_yzx || !target_yzx || (!grad && !loss)) return fail(SAGEN_ERR_NULL, "sagen_stft_loss_grad: null argument");
    if (batch <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_stft_loss_grad: batch=%d", batch);
    if (loss && ((uintptr_t)loss) % 8) return fail(SAGEN_ERR_SHAPE, "sagen_stft_loss_grad: loss must be 8-byte aligned");
    return stft_loss_grad_launch(pred_yzx, target_yzx, mask, batch, grad, loss, (hipStream_t)stream);
}

int sagen_adam_update(float* params, const float* grads, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                      float epsilon, float grad_scale, void* stream) {
    if (!params || !grads || !m || !v) return fail(SAGEN_ERR_NULL, "sagen_adam_update: null argument");
    if (n <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_adam_update: n=%ld", (long)n);
    if ((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)m) | ((uintptr_t)v)) % 16)
        return fail(SAGEN_ERR_SHAPE, "sagen_adam_update: buckets must be 16-byte aligned");
    return adam_update_launch(params, grads, m, v, n, lr_t, beta1, beta2, epsilon, grad_scale, (hipStream_t)stream);
}


/* ---- training step (train_model.hip) ---- */
size_t sagen_train_workspace_bytes(sagen_ctx* ctx) { return ctx ? sagen_train_workspace_bytes_impl(ctx) : 0; }
int sagen_train_bind(sagen_ctx* ctx, const sagen_tensor* grads, int n_grads, const sagen_tensor* moving, int n_moving,
                     void* train_workspace, size_t train_workspace_bytes, void* stream) {
    return guarded([&] { return sagen_train_bind_impl(ctx, grads, n_grads, moving, n_moving, train_workspace, train_workspace_bytes, (hipStream_t)stream); });
}
int sagen_train_step(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, const float* target_yzx,
                     const float* mask, float* pred_yzx, double* loss, int update_moving_averages, void* stream) {
    if (loss && ((uintptr_t)loss) % 8) return fail(SAGEN_ERR_SHAPE, "sagen_train_step: loss must be 8-byte aligned");
    return guarded([&] { return sagen_train_step_impl(ctx, audio, video, flow, target_yzx, mask, pred_yzx, loss, update_moving_averages, (hipStream_t)stream); });
}
int sagen_train_step_u8(sagen_ctx* ctx, const float* audio, const uint8_t* video_u8, const float* flow, const float* target_yzx,
                        const float* mask, float* pred_yzx, double* loss, int update_moving_averages, void* stream) {
    if (loss && ((uintptr_t)loss) % 8) return fail(SAGEN_ERR_SHAPE, "sagen_train_step_u8: loss must be 8-byte aligned");
    return guarded([&] { return sagen_train_step_u8_impl(ctx, audio, video_u8, flow, target_yzx, mask, pred_yzx, loss, update_moving_averages, (hipStream_t)stream); });
}
int sagen_train_autotune(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, const float* target_yzx,
                         const float* mask, void* stream) {
    return guarded([&] { return sagen_train_autotune_impl(ctx, audio, video, flow, target_yzx, mask, (hipStream_t)stream); });
}
int sagen_train_set_grad_events(sagen_ctx* ctx, const char* const* names, const int32_t* bucket, int n, void* const* events, int n_buckets) {
    if (!ctx) return fail(SAGEN_ERR_NULL, "sagen_train_set_grad_events: null ctx");
    return guarded([&] { return sagen_train_set_grad_events_impl(ctx, names, bucket, n, events, n_buckets); });
}

int sagen_train_get_buffer(const sagen_ctx* ctx, const char* name, const float** data, size_t* n_floats) {
    if (!ctx || !name || !data || !n_floats) return fail(SAGEN_ERR_NULL, "sagen_train_get_buffer: null argument");
    return guarded([&] { return sagen_train_get_buffer_impl(ctx, name, data, n_floats); });
}

/* ---- backward, op level ---- */
size_t sagen_wgrad_scratch_bytes(int kh, int kw, int cg, int cd) { return align_up((size_t)64 * kh * kw * cg * cd * sizeof(float), 256); }

int sagen_wgrad(const float* g, int batch, int hg, int wg, int cg, const float* d, int hd, int wd, int cd, int kh, int kw, int sh, int sw,
                int h0, int w0, float* dw, void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        if (!g || !d || !dw) return fail(SAGEN_ERR_NULL, "sagen_wgrad: null argument");
        if (cd % 4) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_wgrad: cd=%d must be a multiple of 4 at the op level (dense rows)", cd);
        WgradDesc w;
        w.g = g; w.d = d; w.out = dw; w.B = batch; w.Hd = hd; w.Wd = wd; w.HG = hg; w.WG = wg; w.ldg = cg; w.Cg = cg; w.ldd = cd; w.Cd = cd;
        w.g_rstride = (unsigned)((long)wg * cg); w.g_bstride = (unsigned)((long)hg * wg * cg);
        w.d_rstride = (unsigned)((long)wd * cd); w.d_bstride = (unsigned)((long)hd * wd * cd);
        w.sh = sh; w.sw = sw; w.TH = kh; w.TW = kw; w.h0 = h0; w.w0 = w0;
        w.ws = (float*)scratch;
        w.splitk = scratch ? wgrad_pick_splitk(w, scratch_bytes / sizeof(float)) : 1;
        return wgrad_launch(w, (hipStream_t)stream);
    });
}

size_t sagen_conv2d_bwd_data_scratch_bytes(int kh, int kw, int cin, int cout, int sh, int sw) {
    const long n = (sh == 1 && sw == 1) ? cin : (long)sh * sw * cin;
    const long k = (sh == 1 && sw == 1) ? (long)kh * kw * cout : (long)cdiv(kh, sh) * cdiv(kw, sw) * cout;
    return pk_bytes(n, k);
}

int sagen_conv2d_bwd_data(const float* dy, int batch, int hout, int wout, int cout, const float* w_hwio, int kh, int kw, int cin,
                          int sh, int sw, int padding, int h, int w, float* dx, void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!dy || !w_hwio || !dx || !scratch) return fail(SAGEN_ERR_NULL, "sagen_conv2d_bwd_data: null argument");
        if (ilog2_exact(cout) < 2 || cin % 4) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d_bwd_data: cout must be a power of two >= 4, cin a multiple of 4");
        if (scratch_bytes < sagen_conv2d_bwd_data_scratch_bytes(kh, kw, cin, cout, sh, sw)) return fail(SAGEN_ERR_WORKSPACE, "sagen_conv2d_bwd_data: scratch too small");
        int pt = 0, pl = 0;
        if (padding == 1) {
            pt = std::max((cdiv(h, sh) - 1) * sh + kh - h, 0) / 2;
            pl = std::max((cdiv(w, sw) - 1) * sw + kw - w, 0) / 2;
            if (hout != cdiv(h, sh) || wout != cdiv(w, sw)) return fail(SAGEN_ERR_SHAPE, "sagen_conv2d_bwd_data: output size does not match SAME padding");
        } else if (hout != (h - kh) / sh + 1 || wout != (w - kw) / sw + 1) {
            return fail(SAGEN_ERR_SHAPE, "sagen_conv2d_bwd_data: output size does not match VALID padding");
        }
        float* wp = (float*)scratch;
        IgemmDesc d;
        d.x = dy; d.w = wp; d.y = dx; d.w_split = 1;
        d.Hin = hout; d.Win = wout; d.Cin = cout; d.ldx = cout; d.x_bstride = (long)hout * wout * cout; d.log2Cin = ilog2_exact(cout);
        d.Cout = cin; d.ldy = cin; d.y_rstride = (long)w * cin; d.y_bstride = (long)h * w * cin; d.Hlim = h; d.Wlim = w;
        int rc;
        if (sh == 1 && sw == 1) {
            // dx = correlation of dy with the tap-reversed, channel-transposed filter, padded (kh-1-pt, kw-1-pl) before
            d.M = batch * h * w; d.N = cin; d.K = kh * kw * cout; d.Kpad = (d.K + 15) / 16 * 16;
            d.Hg = h; d.Wg = w; d.ntaps = kh * kw; d.TW = kw; d.tap_h0 = -(kh - 1 - pt); d.tap_w0 = -(kw - 1 - pl);
            rc = pack_conv_flipT_launch(w_hwio, kh * kw, cin, cout, wp, d.Kpad, s);
        } else {
            if (pt || pl) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_conv2d_bwd_data: strided conv with padding before (%d,%d)", pt, pl);
            const int nth = cdiv(kh, sh), ntw = cdiv(kw, sw);
            d.Hg = cdiv(h, sh); d.Wg = cdiv(w, sw);
            d.M = batch * d.Hg * d.Wg; d.N = sh * sw * cin; d.K = nth * ntw * cout; d.Kpad = (d.K + 15) / 16 * 16;
            d.ntaps = nth * ntw; d.TW = ntw; d.tap_sh = -1; d.tap_sw = -1; d.dsh = sh; d.dsw = sw;
            rc = pack_deconv_launch(w_hwio, kh, kw, cin, cout, sh, sw, wp, d.N, d.Kpad, s);
        }
        if (!rc) rc = pack_split_launch(wp, d.N, d.Kpad, s);
        if (rc) return rc;
        return igemm_launch(d, TILE_AUTO, s);
    });
}

size_t sagen_bn_bwd_scratch_bytes(int c) { return align_up((size_t)2 * c * sizeof(double), 256) + reduce_scratch_floats(c) * sizeof(float); }

/* bn_stats = the fp64 (sum, sumsq) accumulators sagen_conv2d filled for the raw conv output y */
int sagen_bn_bwd(const float* ga, const float* gb, const float* act, const float* y, const float* bn_stats, const float* gamma,
                 const float* beta, float eps, int64_t n_pixels, int c, float* dy, float* dz, float* dgamma, float* dbeta,
                 void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!ga || !y || !bn_stats || !gamma || !beta || !dy || !scratch) return fail(SAGEN_ERR_NULL, "sagen_bn_bwd: null argument");
        if (scratch_bytes < sagen_bn_bwd_scratch_bytes(c) || ((uintptr_t)scratch) % 16) return fail(SAGEN_ERR_WORKSPACE, "sagen_bn_bwd: scratch too small / unaligned");
        BnRef bn;
        bn.acc = (const double*)bn_stats; bn.gamma = gamma; bn.beta = beta; bn.inv_count = 1.0 / (double)n_pixels; bn.eps = eps;
        float* part = (float*)((char*)scratch + align_up((size_t)2 * c * sizeof(double), 256));
        int rc = bn_bwd_reduce_launch(ga, gb, act, y, bn, n_pixels, c, (double*)scratch, part, s);
        if (rc) return rc;
        return bn_bwd_apply_launch(ga, gb, act, y, bn, (const double*)scratch, n_pixels, c, dy, dz, dgamma, dbeta, s);
    });
}

int sagen_maxpool3x3s2_bwd(const float* y0, const float* bn_stats, const float* gamma, const float* beta, float eps, const float* pooled,
                           const float* ga, const float* gb, float* dz, int batch, int h, int w, int c, void* stream) {
    if (!y0 || !bn_stats || !gamma || !beta || !pooled || !ga || !dz) return fail(SAGEN_ERR_NULL, "sagen_maxpool3x3s2_bwd: null argument");
    BnRef bn;
    bn.acc = (const double*)bn_stats; bn.gamma = gamma; bn.beta = beta; bn.inv_count = 1.0 / ((double)batch * h * w); bn.eps = eps;
    return maxpool_bwd_launch(y0, bn, pooled, ga, gb, dz, batch, h, w, c, (hipStream_t)stream);
}

size_t sagen_mask_istft_mix_bwd_scratch_bytes(int batch, int ntracks) { return mask_istft_bwd_scratch_floats(batch, ntracks) * sizeof(float); }

int sagen_mask_istft_mix_bwd(const float* dmask, const float* spec, const float* coeffs, const float* dpred, int batch, int ntracks,
                             float* d_dmask, float* d_coeffs, void* scratch, size_t scratch_bytes, void* stream) {
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream;
        if (!dmask || !spec || !coeffs || !dpred || !d_dmask || !d_coeffs || !scratch) return fail(SAGEN_ERR_NULL, "sagen_mask_istft_mix_bwd: null argument");
        if (batch <= 0 || scratch_bytes < sagen_mask_istft_mix_bwd_scratch_bytes(batch, ntracks)) return fail(SAGEN_ERR_WORKSPACE, "sagen_mask_istft_mix_bwd: scratch too small");
        // frames 0 and 24..27 never reach the cropped window: zero gradient
        SAGEN_HIP_CHECK(hipMemsetAsync(d_dmask, 0, (size_t)batch * 28 * 1024 * ntracks * sizeof(float), s));
        return mask_istft_mix_bwd_launch(dmask, 28L * 1024 * ntracks, 0, spec, coeffs, dpred, batch, ntracks, d_dmask, 28L * 1024 * ntracks, 0,
                                         d_coeffs, 3 * (ntracks + 1), (float*)scratch, s);
    });
}

}  // extern "C"
