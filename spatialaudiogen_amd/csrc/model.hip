// sagen_ctx: the native runtime of the inference path — variable inventory, workspace carving,
// filter repacking and the launch sequence that replaces one sess.run of
// SptAudioGen.inference_ops (reference model.py:356-434; called at deploy.py:141, eval.py:145).
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace sagen {

struct VarSpec {
    std::string name;
    int ndim;
    int64_t shape[4];
    long numel() const {
        long n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

struct Buf {            // region of the workspace, in floats
    size_t off = 0, n = 0;
};

struct Named {          // intermediate exposed to parity tests
    Buf buf;
    size_t extra_off = 0;          // float offset inside buf (channel offset of a concat buffer)
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    int64_t pixel_stride = 0;
};

static const int AENC_F[5] = {32, 64, 128, 256, 512};
static const int AENC_K[5][2] = {{7, 16}, {3, 7}, {3, 5}, {3, 5}, {3, 5}};
static const int AENC_S[5][2] = {{4, 8}, {2, 4}, {2, 2}, {1, 1}, {1, 1}};

}  // namespace sagen

using namespace sagen;

struct ProfRec {
    std::string kernel, layer;
    double flops = 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

struct Choice {          // how one contraction is launched
    int tile = -1;       // IgemmTile, -1 = heuristic
    int splitk = 0;      // 0 = heuristic
    float us = 0.f;      // measured time of the choice (autotune)
};

struct sagen_ctx {
    sagen_config cfg;
    // per-layer launch plan (filled by sagen_autotune; empty = heuristics)
    std::map<std::string, Choice> plan;
    std::map<std::string, bool> materialize;     // conv_2 layers: apply the producer's BN+ReLU in a separate pass?
    bool tuning = false;
    bool fp32_only = false;                      // SAGEN_FP32_ONLY=1: never use the bf16x3 tiles
    bool use_p3 = true;                          // 3x3 stride-1 trunk convs read pre-split bf16 planes (conv3p.hip); SAGEN_NO_P3=1 disables
    int p3_from_stage = 3;                       // ... from this ResNet stage on (2..5; SAGEN_P3_FROM_STAGE): see resnet()
    hipEvent_t tune_e0 = nullptr, tune_e1 = nullptr;
    // second, context-owned stream: the audio chain (and the flow trunk) run under the video trunk
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_stft = nullptr;
    // optional per-launch HIP-event profiler (sagen_profile_enable)
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    int B = 0;
    int snd_size = 52799, snd_contx = 48000, snd_dur = 4800;
    int enc_h[6], enc_w[6], enc_c[6];   // audio encoder pyramid (index 0 = magnitude)
    int Cb = 0;                         // bottleneck width
    int nsep = 32;
    bool has_video = false, has_flow = false, freq_mask = true;

    std::vector<VarSpec> vars;
    std::map<std::string, int> var_index;
    std::vector<const float*> var_ptr;
    bool bound = false;

    // workspace
    size_t ws_floats = 0;
    float* ws = nullptr;
    std::map<std::string, Buf> bufs;
    std::map<std::string, Named> named;

    Buf alloc(const std::string& name, size_t n) {
        Buf b;
        b.off = ws_floats;
        b.n = n;
        ws_floats += (n + 63) / 64 * 64;     // 256-byte granules
        bufs[name] = b;
        return b;
    }
    float* p(const std::string& name) const { return ws + bufs.at(name).off; }
    const float* v(const std::string& name) const { return var_ptr[var_index.at(name)]; }
    void add_var(const std::string& name, std::initializer_list<int64_t> shape) {
        VarSpec s;
        s.name = name;
        s.ndim = (int)shape.size();
        int i = 0;
        for (auto d : shape) s.shape[i++] = d;
        for (; i < 4; ++i) s.shape[i] = 1;
        var_index[name] = (int)vars.size();
        vars.push_back(s);
    }
    void expose(const std::string& name, const std::string& buf, size_t extra, std::initializer_list<int64_t> shape,
                int64_t pixel_stride) {
        Named nm;
        nm.buf = bufs.at(buf);
        nm.extra_off = extra;
        nm.ndim = (int)shape.size();
        int i = 0;
        for (auto d : shape) nm.shape[i++] = d;
        nm.pixel_stride = pixel_stride;
        named[name] = nm;
    }
};

namespace sagen {

static void add_resnet_vars(sagen_ctx* c, const std::string& scope) {
    auto bn = [&](const std::string& p, int ch) {
        for (const char* leaf : {"beta", "gamma", "moving_mean", "moving_variance"}) c->add_var(p + "/bn/" + leaf, {ch});
    };
    c->add_var(scope + "/conv1/conv/weights", {7, 7, 3, 64});
    bn(scope + "/conv1/conv", 64);
    int cin = 64;
    const int couts[4] = {64, 128, 256, 512};
    for (int st = 0; st < 4; ++st) {
        const int cout = couts[st];
        for (int unit = 1; unit <= 2; ++unit) {
            const std::string pfx = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(unit);
            if (unit == 1 && cin != cout) c->add_var(pfx + "/shortcut/weights", {1, 1, cin, cout});
            c->add_var(pfx + "/conv_1/weights", {3, 3, cin, cout});
            bn(pfx + "/conv_1", cout);
            c->add_var(pfx + "/conv_2/weights", {3, 3, cout, cout});
            bn(pfx + "/conv_2", cout);
            cin = cout;
        }
    }
}

static size_t packed_floats(long N, long K) { return (size_t)N * ((K + 15) / 16 * 16); }

// choose a split-K factor for low-parallelism contractions (>= ~2 workgroups per CU, >= 8 K tiles per split)
static int auto_splitk(const IgemmDesc& d, IgemmTile tile) {
    const int bm = (tile == TILE_32x128) ? 32 : 64, bn = (tile == TILE_32x128) ? 128 : 64;
    const long blocks = (long)cdiv(d.M, bm) * cdiv(d.N, bn);
    const int nk = d.Kpad / 16;
    if (blocks >= 384 || nk < 16) return 1;
    int sk = (int)std::min<long>({(512 + blocks - 1) / blocks, (long)nk / 8, 64L});
    return std::max(sk, 1);
}

}  // namespace sagen

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
int sagen_create_impl(sagen_ctx** out, const sagen_config* cfg) {
    if (!out || !cfg) return fail(SAGEN_ERR_NULL, "sagen_create: null argument");
    if (cfg->batch <= 0) return fail(SAGEN_ERR_SHAPE, "sagen_create: batch=%d", cfg->batch);
    if (!(cfg->encoders & SAGEN_ENC_AUDIO))
        return fail(SAGEN_ERR_UNSUPPORTED, "the audio encoder is mandatory (reference model.py:207)");
    if (cfg->audio_rate != 48000 || cfg->video_rate != 10 || cfg->context != 1.0f || cfg->sample_duration != 0.1f ||
        cfg->ambi_order != 1 || cfg->fft_window != 0.025f)
        return fail(SAGEN_ERR_UNSUPPORTED,
                    "HIP path implements audio_rate=48000 video_rate=10 context=1.0 sample_duration=0.1 ambi_order=1 "
                    "fft_window=0.025 (the geometry of every BASELINE config)");
    if (cfg->separation != SAGEN_SEP_NONE && cfg->separation != SAGEN_SEP_FREQ_MASK)
        return fail(SAGEN_ERR_UNSUPPORTED, "unknown separation mode %d", cfg->separation);
    if (cfg->separation == SAGEN_SEP_NONE && cfg->num_sep_tracks > 1)
        return fail(SAGEN_ERR_UNSUPPORTED, "separation 'none' decodes the mono track only: num_sep_tracks must be 1 (got %d); the reference's fc3 "
                    "would be [.., 3*(num_sep_tracks+1)] (model.py:254) against a single separated track (model.py:274-280)", cfg->num_sep_tracks);
    if (cfg->separation == SAGEN_SEP_FREQ_MASK && cfg->num_sep_tracks != 16 && cfg->num_sep_tracks != 32 &&
        cfg->num_sep_tracks != 64)
        return fail(SAGEN_ERR_UNSUPPORTED, "num_sep_tracks=%d (supported 16/32/64)", cfg->num_sep_tracks);
    if (cfg->n_loc_units < 0 || cfg->n_loc_units > 4) return fail(SAGEN_ERR_SHAPE, "n_loc_units=%d", cfg->n_loc_units);
    for (int i = 0; i < cfg->n_loc_units; ++i)
        if (cfg->loc_units[i] <= 0 || cfg->loc_units[i] % 4) return fail(SAGEN_ERR_UNSUPPORTED, "loc_units[%d]=%d must be a positive multiple of 4", i, cfg->loc_units[i]);

    sagen_ctx* c = new sagen_ctx();

    c->fp32_only = getenv("SAGEN_FP32_ONLY") != nullptr;
    c->use_p3 = !c->fp32_only && getenv("SAGEN_NO_P3") == nullptr;
    if (const char* e = getenv("SAGEN_P3_FROM_STAGE")) c->p3_from_stage = atoi(e);
    c->cfg = *cfg;
    c->B = cfg->batch;
    c->has_video = cfg->encoders & SAGEN_ENC_VIDEO;
    c->has_flow = cfg->encoders & SAGEN_ENC_FLOW;
    c->freq_mask = cfg->separation == SAGEN_SEP_FREQ_MASK;
    c->nsep = c->freq_mask ? cfg->num_sep_tracks : 1;
    const int B = c->B;

    // audio encoder pyramid (model.py:161-187), H = frames 46:173, W = 1024 bins
    c->enc_h[0] = 127; c->enc_w[0] = 1024; c->enc_c[0] = 1;
    for (int l = 0; l < 5; ++l) {
        c->enc_h[l + 1] = (c->enc_h[l] - AENC_K[l][0]) / AENC_S[l][0] + 1;
        c->enc_w[l + 1] = (c->enc_w[l] - AENC_K[l][1]) / AENC_S[l][1] + 1;
        c->enc_c[l + 1] = AENC_F[l];
    }
    c->Cb = 1024 + (c->has_video ? 512 : 0) + (c->has_flow ? 512 : 0);

    // ---- variable inventory ----
    {
        int cin = 1;
        for (int l = 0; l < 5; ++l) {
            const std::string n = "audio_encoder/conv" + std::to_string(l + 1);
            c->add_var(n + "/weights", {AENC_K[l][0], AENC_K[l][1], cin, AENC_F[l]});
            c->add_var(n + "/biases", {AENC_F[l]});
            cin = AENC_F[l];
        }
        if (c->has_video) add_resnet_vars(c, "video_encoder");
        if (c->has_flow) add_resnet_vars(c, "flow_encoder");
        c->add_var("bottleneck/audio-fc/weights", {c->enc_w[5] * c->enc_c[5], 1024});
        c->add_var("bottleneck/audio-fc/biases", {1024});
        for (const char* e : {"video", "flow"}) {
            if ((std::string(e) == "video" && !c->has_video) || (std::string(e) == "flow" && !c->has_flow)) continue;
            c->add_var(std::string("bottleneck/") + e + "-fc-red/weights", {512, 128});
            c->add_var(std::string("bottleneck/") + e + "-fc-red/biases", {128});
            c->add_var(std::string("bottleneck/") + e + "-fc/weights", {7 * 14 * 128, 512});
            c->add_var(std::string("bottleneck/") + e + "-fc/biases", {512});
        }
        int fin = c->Cb;
        for (int i = 0; i < cfg->n_loc_units; ++i) {
            const std::string n = "localization/fc" + std::to_string(i + 1);
            c->add_var(n + "/weights", {fin, cfg->loc_units[i]});
            c->add_var(n + "/biases", {cfg->loc_units[i]});
            fin = cfg->loc_units[i];
        }
        const int nlast = 3 * 1 * (c->nsep + 1);
        const std::string n = "localization/fc" + std::to_string(cfg->n_loc_units + 1);
        c->add_var(n + "/weights", {fin, nlast});
        c->add_var(n + "/biases", {nlast});
        if (c->freq_mask) {
            c->add_var("separation/fc-feats/weights", {c->Cb, 512});
            c->add_var("separation/fc-feats/biases", {512});
            const int nfs[5] = {c->nsep, 32, 64, 128, 256};
            int dcin = 1024;
            for (int l = 4; l >= 0; --l) {
                const std::string dn = "separation/deconv" + std::to_string(l + 1);
                c->add_var(dn + "/weights", {AENC_K[l][0], AENC_K[l][1], nfs[l], dcin});
                c->add_var(dn + "/biases", {nfs[l]});
                dcin = nfs[l] + (l > 0 ? AENC_F[l - 1] : 0);
            }
        }
    }
    c->var_ptr.assign(c->vars.size(), nullptr);

    // ---- workspace carving ----
    // packed filters
    for (const auto& vs : c->vars) {
        if (vs.name.size() < 8 || vs.name.compare(vs.name.size() - 8, 8, "/weights") != 0) continue;
        size_t n;
        if (vs.name.find("/deconv") != std::string::npos) {
            const int l = vs.name[vs.name.find("/deconv") + 7] - '1';
            const int sh = AENC_S[l][0], sw = AENC_S[l][1];
            const long taps = (long)cdiv(vs.shape[0], sh) * cdiv(vs.shape[1], sw);
            n = packed_floats((long)sh * sw * vs.shape[2], taps * vs.shape[3]);
        } else if (vs.ndim == 4) {
            long cinp = vs.shape[2] == 3 ? 4 : vs.shape[2];
            long taps = vs.shape[0] * vs.shape[1];
            if (vs.shape[2] == 3) taps = vs.shape[0] * (vs.shape[1] + 1);        // ResNet stem: tap rows padded 7 -> 8 (igemm3s2.hip)
            if (vs.shape[2] == 1) { cinp = vs.shape[1]; taps = vs.shape[0]; }   // audio conv1: kw acts as channels
            n = packed_floats(vs.shape[3], taps * cinp);
        } else {
            n = packed_floats(vs.shape[1], vs.shape[0]);
        }
        c->alloc("pk:" + vs.name, packed_split_floats(n));      // fp32 filter + its three bf16 planes (bf16x3 tiles)
    }
    // activations
    c->alloc("mag", (size_t)B * 127 * 1024);
    c->alloc("spec", (size_t)B * 28 * 513 * 2);
    // concat buffers cat_l: [B, H_l, W_l, C_dec + C_enc]; cat5 = [conv5 | fc-feats]
    for (int l = 1; l <= 5; ++l)
        c->alloc("cat" + std::to_string(l), (size_t)B * c->enc_h[l] * c->enc_w[l] * 2 * c->enc_c[l]);
    c->alloc("bott", (size_t)B * 3 * c->Cb);
    for (int i = 0; i < cfg->n_loc_units; ++i) c->alloc("loc" + std::to_string(i + 1), (size_t)B * 3 * cfg->loc_units[i]);
    c->alloc("coeffs", (size_t)B * 3 * 3 * (c->nsep + 1));
    c->alloc("splitk", std::max<size_t>((size_t)16 << 20, (size_t)B * 56 * 112 * 64 * 2));   // fp32 split-K partials (up to 2 splits of the largest conv), checked per use
    if (c->freq_mask) {
        c->alloc("dmask", (size_t)B * 23 * 1024 * c->nsep);
        c->alloc("frames", mask_istft_scratch_bytes(B) / sizeof(float));
    }
    c->alloc("splitk_aux", (size_t)8 << 20);          // split-K scratch of the second stream (audio chain / flow FCs)
    for (int set = 0; set < 2; ++set) {
        const std::string x = set ? "_b" : "";
        if (set == 0 ? !(c->has_video || c->has_flow) : !(c->has_video && c->has_flow)) continue;   // "_b": flow trunk next to the video trunk
        c->alloc("xpad" + x, (size_t)B * 229 * 454 * 4);
        c->alloc("y0" + x, (size_t)B * 112 * 224 * 64);
        const size_t stage = (size_t)B * 56 * 112 * 64;
        for (const char* nm : {"rx0", "rx1", "ry1", "ry2", "rsc", "ry1n"}) c->alloc(nm + x, stage);
        c->alloc("p3" + x, (p3_bytes(B, 56, 112, 64) + 3) / 4 + 64);   // bf16 planes of the current 3x3 conv input (largest: stage 2)
        c->alloc("bnacc" + x, (size_t)24 * 2 * 512 * 2);   // fp64 (sum, sumsq) accumulators per BN layer
        c->alloc("fcred" + x, (size_t)B * 98 * 128);
        // trunk output (block conv5_2) for parity tests: the ping-pong lands in rx0 after the 8 blocks
        const std::string enc = (set == 1 || !c->has_video) ? "flow_encoder" : "video_encoder";
        c->expose(enc + "/conv5_2", "rx0" + x, 0, {B, 7, 14, 512}, 512);
    }
    // scratch the tuner overwrites between candidate timings to cool the L2 (any buffer that is dead while a contraction runs
    // and is rewritten before its next use: the mask buffer, else the second stream's split-K scratch)
    c->bufs["dmask_or_scratch_flush"] = c->freq_mask ? c->bufs.at("dmask") : c->bufs.at("splitk_aux");
    // intermediates for parity tests
    c->expose("mag", "mag", 0, {B, 127, 1024, 1}, 1);
    c->expose("stft", "spec", 0, {B, 28, 513, 2}, 2);
    for (int l = 1; l <= 5; ++l) {
        const int ce = c->enc_c[l];
        c->expose("audio_encoder/conv" + std::to_string(l), "cat" + std::to_string(l), l == 5 ? 0 : ce,
                  {B, c->enc_h[l], c->enc_w[l], ce}, 2 * ce);
    }
    c->expose("bottleneck", "bott", 0, {B, 3, c->Cb}, c->Cb);
    c->expose("localization/coeffs", "coeffs", 0, {B, 3, 3, c->nsep + 1}, c->nsep + 1);
    if (c->freq_mask) c->expose("separation/deconv1", "dmask", 0, {B, 23, 1024, c->nsep}, c->nsep);
    // second stream + fork/join events (host-side objects; no device memory)
    if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stft, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->aux = nullptr;      // no device / no stream: forward falls back to a single stream (and fails at launch)
    }
    *out = c;
    return SAGEN_OK;
}

// ------------------------------------------------------------------------------------------------
// bind: check + borrow the variables, repack filters into the workspace
// ------------------------------------------------------------------------------------------------
int sagen_bind_impl(sagen_ctx* c, const sagen_tensor* tensors, int n, void* workspace, size_t workspace_bytes,
                    hipStream_t s) {
    if (!c || !tensors || !workspace) return fail(SAGEN_ERR_NULL, "sagen_bind_weights: null argument");
    if (workspace_bytes < c->ws_floats * sizeof(float))
        return fail(SAGEN_ERR_WORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, c->ws_floats * sizeof(float));
    if (((uintptr_t)workspace) % 256) return fail(SAGEN_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    c->ws = (float*)workspace;
    std::vector<const float*> ptr(c->vars.size(), nullptr);
    for (int i = 0; i < n; ++i) {
        const sagen_tensor& t = tensors[i];
        if (!t.name || !t.data) return fail(SAGEN_ERR_NULL, "tensor %d has a null name or data pointer", i);
        auto it = c->var_index.find(t.name);
        if (it == c->var_index.end()) continue;     // checkpoint extras (Adam slots, step, metrics/*) are ignored
        const VarSpec& vs = c->vars[it->second];
        bool ok = t.ndim == vs.ndim;
        for (int k = 0; ok && k < vs.ndim; ++k) ok = t.shape[k] == vs.shape[k];
        if (!ok) return fail(SAGEN_ERR_WEIGHTS, "variable %s has the wrong shape", t.name);
        if (((uintptr_t)t.data) % 16) return fail(SAGEN_ERR_WEIGHTS, "variable %s is not 16-byte aligned", t.name);
        ptr[it->second] = t.data;
    }
    for (size_t i = 0; i < c->vars.size(); ++i) {
        const std::string& nm = c->vars[i].name;
        const bool moving = nm.find("/moving_") != std::string::npos;   // never read: BN runs in train mode (model.py:197)
        if (!ptr[i] && !moving) return fail(SAGEN_ERR_WEIGHTS, "variable %s was not provided", nm.c_str());
    }
    c->var_ptr = ptr;
    int rc = fft_tables_ensure(s);
    if (rc) return rc;
    // repack filters
    for (const auto& vs : c->vars) {
        if (vs.name.size() < 8 || vs.name.compare(vs.name.size() - 8, 8, "/weights") != 0) continue;
        const float* src = c->v(vs.name);
        float* dst = c->p("pk:" + vs.name);
        if (vs.name.find("/deconv") != std::string::npos) {
            const int l = vs.name[vs.name.find("/deconv") + 7] - '1';
            const int sh = AENC_S[l][0], sw = AENC_S[l][1];
            const int taps = cdiv(vs.shape[0], sh) * cdiv(vs.shape[1], sw);
            const int N = sh * sw * (int)vs.shape[2], K = taps * (int)vs.shape[3];
            rc = pack_deconv_launch(src, (int)vs.shape[0], (int)vs.shape[1], (int)vs.shape[2], (int)vs.shape[3], sh, sw, dst,
                                    N, (K + 15) / 16 * 16, s);
            if (!rc) rc = pack_split_launch(dst, N, (K + 15) / 16 * 16, s);
        } else if (vs.ndim == 4) {
            int cin = (int)vs.shape[2], cinp = cin == 3 ? 4 : cin, taps = (int)(vs.shape[0] * vs.shape[1]);
            if (cin == 1) { cin = cinp = (int)vs.shape[1]; taps = (int)vs.shape[0]; }
            const bool stem = cin == 3;                   // tap rows padded 7 -> 8, K = (dh, dw8, c4)
            if (stem) taps = (int)(vs.shape[0] * (vs.shape[1] + 1));
            const int K = taps * cinp;
            rc = pack_conv_launch(src, taps, cin, cinp, (int)vs.shape[3], dst, (int)vs.shape[3], (K + 15) / 16 * 16, s,
                                  stem ? (int)vs.shape[1] : 0, stem ? (int)vs.shape[1] + 1 : 0);
            if (!rc) rc = pack_split_launch(dst, (int)vs.shape[3], (K + 15) / 16 * 16, s);
        } else {
            const int K = (int)vs.shape[0], N = (int)vs.shape[1];
            rc = pack_conv_launch(src, 1, K, K, N, dst, N, (K + 15) / 16 * 16, s);
            if (!rc) rc = pack_split_launch(dst, N, (K + 15) / 16 * 16, s);
        }
        if (rc) return rc;
    }
    c->bound = true;
    return SAGEN_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
namespace sagen {

struct Fwd {
    sagen_ctx* c;
    hipStream_t s;
    int rc = SAGEN_OK;
    std::string layer;      // label of the layer being launched (profiling only)
    std::string wsname = "splitk";   // split-K scratch of this launch stream
    std::string sfx;                 // suffix of the trunk buffers this stream owns ("" or "_b")
    hipEvent_t wait_before_mfma = nullptr;   // event the first contraction of this stream has to wait for (see forward)

    hipEvent_t next_event() {
        if (c->events_used == c->event_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            c->event_pool.push_back(e);
        }
        return c->event_pool[c->events_used++];
    }
    // time one launch (or launch group) with a pair of events on the launch stream
    template <class F>
    void timed(const char* kernel, double flops, F&& launch) {
        if (rc) return;
        if (!c->profiling) { rc = launch(); return; }
        ProfRec r;
        r.kernel = kernel; r.layer = layer; r.flops = flops;
        r.e0 = next_event(); r.e1 = next_event();
        if (!r.e0 || !r.e1) { rc = fail(SAGEN_ERR_HIP, "hipEventCreate failed"); return; }
        hipError_t he = hipEventRecord(r.e0, s);
        rc = launch();
        if (he == hipSuccess) he = hipEventRecord(r.e1, s);
        if (he != hipSuccess && !rc) rc = fail(SAGEN_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(he));
        c->prof.push_back(r);
    }

    // ---- one contraction: direct, or split-K partials + reduce (bias / ReLU / row replication / BN statistics) ----
    bool dense_out(const IgemmDesc& d) const {
        return d.dsh * d.dsw == 1 && d.g_h0 == 0 && d.g_w0 == 0 && d.y_rstride == (long)d.Wg * d.ldy &&
               (d.M <= d.Hg * d.Wg || d.y_bstride == (long)d.Hg * d.Wg * d.ldy);
    }
    size_t ws_capacity() const { return c->bufs.at(wsname).n; }

    // launches the contraction with an explicit choice; returns the number of BN partial rows written (0 if none)
    int run_choice(const IgemmDesc& d, int rep, IgemmTile tile, int sk) {
        if (rc) return 0;
        if (sk > 1 || rep > 1) {
            IgemmDesc e = d;
            e.splitk = sk;
            e.splitk_ws = c->ws + c->bufs.at(wsname).off;
            e.bias = nullptr; e.relu_out = 0; e.stats = nullptr;
            timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(e, tile, s); });
            timed("splitk_reduce_kernel", 0.0, [&] {
                return splitk_reduce_launch(e.splitk_ws, sk, d.M, d.N, d.bias, d.relu_out, d.y, d.ldy, rep, d.stats, s); });
            return 0;
        }
        timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(d, tile, s); });
        return 0;
    }

    Choice heuristic(const IgemmDesc& d, int rep, bool allow_split) const {
        Choice ch;
        const IgemmTile tile = igemm_pick_tile(d);
        ch.tile = (int)tile;
        const bool can_split = allow_split && dense_out(d) && !d.stats;
        ch.splitk = can_split ? auto_splitk(d, tile) : 1;
        while (ch.splitk > 1 && (size_t)ch.splitk * d.M * d.N > ws_capacity()) --ch.splitk;
        return ch;
    }

    // one timed launch (group) of a candidate, in microseconds
    float time_once(const IgemmDesc& d, int rep, IgemmTile tile, int sk) {
        // cold L2: in the forward a layer's filters and activations are not L2-resident from a previous run of the SAME layer;
        // back-to-back timing made the tuner prefer tiles that only win on warm caches (conv5 planes: 14 MB)
        const Buf& fl = c->bufs.at("dmask_or_scratch_flush");
        (void)hipMemsetAsync(c->ws + fl.off, 0, std::min<size_t>(fl.n * sizeof(float), (size_t)48 << 20), s);
        (void)hipEventRecord(c->tune_e0, s);
        run_choice(d, rep, tile, sk);
        (void)hipEventRecord(c->tune_e1, s);
        if (hipEventSynchronize(c->tune_e1) != hipSuccess) { rc = fail(SAGEN_ERR_HIP, "autotune: event sync failed"); return 1e30f; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->tune_e0, c->tune_e1);
        return ms * 1e3f;
    }

    // time every (tile, split-K) candidate on the real operands (sagen_autotune): a first pass (1 warm + 4 timed, min)
    // over all candidates, then a playoff of the three fastest (8 interleaved runs each, median) - single timings of
    // ~10 us launches are too noisy to separate close candidates
    Choice tune(const IgemmDesc& d, int rep, bool allow_split) {
        Choice top[3];
        for (auto& t : top) { t = heuristic(d, rep, allow_split); t.us = 1e30f; }
        const bool dense = dense_out(d);
        static const int SKS[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64};
        for (int t = 0; t < (int)TILE_AUTO && !rc; ++t) {
            const IgemmTile tile = (IgemmTile)t;
            const int bm = igemm_tile_bm(tile), bn = igemm_tile_bn(tile);
            if (!igemm_tile_ok(d, tile)) continue;
            const int nk = d.Kpad / igemm_tile_bk(tile);
            if (bn > 32 && bn >= 2 * d.N) continue;                     // mostly-empty N tile
            if (bm > 32 && bm >= 4 * d.M) continue;
            for (int sk : SKS) {
                if (sk > 1 && (!allow_split || !dense || igemm_tile_p3(tile))) break;
                if (sk > 1 && (nk / sk < 4 || (size_t)sk * d.M * d.N > ws_capacity())) break;
                if (sk == 1 && rep > 1 && (size_t)d.M * d.N > ws_capacity()) continue;
                const long blocks = (long)cdiv(d.M, bm) * cdiv(d.N, bn) * sk;
                if (sk > 1 && blocks > 8192) break;                     // more parallelism than the chip can use
                float t_best = 1e30f;
                for (int it = 0; it < 5 && !rc; ++it) {
                    const float us = time_once(d, rep, tile, sk);
                    if (it > 0) t_best = std::min(t_best, us);          // first run warms caches / code
                }
                Choice ch; ch.tile = t; ch.splitk = sk; ch.us = t_best;
                for (int k = 0; k < 3; ++k)
                    if (ch.us < top[k].us) { std::swap(ch, top[k]); }
            }
        }
        if (rc || top[1].us > 1e29f) return top[0];
        // playoff
        std::vector<float> runs[3];
        const int nc = top[2].us > 1e29f ? 2 : 3;
        for (int it = 0; it < 8 && !rc; ++it)
            for (int k = 0; k < nc; ++k) runs[k].push_back(time_once(d, rep, (IgemmTile)top[k].tile, top[k].splitk));
        int best = 0;
        for (int k = 0; k < nc; ++k) {
            std::sort(runs[k].begin(), runs[k].end());
            top[k].us = runs[k][runs[k].size() / 2];
            if (top[k].us < top[best].us) best = k;
        }
        return top[best];
    }

    int contract(const IgemmDesc& d_in, int rep = 1, bool allow_split = true) {
        if (rc) return 0;
        IgemmDesc d = d_in;
        d.w_split = c->fp32_only ? 0 : 1;          // every bound filter carries its bf16x3 planes
        if (rep > 1 && !dense_out(d)) { rc = fail(SAGEN_ERR_UNSUPPORTED, "replicated store needs a dense plain epilogue"); return 0; }
        Choice ch;
        auto it = c->plan.find(layer);
        if (c->tuning) {
            const bool was_prof = c->profiling;
            c->profiling = false;
            ch = tune(d, rep, allow_split);
            c->profiling = was_prof;
            c->plan[layer] = ch;
            if (d.stats && !rc && hipMemsetAsync(d.stats, 0, (size_t)2 * d.N * sizeof(double), s) != hipSuccess)
                rc = fail(SAGEN_ERR_HIP, "autotune: memset failed");         // candidates polluted the accumulators
        } else if (it != c->plan.end()) {
            ch = it->second;
            if (!igemm_tile_ok(d, (IgemmTile)ch.tile)) ch = heuristic(d, rep, allow_split);
            if (ch.splitk > 1 && (!allow_split || !dense_out(d) || igemm_tile_p3((IgemmTile)ch.tile) || d.Kpad / igemm_tile_bk((IgemmTile)ch.tile) / ch.splitk < 1)) ch.splitk = 1;
        } else {
            ch = heuristic(d, rep, allow_split);
        }
        if ((ch.splitk > 1 || rep > 1) && (size_t)std::max(ch.splitk, 1) * d.M * d.N > ws_capacity()) {
            rc = fail(SAGEN_ERR_WORKSPACE, "split-K scratch too small for %s", layer.c_str());
            return 0;
        }
        return run_choice(d, rep, (IgemmTile)ch.tile, std::max(ch.splitk, 1));
    }
    void gemm(const IgemmDesc& d, int rep = 1, bool allow_split = true) { contract(d, rep, allow_split); }

    // tfw.conv_2d geometry (core.py:156-220): dense NHWC input/output with pixel strides
    IgemmDesc conv_desc(const float* x, int Hin, int Win, int Cin, int ldx, const float* wp, int kh, int kw, int sh, int sw,
                        bool same, int Cout, float* y, int ldy, int& Hout, int& Wout) {
        IgemmDesc d;
        int pt = 0, pl = 0;
        if (same) {
            Hout = cdiv(Hin, sh); Wout = cdiv(Win, sw);
            pt = std::max((Hout - 1) * sh + kh - Hin, 0) / 2;
            pl = std::max((Wout - 1) * sw + kw - Win, 0) / 2;
        } else {
            Hout = (Hin - kh) / sh + 1; Wout = (Win - kw) / sw + 1;
        }
        d.x = x; d.w = wp; d.y = y;
        d.M = c->B * Hout * Wout; d.N = Cout; d.K = kh * kw * Cin; d.Kpad = (d.K + 15) / 16 * 16;
        d.Hg = Hout; d.Wg = Wout;
        d.Hin = Hin; d.Win = Win; d.Cin = Cin; d.ldx = ldx; d.x_bstride = (long)Hin * Win * ldx;
        d.in_sh = sh; d.in_sw = sw;
        d.ntaps = kh * kw; d.TW = kw; d.tap_sh = 1; d.tap_sw = 1; d.tap_h0 = -pt; d.tap_w0 = -pl;
        d.log2Cin = ilog2_exact(Cin);
        d.Cout = Cout; d.Hlim = Hout; d.Wlim = Wout;
        d.ldy = ldy; d.y_rstride = (long)Wout * ldy; d.y_bstride = (long)Hout * Wout * ldy;
        return d;
    }

    // tfw.fully_connected (core.py:43-93) on dense rows
    void fc(const float* x, int M, int K, int ldx, const std::string& name, int N, bool relu, float* y, int ldy, int rep = 1) {
        layer = name;
        IgemmDesc d;
        d.x = x; d.w = c->p("pk:" + name + "/weights"); d.y = y; d.bias = c->v(name + "/biases");
        d.M = M; d.N = N; d.K = K; d.Kpad = (K + 15) / 16 * 16;
        d.Hg = 1; d.Wg = 1; d.Hin = 1; d.Win = 1; d.Cin = K; d.ldx = ldx; d.x_bstride = ldx;
        d.ntaps = 1; d.Cout = N; d.Hlim = 1; d.Wlim = 1; d.ldy = ldy; d.y_rstride = ldy; d.y_bstride = ldy;
        d.relu_out = relu;
        gemm(d, rep);
    }

    // tfw.deconv_2d (core.py:96-153) as a stride-1 conv with a depth-to-space epilogue
    void deconv(const float* x, int Hin, int Win, int Cin, int l, float* y, int ldy, bool relu, int a0, int a1, int Ylim,
                long y_bstride, long y_row0) {
        const std::string name = "separation/deconv" + std::to_string(l + 1);
        layer = name;
        const int kh = AENC_K[l][0], kw = AENC_K[l][1], sh = AENC_S[l][0], sw = AENC_S[l][1];
        const int Cout = l == 0 ? c->nsep : AENC_F[l - 1];
        const int Hout = Hin * sh + kh - sh, Wout = Win * sw + kw - sw;
        const int nth = cdiv(kh, sh), ntw = cdiv(kw, sw);
        IgemmDesc d;
        d.x = x; d.w = c->p("pk:" + name + "/weights"); d.bias = c->v(name + "/biases");
        d.Hg = (a1 > a0 ? a1 - a0 : cdiv(Hout, sh)); d.g_h0 = a0; d.Wg = cdiv(Wout, sw);
        d.M = c->B * d.Hg * d.Wg; d.N = sh * sw * Cout; d.K = nth * ntw * Cin; d.Kpad = (d.K + 15) / 16 * 16;
        d.Hin = Hin; d.Win = Win; d.Cin = Cin; d.ldx = Cin; d.x_bstride = (long)Hin * Win * Cin;
        d.ntaps = nth * ntw; d.TW = ntw; d.tap_sh = -1; d.tap_sw = -1; d.log2Cin = ilog2_exact(Cin);
        d.dsh = sh; d.dsw = sw; d.Cout = Cout;
        d.Hlim = Ylim > 0 ? Ylim : Hout; d.Wlim = Wout;
        d.ldy = ldy; d.y_rstride = (long)Wout * ldy;
        d.y_bstride = y_bstride > 0 ? y_bstride : (long)Hout * Wout * ldy;
        d.y = y - y_row0 * d.y_rstride;
        d.relu_out = relu;
        gemm(d);
    }

    double* bn_acc(int layer_index) { return reinterpret_cast<double*>(c->p("bnacc" + sfx)) + (size_t)layer_index * 2 * 512; }
    // batch-norm of layer `bn_name` by reference to its statistics accumulators (consumers finalize in-kernel)
    BnRef bn_ref(int layer_index, const std::string& bn_name, long count) {
        BnRef r;
        r.acc = bn_acc(layer_index);
        r.gamma = c->v(bn_name + "/bn/gamma");
        r.beta = c->v(bn_name + "/bn/beta");
        r.inv_count = 1.0 / (double)count;
        r.eps = 1e-3f;
        return r;
    }

    // conv of the ResNet trunk: raw output + batch statistics into accumulator `bn_index`; `bn_in` = the producer's
    // batch-norm + ReLU applied to the input on the fly
    void conv_bn(const float* x, int Hin, int Win, int Cin, const std::string& name, int k, int stride, int Cout,
                 const BnRef& bn_in, float* y, int& Hout, int& Wout, int bn_index, const std::string& plan_key = "",
                 const void* planes = nullptr) {
        if (rc) return;
        IgemmDesc d = conv_desc(x, Hin, Win, Cin, Cin, c->p("pk:" + name + "/weights"), k, k, stride, stride, true, Cout, y,
                                Cout, Hout, Wout);
        if (planes) {                       // the input as pre-split bf16 planes (p3.hip); x may be null then
            d.xp3 = planes;
            d.p3_np = c->B * Hin * (Win + 1);
            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 96);
            d.xp3_bytes = (unsigned)p3_bytes(c->B, Hin, Win, Cin);
        }
        d.bn_in = bn_in;
        d.stats = bn_acc(bn_index);
        layer = plan_key.empty() ? name : plan_key;
        contract(d);
    }

    // ResNet18 -> conv5_2 in training-mode BN (resnet.py:123-236); returns the [B,7,14,512] output
    const float* resnet(const float* img, const std::string& scope) {
        const int B = c->B;
        int li = 0;
        if (!rc && hipMemsetAsync(c->p("bnacc" + sfx), 0, c->bufs.at("bnacc" + sfx).n * sizeof(float), s) != hipSuccess)
            rc = fail(SAGEN_ERR_HIP, "hipMemsetAsync(bn accumulators) failed");
        layer = scope + "/pad";
        timed("pad_nhwc3to4_kernel", 0.0, [&] { return pad_nhwc3to4_launch(img, c->p("xpad" + sfx), B, 224, 448, 2, 3, 2, 4, s); });
        // conv1 7x7/2 SAME == VALID 7x8 (8th tap column = zero weights) on the padded 4-channel image
        int H = 0, W = 0;
        if (!rc && wait_before_mfma) {
            if (hipStreamWaitEvent(s, wait_before_mfma, 0) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "hipStreamWaitEvent failed");
            wait_before_mfma = nullptr;
        }
        {
            const std::string name = scope + "/conv1/conv";
            IgemmDesc d = conv_desc(c->p("xpad" + sfx), 229, 454, 4, 4, c->p("pk:" + name + "/weights"), 7, 8, 2, 2, false, 64,
                                    c->p("y0" + sfx), 64, H, W);
            d.stats = bn_acc(li);
            layer = name;
            contract(d);
            const BnRef bn = bn_ref(li, name, (long)B * H * W);
            if (c->use_p3 && c->p3_from_stage <= 2)       // pooled block input as fp32 (residual) AND as planes (operand of conv2_1/conv_1)
                timed("p3_maxpool_kernel", 0.0, [&] { return p3_maxpool_launch(c->p("y0" + sfx), nullptr, nullptr, bn, c->p("rx0" + sfx), c->p("p3" + sfx), B, H, W, 64, s); });
            else
                timed("maxpool3x3s2_kernel", 0.0, [&] { return maxpool3x3s2_launch(c->p("y0" + sfx), nullptr, nullptr, bn, c->p("rx0" + sfx), B, H, W, 64, s); });
            ++li;
            H = (H + 1) / 2; W = (W + 1) / 2;
        }
        float* xin = c->p("rx0" + sfx);
        float* xout = c->p("rx1" + sfx);
        int cin = 64;
        const int couts[4] = {64, 128, 256, 512};
        for (int st = 0; st < 4; ++st) {
            const int cout = couts[st];
            for (int unit = 1; unit <= 2; ++unit) {
                const std::string pfx = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(unit);
                const bool first = unit == 1 && cin != cout;
                const int stride = first ? 2 : 1;
                int Ho = 0, Wo = 0;
                const float* shortcut = xin;
                if (first) {   // 1x1/2 projection, no bias, no BN (resnet.py:211-212)
                    IgemmDesc d = conv_desc(xin, H, W, cin, cin, c->p("pk:" + pfx + "/shortcut/weights"), 1, 1, 2, 2, true,
                                            cout, c->p("rsc" + sfx), cout, Ho, Wo);
                    layer = pfx + "/shortcut";
                    gemm(d, 1, false);
                    shortcut = c->p("rsc" + sfx);
                }
                // Pre-split planes pay where the tensors are small next to the contraction: per residual block the plane-writing
                // passes cost 77 / 38 / 27 / 22 us (stage 2..5, batch 32) against ~30 / 15 / 8 / 5 us for the fp32 BN passes they
                // replace, while conv3p saves ~19 us per conv at every stage (profiles/r02_*): stage 2 stays on igemm3dw.
                const bool p3_here = c->use_p3 && st + 2 >= c->p3_from_stage;
                void* planes = p3_here ? (void*)c->p("p3" + sfx) : nullptr;
                // stride-1 conv_1: its input planes were written by the pool / the previous block's merge
                conv_bn(xin, H, W, cin, pfx + "/conv_1", 3, stride, cout, BnRef(), c->p("ry1" + sfx), Ho, Wo, li, "",
                        stride == 1 ? planes : nullptr);
                const BnRef bn1 = bn_ref(li, pfx + "/conv_1", (long)B * Ho * Wo);
                ++li;
                int H2, W2;
                if (p3_here) {
                    // relu(bn1(y1)) -> planes (one elementwise pass), conv_2 on the planes, then the residual merge, which also
                    // writes the planes of the block output when the next conv_1 is a stride-1 3x3 (unit 1 of a stage)
                    layer = pfx + "/bn1-relu";
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("ry1" + sfx), nullptr, nullptr, bn1, nullptr, 1, nullptr, planes, B, Ho, Wo, cout, s); });
                    conv_bn(nullptr, Ho, Wo, cout, pfx + "/conv_2", 3, 1, cout, BnRef(), c->p("ry2" + sfx), H2, W2, li, "", planes);
                    const BnRef bn2 = bn_ref(li, pfx + "/conv_2", (long)B * Ho * Wo);
                    layer = pfx + "/merge";
                    const bool next_p3 = unit == 1;      // the next conv_1 is a stride-1 3x3 of this stage
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("ry2" + sfx), nullptr, nullptr, bn2, shortcut, 1, xout, next_p3 ? planes : nullptr, B, Ho, Wo, cout, s); });
                    ++li;
                    std::swap(xin, xout);
                    H = Ho; W = Wo; cin = cout;
                    continue;
                }
                // conv_2 input = relu(bn1(y1)): on the fly in the conv's fragment path, or materialised once
                // (cheaper for the small late-stage tensors, where every wave would redo the transform)
                const std::string l2 = pfx + "/conv_2";
                auto run_prologue = [&] { conv_bn(c->p("ry1" + sfx), Ho, Wo, cout, l2, 3, 1, cout, bn1, c->p("ry2" + sfx), H2, W2, li); };
                auto run_materialized = [&](const std::string& key) {
                    layer = pfx + "/bn1-relu";
                    timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(c->p("ry1" + sfx), nullptr, nullptr, bn1, nullptr, c->p("ry1n" + sfx), (long)B * Ho * Wo, cout, s); });
                    conv_bn(c->p("ry1n" + sfx), Ho, Wo, cout, l2, 3, 1, cout, BnRef(), c->p("ry2" + sfx), H2, W2, li, key);
                };
                if (c->tuning && !rc) {
                    run_prologue();
                    const float t_pro = c->plan[l2].us;
                    (void)hipEventRecord(c->tune_e0, s);
                    (void)bn_apply_relu_launch(c->p("ry1" + sfx), nullptr, nullptr, bn1, nullptr, c->p("ry1n" + sfx), (long)B * Ho * Wo, cout, s);
                    (void)hipEventRecord(c->tune_e1, s);
                    (void)hipEventSynchronize(c->tune_e1);
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, c->tune_e0, c->tune_e1);
                    if (!rc && hipMemsetAsync(bn_acc(li), 0, (size_t)2 * cout * sizeof(double), s) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "memset failed");
                    run_materialized(l2 + "#mat");
                    const float t_mat = c->plan[l2 + "#mat"].us + ms * 1e3f;
                    c->materialize[l2] = t_mat < t_pro;
                    if (!c->materialize[l2] && !rc) {        // leave the prologue result in place
                        if (hipMemsetAsync(bn_acc(li), 0, (size_t)2 * cout * sizeof(double), s) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "memset failed");
                        c->tuning = false; run_prologue(); c->tuning = true;
                    }
                } else if (c->materialize.count(l2) && c->materialize[l2]) {
                    run_materialized(l2 + "#mat");
                } else {
                    run_prologue();
                }
                const BnRef bn2 = bn_ref(li, pfx + "/conv_2", (long)B * Ho * Wo);
                layer = pfx + "/merge";
                timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(c->p("ry2" + sfx), nullptr, nullptr, bn2, shortcut, xout, (long)B * Ho * Wo, cout, s); });
                ++li;
                std::swap(xin, xout);
                H = Ho; W = Wo; cin = cout;
            }
        }
        return xin;
    }
};

}  // namespace sagen

int sagen_forward_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s) {
    if (!c || !audio || !out) return fail(SAGEN_ERR_NULL, "sagen_forward: null argument");
    if (!c->bound) return fail(SAGEN_ERR_WEIGHTS, "sagen_forward: weights are not bound");
    if (c->has_video && !video) return fail(SAGEN_ERR_NULL, "sagen_forward: video encoder enabled but video is NULL");
    if (c->has_flow && !flow) return fail(SAGEN_ERR_NULL, "sagen_forward: flow encoder enabled but flow is NULL");
    const int B = c->B;
    c->events_used = 0;
    c->prof.clear();

    // Two launch streams: `f` (the caller's) carries the video trunk and everything after the bottleneck; `g`
    // (context-owned) carries the independent audio chain and, with three encoders, the flow trunk.  They fork at
    // entry and join before the localisation FCs.  While autotuning, or without visual encoders, everything
    // stays on the caller's stream.
    static const bool one_stream = getenv("SAGEN_ONE_STREAM") != nullptr;
    const bool forked = c->aux && !c->tuning && !one_stream && (c->has_video || c->has_flow);
    Fwd f{c, s};
    Fwd g{c, forked ? c->aux : s};
    // error exit: work may still be queued on the context's stream, reading the caller's tensors / the workspace - the caller's
    // stream must not run ahead of it (torch's caching allocator could hand that memory out again on `s`)
    auto bail = [&](int rc) {
        if (forked) {
            if (hipEventRecord(c->ev_join, c->aux) == hipSuccess) (void)hipStreamWaitEvent(s, c->ev_join, 0);
            else (void)hipStreamSynchronize(c->aux);
        }
        return rc;
    };
    if (forked) {
        g.wsname = "splitk_aux";
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_fork, s));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    }

    // ---- stream g: STFT (myutils.py:119-147) -> |.| of frames 46:173 (model.py:166-178) + spectrum of frames 89:117
    g.layer = "stft";
    g.timed("stft_kernel", 0.0, [&] { return stft_launch(audio, B, c->snd_size, 46, 173, c->p("mag"), 89, 117, c->p("spec"), g.s); });
    if (forked) {
        // The LDS FFT kernels give wrong results when bf16x3 contraction waves of ANOTHER stream share their CUs
        // (open): the first matrix launch of the main stream waits for the STFT (it overlaps the pad kernel).
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_stft, c->aux));
        f.wait_before_mfma = c->ev_stft;
    }

    // audio encoder (model.py:161-187): conv l writes the encoder half of concat buffer l
    for (int l = 0; l < 5 && !g.rc; ++l) {
        const std::string name = "audio_encoder/conv" + std::to_string(l + 1);
        const int ce = c->enc_c[l + 1];
        float* y = c->p("cat" + std::to_string(l + 1)) + (l == 4 ? 0 : ce);
        int Ho, Wo;
        IgemmDesc d;
        if (l == 0) {
            // Cin = 1: the 16 taps along frequency are contiguous floats -> treat kw as 16 channels of a 7-tap conv
            d = g.conv_desc(c->p("mag"), 127, 1024, 16, 1, c->p("pk:" + name + "/weights"), 7, 1, 4, 8, false, ce, y, 2 * ce, Ho, Wo);
            Wo = c->enc_w[1];
            d.M = B * Ho * Wo; d.Wg = Wo; d.Wlim = Wo;
            d.y_rstride = (long)Wo * 2 * ce; d.y_bstride = (long)Ho * Wo * 2 * ce;
        } else {
            const int cp = c->enc_c[l];
            const float* x = c->p("cat" + std::to_string(l)) + (l == 5 ? 0 : cp);
            d = g.conv_desc(x, c->enc_h[l], c->enc_w[l], cp, 2 * cp, c->p("pk:" + name + "/weights"), AENC_K[l][0], AENC_K[l][1],
                            AENC_S[l][0], AENC_S[l][1], false, ce, y, 2 * ce, Ho, Wo);
        }
        d.bias = c->v(name + "/biases");
        d.relu_out = 1;
        g.layer = name;
        g.gemm(d);
    }

    // bottleneck (model.py:203-239)
    float* bott = c->p("bott");
    {   // audio-fc over (w, c) of conv5: 6 taps along W, 512 channels each
        IgemmDesc d;
        d.x = c->p("cat5"); d.w = c->p("pk:bottleneck/audio-fc/weights"); d.y = bott; d.bias = c->v("bottleneck/audio-fc/biases");
        d.M = B * 3; d.N = 1024; d.K = 6 * 512; d.Kpad = d.K;
        d.Hg = 3; d.Wg = 1; d.Hin = 3; d.Win = 6; d.Cin = 512; d.ldx = 1024; d.x_bstride = 3L * 6 * 1024;
        d.ntaps = 6; d.TW = 6; d.log2Cin = 9;
        d.Cout = 1024; d.Hlim = 3; d.Wlim = 1; d.ldy = c->Cb; d.y_rstride = c->Cb; d.y_bstride = 3L * c->Cb;
        d.relu_out = 1;
        g.layer = "bottleneck/audio-fc";
        g.gemm(d);
    }
    // visual encoders: video on f; flow on g when both exist (its own buffer set "_b"), else on f
    int choff = 1024;
    for (int e = 0; e < 2; ++e) {
        const bool on = e == 0 ? c->has_video : c->has_flow;
        if (!on) continue;
        const std::string enc = e == 0 ? "video" : "flow";
        Fwd& w = (e == 1 && c->has_video) ? g : f;
        w.sfx = (e == 1 && c->has_video) ? "_b" : "";
        const float* feat = w.resnet(e == 0 ? video : flow, enc + "_encoder");          // [B,7,14,512]
        w.fc(feat, B * 98, 512, 512, "bottleneck/" + enc + "-fc-red", 128, true, c->p("fcred" + w.sfx), 128);
        w.fc(c->p("fcred" + w.sfx), B, 98 * 128, 98 * 128, "bottleneck/" + enc + "-fc", 512, true, bott + choff, c->Cb, 3);   // tile x3 (model.py:230-232)
        choff += 512;
    }
    if (f.rc || g.rc) return bail(f.rc ? f.rc : g.rc);
    if (forked) {
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_join, c->aux));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }

    // localization (model.py:241-271).  With the mask decoder present the localisation FCs and the deconv chain are
    // independent consumers of the bottleneck: second fork, the FCs go to the context's stream.
    static const bool no_fork2 = getenv("SAGEN_NO_FORK2") != nullptr;
    const bool fork2 = forked && c->freq_mask && !no_fork2;
    if (fork2) {
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_fork, s));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    }
    {
        Fwd& w = fork2 ? g : f;
        const float* x = bott;
        int K = c->Cb;
        for (int i = 0; i < c->cfg.n_loc_units; ++i) {
            float* y = c->p("loc" + std::to_string(i + 1));
            w.fc(x, B * 3, K, K, "localization/fc" + std::to_string(i + 1), c->cfg.loc_units[i], true, y, c->cfg.loc_units[i]);
            x = y; K = c->cfg.loc_units[i];
        }
        const int nlast = 3 * (c->nsep + 1);
        w.fc(x, B * 3, K, K, "localization/fc" + std::to_string(c->cfg.n_loc_units + 1), nlast, false, c->p("coeffs"), nlast);
        if (f.rc || g.rc) return bail(f.rc ? f.rc : g.rc);
    }

    if (!c->freq_mask) {
        f.layer = "decoder";
        f.timed("nosep_mix_kernel", 0.0, [&] { return nosep_mix_launch(audio, c->p("coeffs"), out, B, c->snd_size, c->snd_contx, c->snd_dur, 3, s); });
        return f.rc;
    }

    // separation (model.py:282-348)
    f.fc(bott, B * 3, c->Cb, c->Cb, "separation/fc-feats", 512, true, c->p("cat5") + 512, 1024, 6);     // tile over 6 freq columns
    for (int l = 4; l >= 1; --l) {
        const int Cin = 2 * c->enc_c[l + 1];
        f.deconv(c->p("cat" + std::to_string(l + 1)), c->enc_h[l + 1], c->enc_w[l + 1], Cin, l, c->p("cat" + std::to_string(l)),
                 2 * c->enc_c[l], true, 0, 0, 0, 0, 0);
    }
    // deconv1: only output rows 44..66 (mask frames 1..23) reach the cropped window -> grid rows a = 11..16
    f.deconv(c->p("cat1"), 31, 127, 64, 0, c->p("dmask"), c->nsep, false, 11, 17, 67, 23L * 1024 * c->nsep, 44);
    if (fork2) {                                         // the mix needs the localisation coefficients
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_join, c->aux));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    f.layer = "separation/mask-istft-mix";
    f.timed("mask_istft_kernel+ola_mix_kernel", 0.0, [&] {
        return mask_istft_mix_launch(c->p("dmask"), 23L * 1024 * c->nsep, 1, c->p("spec"), c->p("coeffs"), B, c->nsep, out,
                                     c->p("frames"), s); });
    return f.rc;
}
void sagen_destroy_impl(sagen_ctx* c) {
    if (!c) return;
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->tune_e0) { (void)hipEventDestroy(c->tune_e0); (void)hipEventDestroy(c->tune_e1); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_stft) (void)hipEventDestroy(c->ev_stft);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    delete c;
}


// times every (tile, split-K) candidate of every contraction on the given inputs and stores the plan
int sagen_autotune_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s) {
    if (!c) return fail(SAGEN_ERR_NULL, "sagen_autotune: null ctx");
    if (getenv("SAGEN_NO_AUTOTUNE")) return SAGEN_OK;
    if (!c->tune_e0) {
        SAGEN_HIP_CHECK(hipEventCreate(&c->tune_e0));
        SAGEN_HIP_CHECK(hipEventCreate(&c->tune_e1));
    }
    c->plan.clear();
    c->materialize.clear();
    c->tuning = true;
    const int rc = sagen_forward_impl(c, audio, video, flow, out, s);
    c->tuning = false;
    if (rc) { c->plan.clear(); return rc; }
    SAGEN_HIP_CHECK(hipStreamSynchronize(s));
    return SAGEN_OK;
}

int sagen_plan_set_impl(sagen_ctx* c, const char* layer, int tile, int splitk) {
    const std::string name(layer);
    const std::string sfx = "#materialize";       // pseudo-entry: apply the producer's BN+ReLU in its own pass?
    if (name.size() > sfx.size() && name.compare(name.size() - sfx.size(), sfx.size(), sfx) == 0) {
        c->materialize[name.substr(0, name.size() - sfx.size())] = splitk != 0;
        return SAGEN_OK;
    }
    if (tile < 0 || tile >= (int)TILE_AUTO || splitk < 1 || splitk > 64) return fail(SAGEN_ERR_SHAPE, "sagen_plan_set: tile=%d splitk=%d", tile, splitk);
    Choice ch;
    ch.tile = tile; ch.splitk = splitk;
    c->plan[layer] = ch;
    return SAGEN_OK;
}

// "layer\ttile\tsplitk\tmicroseconds\n" per contraction of the current plan
int sagen_plan_describe_impl(sagen_ctx* c, char* buf, size_t buflen) {
    std::string out;
    for (const auto& kv : c->plan) {
        char line[512];
        snprintf(line, sizeof line, "%s\t%s\t%d\t%.2f\n", kv.first.c_str(), igemm_tile_name((IgemmTile)kv.second.tile),
                 kv.second.splitk, kv.second.us);
        out += line;
    }
    int n = (int)c->plan.size();
    for (const auto& kv : c->materialize) {
        out += kv.first + "#materialize\t-\t" + (kv.second ? "1" : "0") + "\t0\n";
        ++n;
    }
    if (out.size() + 1 > buflen) return fail(SAGEN_ERR_WORKSPACE, "plan description needs %zu bytes", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return n;
}

int sagen_profile_enable_impl(sagen_ctx* c, int on) {
    c->profiling = on != 0;
    c->prof.clear();
    c->events_used = 0;
    return SAGEN_OK;
}

// one line per launch of the last forward: kernel \t layer \t microseconds \t flops
int sagen_profile_report_impl(sagen_ctx* c, char* buf, size_t buflen) {
    std::string out;
    for (const ProfRec& r : c->prof) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (e != hipSuccess) return fail(SAGEN_ERR_HIP, "profile: %s", hipGetErrorString(e));
        char line[512];
        snprintf(line, sizeof line, "%s\t%s\t%.3f\t%.6g\n", r.kernel.c_str(), r.layer.c_str(), ms * 1e3, r.flops);
        out += line;
    }
    if (out.size() + 1 > buflen) return fail(SAGEN_ERR_WORKSPACE, "profile report needs %zu bytes", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)c->prof.size();
}
size_t sagen_workspace_bytes_impl(const sagen_ctx* c) { return c->ws_floats * sizeof(float); }
int sagen_num_variables_impl(const sagen_ctx* c) { return (int)c->vars.size(); }
int sagen_variable_spec_impl(const sagen_ctx* c, int i, const char** name, int32_t* ndim, int64_t shape[4]) {
    if (i < 0 || i >= (int)c->vars.size()) return fail(SAGEN_ERR_SHAPE, "variable index %d out of range", i);
    *name = c->vars[i].name.c_str();
    *ndim = c->vars[i].ndim;
    for (int k = 0; k < 4; ++k) shape[k] = c->vars[i].shape[k];
    return SAGEN_OK;
}
int sagen_get_intermediate_impl(const sagen_ctx* c, const char* name, const float** data, int32_t* ndim, int64_t shape[4],
                                int64_t* pixel_stride) {
    if (!c->ws) return fail(SAGEN_ERR_WORKSPACE, "no workspace bound");
    auto it = c->named.find(name);
    if (it == c->named.end()) return fail(SAGEN_ERR_SHAPE, "unknown intermediate %s", name);
    const Named& nm = it->second;
    *data = c->ws + nm.buf.off + nm.extra_off;
    *ndim = nm.ndim;
    for (int k = 0; k < 4; ++k) shape[k] = nm.shape[k];
    *pixel_stride = nm.pixel_stride;
    return SAGEN_OK;
}
