// The training step of the path (reference train.py:137-236): forward with retained activations, the loss `stft/avg`
// (model.py:122-127, 156-159), and the backward pass of SptAudioGen.inference_ops (model.py:356-434) - what tf.gradients builds
// under opt.minimize (myutils.py:220-221) - written as explicit launches:
//   * weight gradients: wgrad_kernel (wgrad.hip), straight into the caller's gradient buckets (TF variable layouts);
//   * data gradients: the forward's own implicit-GEMM kernels on re-packed filters - a stride-1 conv's dx is a conv of dy with
//     the tap-reversed / channel-transposed filter; a strided conv's dx is its conv2d_transpose (depth-to-space form); a
//     conv2d_transpose's dx is the strided VALID conv of dy; an FC's dx is dy . W^T;
//   * training-mode batch-norm backward, ReLU masks, bias sums, max-pool routing, tile / concat fan-in (backward.hip);
//   * the mask -> iSTFT -> mix adjoint (fft.hip).
// Nothing before the first trainable layer is differentiated (the STFT magnitude and the video frame have no parameters
// upstream).  The optimiser (fused Adam over flat buckets) and the gradient all-reduce live above the ABI (train.py).
#include "model.h"

namespace sagen {

static const int RS_H[4] = {56, 28, 14, 7}, RS_W[4] = {112, 56, 28, 14}, RS_C[4] = {64, 128, 256, 512};
constexpr int CACC_SLOT = 1024;            // doubles per bias-gradient accumulator
constexpr int CACC_SLOTS = 48;
constexpr size_t WGWS_FLOATS = (size_t)24 << 20;

static inline int ceil4(int x) { return (x + 3) / 4 * 4; }
static inline int ceil16(int x) { return (x + 15) / 16 * 16; }

enum PkdKind { PKD_NONE = 0, PKD_FLIPT, PKD_STRIDED, PKD_DECONV, PKD_ROWS };

// the stride-1 3x3 data gradients of the trunks on fp16x2 planes (needs the plane kernels and the fp16x2 format of the forward)
static bool h2d_on(const sagen_ctx* c) { return c->train_h2d && c->train_h2 && c->use_h2 && c->use_p3 && !c->fp32_only; }
// ... and their weight gradients (wgrad3h.hip; the forward then retains its activation planes)
static bool h2w_on(const sagen_ctx* c) { return h2d_on(c) && c->train_h2w && c->p3_from_stage <= 2 && wgrad_planes_enabled(); }

struct PkdSpec { PkdKind kind = PKD_NONE; int N = 0, K = 0, kh = 0, kw = 0, sh = 1, sw = 1, a = 0, b = 0; };

// how the data-gradient filter of a "/weights" variable is packed
static PkdSpec pkd_spec(const VarSpec& vs) {
    PkdSpec p;
    const std::string& n = vs.name;
    if (n.find("/deconv") != std::string::npos) {                   // [kh,kw,Cout,Cin]: dx = VALID strided conv of dy
        p.kind = PKD_DECONV; p.kh = (int)vs.shape[0]; p.kw = (int)vs.shape[1]; p.a = (int)vs.shape[2]; p.b = (int)vs.shape[3];
        p.N = p.b; p.K = p.kh * p.kw * p.a;
    } else if (vs.ndim == 2) {                                      // [in,out]: dx = dy . W^T
        p.kind = PKD_ROWS; p.a = (int)vs.shape[0]; p.b = (int)vs.shape[1];
        p.N = p.a; p.K = ceil4(p.b);
    } else if (n.rfind("audio_encoder/", 0) == 0) {
        const int l = n[std::string("audio_encoder/conv").size()] - '1';
        if (l == 0) return p;                                       // the spectrogram has no gradient
        p.kind = PKD_STRIDED; p.kh = AENC_K[l][0]; p.kw = AENC_K[l][1]; p.sh = AENC_S[l][0]; p.sw = AENC_S[l][1];
        p.a = (int)vs.shape[2]; p.b = (int)vs.shape[3];
        p.N = p.sh * p.sw * p.a; p.K = cdiv(p.kh, p.sh) * cdiv(p.kw, p.sw) * p.b;
    } else if (n.find("/conv1/conv/") != std::string::npos) {
        return p;                                                   // the video frame has no gradient
    } else {                                                        // ResNet 3x3 / 1x1
        p.kh = (int)vs.shape[0]; p.kw = (int)vs.shape[1]; p.a = (int)vs.shape[2]; p.b = (int)vs.shape[3];
        const bool strided = n.find("/shortcut/") != std::string::npos || (n.find("/conv_1/") != std::string::npos && p.a != p.b);
        if (strided) { p.kind = PKD_STRIDED; p.sh = p.sw = 2; p.N = 4 * p.a; p.K = cdiv(p.kh, 2) * cdiv(p.kw, 2) * p.b; }
        else { p.kind = PKD_FLIPT; p.N = p.a; p.K = p.kh * p.kw * p.b; }
    }
    return p;
}

// The stride-2 convs of the ResNet trunk (3x3 conv_1 and 1x1 shortcut of the first block of a stage; no padding before): their
// input gradient in the depth-to-space form is a 2x2-tap contraction for each of the 4 output phases, but phase (ry, rx) only has
// the taps with ry + 2 dp < kh, rx + 2 dq < kw - 9 of the 16 (tap, phase) blocks of a 3x3, 1 of the 4 of a 1x1.  One contraction
// per phase over its own taps, writing every second pixel of every second row (SAGEN_DGRAD_UNPHASED=1: the single padded form).
// Measured at batch 32 (tools/train_profile.py): the 1x1 shortcuts 31 / 36 / 41 -> 15 / 15 / 19 us; the 3x3 of stage 3 97 -> 90 us, but
// those of stages 4 and 5 88 -> 98 and 92 -> 137 us: four contractions of 196 tiles each, no split-K into a strided output - so the
// 3x3 goes phase-wise only where a phase still fills the chip (>= 512 tiles of 64 x 64).
static bool dgrad_phased(int batch, const std::string& n, const PkdSpec& p) {
    static const bool off = getenv("SAGEN_DGRAD_UNPHASED") != nullptr;
    if (off || p.kind != PKD_STRIDED || p.sh != 2 || p.sw != 2 || n.rfind("audio_encoder/", 0) == 0 || p.kh != p.kw) return false;
    if (p.kh == 1) return true;
    if (p.kh != 3) return false;
    const size_t at = n.find("_encoder/conv");
    if (at == std::string::npos) return false;
    const int st = n[at + 13] - '2';                              // "convS_1": stage index S - 2
    if (st < 1 || st > 3) return false;
    const long tiles = (long)cdiv(batch * RS_H[st] * RS_W[st], 64) * cdiv(p.a, 64);
    static const long need = getenv("SAGEN_DGRAD_PHASE_TILES") ? atol(getenv("SAGEN_DGRAD_PHASE_TILES")) : 512;      // (tests: 0 = always)
    return tiles >= need;
}
static int phase_taps(int k, int r) { return (k - r + 1) / 2; }      // taps dp >= 0 with r + 2 dp < k

static bool is_weights(const std::string& n) { return n.size() >= 8 && n.compare(n.size() - 8, 8, "/weights") == 0; }

static void train_carve(sagen_ctx* c) {
    if (c->tws_floats) return;
    const int B = c->B, nsep = c->nsep, Cb = c->Cb;
    const int ldc = ceil4(3 * (nsep + 1));
    c->talloc("t:dpred", (size_t)B * 4800 * 3);
    c->talloc("t:pred", (size_t)B * 4800 * 3);
    c->talloc("t:loss", 64);
    c->talloc("t:cacc", (size_t)CACC_SLOTS * CACC_SLOT * 2);
    c->talloc("t:wgws", WGWS_FLOATS);
    c->talloc("t:redws", reduce_scratch_floats(1024));
    c->talloc("t:redws2", reduce_scratch_floats(1024));       // ... of the launcher on the second stream
    c->talloc("t:ddmask", (size_t)B * 31 * 1024 * nsep);
    c->talloc("t:mbw", mask_istft_bwd_scratch_floats(B, nsep));
    c->talloc("t:dcoeffs", (size_t)B * 3 * ldc);
    for (int i = 0; i < c->cfg.n_loc_units; ++i) {
        c->talloc("t:g:loc" + std::to_string(i + 1), (size_t)B * 3 * c->cfg.loc_units[i]);
        c->talloc("t:dy:loc" + std::to_string(i + 1), (size_t)B * 3 * c->cfg.loc_units[i]);
    }
    c->talloc("t:g:bott_loc", (size_t)B * 3 * Cb);
    c->talloc("t:g:bott_sep", (size_t)B * 3 * Cb);
    for (int l = 1; l <= 5; ++l) {
        const size_t px = (size_t)B * c->enc_h[l] * c->enc_w[l];
        c->talloc("t:dcat" + std::to_string(l), px * 2 * c->enc_c[l]);
        c->talloc("t:dy:conv" + std::to_string(l), px * c->enc_c[l]);
        if (l <= 4) {
            c->talloc("t:dydec" + std::to_string(l), px * c->enc_c[l]);
            c->talloc("t:g:conv" + std::to_string(l), px * c->enc_c[l]);
        }
    }
    c->talloc("t:g:fcfeats", (size_t)B * 3 * 512);
    c->talloc("t:dy:fcfeats", (size_t)B * 3 * 512);
    c->talloc("t:dy:audiofc", (size_t)B * 3 * 1024);
    c->talloc("t:g:conv5_fc", (size_t)B * 3 * 3072);
    for (int set = 0; set < 2; ++set) {
        const std::string x = set ? "_b" : "";
        if (set == 0 ? !(c->has_video || c->has_flow) : !(c->has_video && c->has_flow)) continue;
        const size_t stage = (size_t)B * 56 * 112 * 64;
        c->talloc("t:x0" + x, stage);
        for (int k = 0; k < 8; ++k) {
            const size_t sz = (size_t)B * RS_H[k / 2] * RS_W[k / 2] * RS_C[k / 2];
            for (const char* nm : {"t:y1:", "t:a1:", "t:y2:", "t:out:"}) c->talloc(nm + std::to_string(k) + x, sz);
        }
        // gradient activations of the trunk.  The weight gradients run on the context's second stream, so what they read (dy of conv_2 /
        // conv_1, dz of the merge) is double-buffered over the block parity: block k may overwrite only what block k+2 left behind
        for (const char* nm : {"t:A0", "t:A1", "t:Z0", "t:Z1", "t:DYc0", "t:DYc1", "t:DYd0", "t:DYd1", "t:DA"}) c->talloc(nm + x, stage);
        // d(block input) through the 1x1 stride-2 shortcut: one buffer per stage, because the phase-wise data gradient only ever writes
        // phase (0, 0) and relies on the other three being zero since bind (a shared buffer would keep another stage's values there)
        for (int st = 1; st <= 3; ++st) c->talloc("t:S" + std::to_string(st) + x, stage);
        c->talloc("t:dz0" + x, (size_t)B * 112 * 224 * 64);
        if (set == 0 && c->has_video) c->talloc("t:s8plane" + x, stem8_plane_bytes(B) / sizeof(float) + 64);     // uint8 frames: the centred bf16 plane (stem8.hip)
        c->talloc("t:bnbacc" + x, (size_t)24 * 2 * 512 * 2);
        if (h2d_on(c)) {                   // dy of a stride-1 3x3 conv as fp16x2 planes (largest: stage 2) + the reduce pass's per-workgroup maxima
            // (double-buffered over the block parity like the fp32 dy: the weight gradients on the second stream read them too)
            for (const char* nm : {"t:DPc0", "t:DPc1", "t:DPd0", "t:DPd1"}) c->talloc(nm + x, p3h_bytes(B, 56, 112, 64) / sizeof(float));
            c->talloc("t:mxpart" + x, 2 * 1024);
        }
        if (h2w_on(c)) {                   // retained activation planes: the input of every stride-1 3x3 conv (model.h: resnet_train)
            for (int k = 0; k < 8; ++k) {
                const size_t n = p3h_bytes(B, RS_H[k / 2], RS_W[k / 2], RS_C[k / 2]) / sizeof(float);
                c->talloc("t:pl:a1:" + std::to_string(k) + x, n);
                // the ReLU mask of the block output, one bit per element (written by the merge pass, read by conv_2's batch-norm backward)
                c->talloc("t:mb:" + std::to_string(k) + x, ((size_t)B * RS_H[k / 2] * RS_W[k / 2] * RS_C[k / 2] / 8 + 3) / 4);
                {   // the block input (stride-2 first blocks: the previous stage's geometry)
                    const int ki = (k > 0 && !(k & 1)) ? k - 1 : k;
                    c->talloc("t:pl:x:" + std::to_string(k) + x, p3h_bytes(B, RS_H[ki / 2], RS_W[ki / 2], RS_C[ki / 2]) / sizeof(float));
                }
            }
            c->talloc("t:h2a" + x, 32);
        }
        c->talloc("t:stemtmp" + x, (size_t)7 * 32 * 64);
        c->talloc("t:g:feat" + x, (size_t)B * 98 * 512);
        c->talloc("t:g:fcred" + x, (size_t)B * 98 * 128);
        c->talloc("t:dy:fcred" + x, (size_t)B * 98 * 128);
        c->talloc("t:g:vfc" + x, (size_t)B * 512);
        c->talloc("t:dy:vfc" + x, (size_t)B * 512);
    }
    size_t h2d_pack_blocks = 0;
    for (const auto& vs : c->vars) {
        if (!is_weights(vs.name)) continue;
        const PkdSpec p = pkd_spec(vs);
        if (p.kind == PKD_NONE) continue;
        if (dgrad_phased(c->B, vs.name, p)) {
            for (int ph = 0; ph < 4; ++ph) {
                const int nt = phase_taps(p.kh, ph >> 1) * phase_taps(p.kw, ph & 1);
                if (nt > 0) c->talloc("pkd:" + vs.name + "#p" + std::to_string(ph), packed_split_floats((size_t)p.a * ceil16(nt * p.b)));
            }
        } else {
            c->talloc("pkd:" + vs.name, packed_split_floats((size_t)p.N * ceil16(p.K)));
            if (p.kind == PKD_FLIPT && p.kh == 3 && h2d_on(c)) {           // ... and as two fp16 planes for conv3h_kernel
                c->talloc("pkdh:" + vs.name, (size_t)p.N * ceil16(p.K));
                h2d_pack_blocks += (size_t)p.N * ceil16(p.K) / 1024 + 1;
                const int slot = 8 + (int)c->h2d_slot.size();
                c->h2d_slot[vs.name] = slot;
            }
        }
    }
    if (h2d_on(c)) {
        c->talloc("t:h2d", 8 + c->h2d_slot.size() + 8);          // [4 * trunk + 2 * (conv_1) + parity]: 2^-kd of that dy plane buffer; [8..]: 2^-kw per filter
        c->talloc("t:h2d:jobs", (c->h2d_slot.size() + 1) * sizeof(H2Job) / sizeof(float) + 64);
        c->talloc("t:h2d:amax", c->h2d_slot.size() + h2d_pack_blocks + 64);
    }
    c->talloc("pkd:jobs", (c->vars.size() + 64) * sizeof(PackJob) / sizeof(float) + 64);
}

// data-gradient filter packs, once per step (the optimiser rewrote the variables): one batched launch (igemm.hip: pack_multi_kernel)
static int repack_dgrad(sagen_ctx* c, hipStream_t s) {
    if (c->pack_jobs_bwd.empty()) {
        for (const auto& vs : c->vars) {
            if (!is_weights(vs.name)) continue;
            const PkdSpec p = pkd_spec(vs);
            if (p.kind == PKD_NONE) continue;
            PackJob j;
            j.src = c->v(vs.name);
            if (dgrad_phased(c->B, vs.name, p)) {
                for (int ph = 0; ph < 4; ++ph) {
                    const int ry = ph >> 1, rx = ph & 1;
                    const int nth = phase_taps(p.kh, ry), ntw = phase_taps(p.kw, rx);
                    if (nth * ntw == 0) continue;
                    PackJob q = j;
                    q.dst = c->p("pkd:" + vs.name + "#p" + std::to_string(ph));
                    q.kind = PACK_DECONV_PHASE; q.N = p.a; q.Kpad = ceil16(nth * ntw * p.b);
                    q.p[0] = p.kh; q.p[1] = p.kw; q.p[2] = p.a; q.p[3] = p.b; q.p[4] = ry; q.p[5] = rx; q.p[6] = ntw;
                    c->pack_jobs_bwd.push_back(q);
                }
                continue;
            }
            j.dst = c->p("pkd:" + vs.name);
            j.N = p.N; j.Kpad = ceil16(p.K);
            switch (p.kind) {
                case PKD_FLIPT: j.kind = PACK_FLIPT; j.p[0] = p.kh * p.kw; j.p[1] = p.a; j.p[2] = p.b; break;
                case PKD_STRIDED:                                   // HWIO read as [kh,kw,"Cout"=Cin,"Cin"=Cout]
                    j.kind = PACK_DECONV; j.p[0] = p.kh; j.p[1] = p.kw; j.p[2] = p.a; j.p[3] = p.b; j.p[4] = p.sh; j.p[5] = p.sw; j.p[6] = cdiv(p.kw, p.sw); break;
                case PKD_DECONV:                                    // [kh,kw,Cout,Cin] read as HWIO
                    j.kind = PACK_CONV; j.p[0] = p.kh * p.kw; j.p[1] = p.a; j.p[2] = p.a; j.p[3] = p.b; break;
                default: j.kind = PACK_ROWS; j.p[0] = p.a; j.p[1] = p.b; break;
            }
            c->pack_jobs_bwd.push_back(j);
        }
        if (c->pack_jobs_bwd.size() * sizeof(PackJob) > c->tbufs.at("pkd:jobs").n * sizeof(float)) return fail(SAGEN_ERR_WORKSPACE, "pack job table too small");
        c->pack_blocks_bwd = sagen_upload_pack_jobs(c->pack_jobs_bwd, c->p("pkd:jobs"), s);
        if (c->pack_blocks_bwd < 0) return fail(SAGEN_ERR_HIP, "pack job upload failed");
    }
    int rc = pack_multi_launch(reinterpret_cast<const PackJob*>(c->p("pkd:jobs")), (int)c->pack_jobs_bwd.size(), c->pack_blocks_bwd, s);
    if (rc || !h2d_on(c) || c->h2d_slot.empty()) return rc;
    // the flipped / transposed 3x3 filters of the stride-1 trunk convs also as fp16x2 planes, from the fp32 packs just written
    if (c->h2d_jobs.empty()) {
        int nb = 0;
        for (const auto& kv : c->h2d_slot) {
            const VarSpec* vs = nullptr;
            for (const auto& v : c->vars) if (v.name == kv.first) vs = &v;
            if (!vs) return fail(SAGEN_ERR_WEIGHTS, "no variable %s", kv.first.c_str());
            const PkdSpec p = pkd_spec(*vs);
            H2Job j;
            j.N = p.N; j.Kpad = ceil16(p.K);
            j.wp = c->p("pkd:" + vs->name); j.w2 = c->p("pkdh:" + vs->name); j.w_inv = c->p("t:h2d") + kv.second;
            j.first_block = nb;
            nb += (int)(((long)j.N * j.Kpad + 1023) / 1024);
            c->h2d_jobs.push_back(j);
        }
        c->h2d_blocks = nb;
        if (c->h2d_jobs.size() * sizeof(H2Job) > c->tbufs.at("t:h2d:jobs").n * sizeof(float)) return fail(SAGEN_ERR_WORKSPACE, "fp16x2 job table too small");
        if (hipMemcpyAsync(c->p("t:h2d:jobs"), c->h2d_jobs.data(), c->h2d_jobs.size() * sizeof(H2Job), hipMemcpyHostToDevice, s) != hipSuccess)
            return fail(SAGEN_ERR_HIP, "fp16x2 job upload failed");
    }
    return h2_filter_pack_multi_launch(reinterpret_cast<const H2Job*>(c->p("t:h2d:jobs")), (int)c->h2d_jobs.size(), c->h2d_blocks,
                                       reinterpret_cast<unsigned*>(c->p("t:h2d:amax")), s);
}

struct Bwd : Fwd {
    int cslot = 0;
    std::string redws = "t:redws";                                // scratch of this launcher's channel reductions
    bool split_ok = getenv("SAGEN_BWD_NOSPLIT") == nullptr;      // debugging: no split-K in the data-gradient contractions
    Bwd(sagen_ctx* ctx, hipStream_t st) { c = ctx; s = st; }

    // where variable v's gradient goes; the caller is about to enqueue the kernel that writes it (bucket milestones: ms_flush)
    float* grad(const std::string& v) {
        const int i = c->var_index.at(v);
        if (!c->ms_bucket.empty() && c->ms_bucket[i] >= 0 && !c->ms_done[i]) { c->ms_done[i] = 1; c->ms_touched.push_back(i); }
        return c->grad_ptr[i];
    }
    // every gradient requested so far has its producer enqueued (on this stream or on the second one): buckets that are now complete
    // get their event - on the second stream, behind everything the main stream has enqueued (one more wait for a stream that runs
    // behind the main one anyway)
    void ms_flush() {
        if (rc || c->ms_touched.empty()) return;
        for (int i : c->ms_touched) {
            const int b = c->ms_bucket[i];
            if (--c->ms_left[b] != 0) continue;
            hipError_t e = hipSuccess;
            if (wg) {
                e = hipEventRecord(c->ms_tmp, s);
                if (e == hipSuccess) e = hipStreamWaitEvent(wg->s, c->ms_tmp, 0);
                if (e == hipSuccess) e = hipEventRecord(c->ms_event[b], wg->s);
            } else {
                e = hipEventRecord(c->ms_event[b], s);
            }
            if (e != hipSuccess) { rc = fail(SAGEN_ERR_HIP, "bucket milestone: %s", hipGetErrorString(e)); break; }
        }
        c->ms_touched.clear();
    }
    double* cacc_next() {
        if (cslot >= CACC_SLOTS) { if (!rc) rc = fail(SAGEN_ERR_WORKSPACE, "bias accumulator slots exhausted"); return nullptr; }
        return reinterpret_cast<double*>(c->p("t:cacc")) + (size_t)(cslot++) * CACC_SLOT;
    }

    // dy = (ga + gb) * (act > 0); with `bias_var` also the bias gradient sum_rows dy
    // band_rows > 0: only rows [row0, row0 + band_rows) of each batch element's batch_rows (R = B * band_rows); the rest of dy is not touched
    void relu_bwd(const std::string& label, const float* ga, int lda, const float* gb, int ldb, const float* act, int ldact, float* dy,
                  int lddy, long R, int C, const std::string& bias_var, long band_rows = 0, long batch_rows = 0, long row0 = 0) {
        if (rc) return;
        layer = label;
        double* acc = bias_var.empty() ? nullptr : cacc_next();
        if (rc) return;
        timed("relu_bwd_kernel", 0.0, [&] { return relu_bwd_launch(ga, lda, gb, ldb, act, ldact, dy, lddy, R, C, acc, c->p(redws), s, acc ? grad(bias_var) : nullptr,
                                                                   band_rows, batch_rows, row0); });
    }

    WgradDesc wdesc(const float* g, int HG, int WG, int ldg, int Cg, const float* dd, int Hd, int Wd, int ldd, int Cd, int kh, int kw,
                    int sh, int sw, int h0, int w0) {
        WgradDesc d;
        d.g = g; d.d = dd; d.B = c->B; d.Hd = Hd; d.Wd = Wd; d.HG = HG; d.WG = WG; d.ldg = ldg; d.Cg = Cg; d.ldd = ldd; d.Cd = Cd;
        d.g_rstride = (unsigned)((long)WG * ldg); d.g_bstride = (unsigned)((long)HG * WG * ldg);
        d.d_rstride = (unsigned)((long)Wd * ldd); d.d_bstride = (unsigned)((long)Hd * Wd * ldd);
        d.sh = sh; d.sw = sw; d.TH = kh; d.TW = kw; d.h0 = h0; d.w0 = w0;
        return d;
    }
    // Weight gradients go to the context's SECOND stream (`wg`): nothing downstream of the backward chain waits for them, and they
    // are matrix-core bound while the batch-norm / ReLU passes between the data gradients are HBM bound - the two overlap.
    // Ordering: the wgrad waits for everything enqueued so far on the main stream (its operands' producers); the main stream
    // waits (aux_fence) before it overwrites a buffer the second stream may still be reading, and joins it at the end of the step.
    Bwd* wg = nullptr;
    void wgrad_here(const std::string& label, WgradDesc d, float* out) {
        if (rc) return;
        layer = label;
        d.out = out;
        d.ws = c->p("t:wgws");
        d.splitk = wgrad_pick_splitk(d, c->cap("t:wgws"));
        const double flops = 2.0 * d.B * d.Hd * d.Wd * d.TH * d.TW * d.Cg * d.Cd;
        d.defer_reduce = 1;
        timed(wgrad_kernel_name(d), flops, [&] { return wgrad_launch(d, s); });
        if (d.splitk > 1) timed("splitk_reduce_kernel", 0.0, [&] { return wgrad_reduce_launch(d, s); });
    }
    void wgrad(const std::string& label, const WgradDesc& d, float* out) {
        if (rc) return;
        if (!wg) { wgrad_here(label, d, out); return; }
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(wg->s, e, 0) != hipSuccess) { rc = fail(SAGEN_ERR_HIP, "stream fork failed"); return; }
        wg->wgrad_here(label, d, out);
        if (wg->rc) rc = wg->rc;
    }
    // an event after everything enqueued on the second stream so far (nullptr when there is none)
    hipEvent_t aux_mark() {
        if (!wg || rc) return nullptr;
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, wg->s) != hipSuccess) { rc = fail(SAGEN_ERR_HIP, "hipEventRecord failed"); return nullptr; }
        return e;
    }
    void aux_wait(hipEvent_t e) {
        if (e && !rc && hipStreamWaitEvent(s, e, 0) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "hipStreamWaitEvent failed");
    }

    // ---- data gradients on the forward's contraction kernels ----
    // stride-1 SAME conv (kh x kw odd): dx = conv(dy, flipped / transposed filter)
    // `planes`: dy as fp16x2 planes (bn_bwd wrote them beside the fp32 dy the weight gradient reads) -> conv3h_kernel
    void dgrad_s1(const std::string& name, const float* dy, int H, int W, int Cout, int Cin, float* dx, const void* planes = nullptr,
                  const float* planes_a_inv = nullptr) {
        if (rc) return;
        int Ho, Wo;
        IgemmDesc d = conv_desc(dy, H, W, Cout, Cout, c->p("pkd:" + name + "/weights"), 3, 3, 1, 1, true, Cin, dx, Cin, Ho, Wo);
        auto hs = c->h2d_slot.find(name + "/weights");
        if (planes && planes_a_inv && hs != c->h2d_slot.end()) {
            d.x = nullptr;                          // only the plane-fed tiles may run: the fp32 dy need not exist (bn_bwd)
            d.xp3 = planes;
            d.p3_np = c->B * H * (W + 1);
            d.xp3_fmt = 1;
            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
            d.xp3_bytes = (unsigned)p3h_bytes(c->B, H, W, Cout);
            d.wh2 = c->p("pkdh:" + name + "/weights");
            d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
            d.h2_a_inv = planes_a_inv;
            d.h2_w_inv = c->p("t:h2d") + hs->second;
        }
        layer = "dgrad:" + name;
        contract(d, 1, split_ok);
    }
    // conv with stride (sh, sw) and NO padding before (VALID, or TF SAME with pad_before = 0): dx = conv2d_transpose(dy) cropped to
    // [H, W], as a stride-1 conv over dy whose N index is (ry, rx, ci) with a depth-to-space epilogue
    void dgrad_strided(const std::string& name, const float* dy, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int H, int W,
                       int Cin, float* dx, int ldy) {
        if (rc) return;
        if (c->tbufs.count("pkd:" + name + "/weights#p0")) {       // one contraction per output phase over its non-zero taps
            for (int ph = 0; ph < 4 && !rc; ++ph) {
                const int ry = ph >> 1, rx = ph & 1;
                const int nth = phase_taps(kh, ry), ntw = phase_taps(kw, rx);
                if (nth * ntw == 0) continue;                          // (1x1: the other three phases stay zero - written once at bind)
                IgemmDesc d;
                d.x = dy; d.w = c->p("pkd:" + name + "/weights#p" + std::to_string(ph));
                d.y = dx + ((long)ry * W + rx) * ldy;
                d.Hg = (H - ry + 1) / 2; d.Wg = (W - rx + 1) / 2;
                d.M = c->B * d.Hg * d.Wg; d.N = Cin; d.K = nth * ntw * Cout; d.Kpad = ceil16(d.K);
                d.Hin = Ho; d.Win = Wo; d.Cin = Cout; d.ldx = Cout; d.x_bstride = (long)Ho * Wo * Cout;
                d.ntaps = nth * ntw; d.TW = ntw; d.tap_sh = -1; d.tap_sw = -1; d.log2Cin = ilog2_exact(Cout);
                d.Cout = Cin; d.Hlim = d.Hg; d.Wlim = d.Wg;
                d.ldy = 2 * ldy; d.y_rstride = 2L * W * ldy; d.y_bstride = (long)H * W * ldy;
                layer = "dgrad:" + name + "#p" + std::to_string(ph);
                contract(d, 1, false);
            }
            return;
        }
        const int nth = cdiv(kh, sh), ntw = cdiv(kw, sw);
        IgemmDesc d;
        d.x = dy; d.w = c->p("pkd:" + name + "/weights"); d.y = dx;
        d.Hg = cdiv(H, sh); d.Wg = cdiv(W, sw);
        d.M = c->B * d.Hg * d.Wg; d.N = sh * sw * Cin; d.K = nth * ntw * Cout; d.Kpad = ceil16(d.K);
        d.Hin = Ho; d.Win = Wo; d.Cin = Cout; d.ldx = Cout; d.x_bstride = (long)Ho * Wo * Cout;
        d.ntaps = nth * ntw; d.TW = ntw; d.tap_sh = -1; d.tap_sw = -1; d.log2Cin = ilog2_exact(Cout);
        d.dsh = sh; d.dsw = sw; d.Cout = Cin; d.Hlim = H; d.Wlim = W;
        d.ldy = ldy; d.y_rstride = (long)W * ldy; d.y_bstride = (long)H * W * ldy;
        layer = "dgrad:" + name;
        contract(d, 1, split_ok);
    }
    // fully_connected: dx[M][K] = dy[M][N] . W^T   (dy rows padded with zeros to a multiple of 4)
    void dgrad_fc(const std::string& name, const float* dy, int lddy, int M, int N, int K, float* dx, int lddx) {
        if (rc) return;
        IgemmDesc d;
        d.x = dy; d.w = c->p("pkd:" + name + "/weights"); d.y = dx;
        d.M = M; d.N = K; d.K = ceil4(N); d.Kpad = ceil16(d.K);
        d.Hg = 1; d.Wg = 1; d.Hin = 1; d.Win = 1; d.Cin = d.K; d.ldx = lddy; d.x_bstride = lddy;
        d.ntaps = 1; d.Cout = K; d.Hlim = 1; d.Wlim = 1; d.ldy = lddx; d.y_rstride = lddx; d.y_bstride = lddx;
        layer = "dgrad:" + name;
        gemm(d, 1, split_ok);
    }
    // y = act(x W + b) backward: weights (and, through `dx`, the input).  dy [M][N] with row stride lddy.
    void fc_bwd(const std::string& name, const float* x, int ldx, int M, int K, const float* dy, int lddy, int N, float* dx, int lddx) {
        WgradDesc w = wdesc(x, 1, 1, ldx, K, dy, 1, 1, lddy, N, 1, 1, 1, 1, 0, 0);
        w.B = M;
        wgrad("wgrad:" + name, w, grad(name + "/weights"));
        if (dx) dgrad_fc(name, dy, lddy, M, N, K, dx, lddx);
    }

    // ---- ResNet18 trunk (resnet.py:123-236) backward; gfeat = dL/d(conv5_2 output) [B,7,14,512] ----
    double* bnb_acc(int li) { return reinterpret_cast<double*>(c->p("t:bnbacc" + sfx)) + (size_t)li * 2 * 512; }
    // training-mode BN of layer `li` backward: dz = (ga + gb) * (act > 0) -> dy (and dz), dgamma, dbeta
    // dy plane buffer `which` (0: of conv_2, 1: of conv_1) of block parity `par`, and where its 2^-kd lives
    void* dp_buf(int which, int par) { return c->p(std::string(which ? "t:DPd" : "t:DPc") + (par ? "1" : "0") + sfx); }
    float* dp_a_inv(int which, int par) { return c->p("t:h2d") + (sfx.empty() ? 0 : 4) + 2 * which + par; }
    // (H, W) > 0: dy also as fp16x2 planes in `planes` (scale to `planes_a_inv`) for the stride-1 data gradient / the weight gradient
    // that follow (returns true when written)
    // planes_only: no fp32 dy at all (the weight gradient reads the planes too)
    bool bn_bwd(const std::string& bn_name, int li, const float* ga, const float* gb, const float* act, const float* y, long npix, int C,
                float* dy, float* dz, bool self_mask = false, int H = 0, int W = 0, void* planes = nullptr, float* planes_a_inv = nullptr,
                bool planes_only = false, const unsigned char* relu_bits = nullptr) {
        if (rc) return false;
        if (H > 0 && planes && h2d_on(c) && c->h2d_slot.count(bn_name + "/weights")) {
            const BnRef bn = bn_ref(li, bn_name, npix);
            double* acc = bnb_acc(li);
            layer = "bnbwd:" + bn_name;
            static const bool no_self = getenv("SAGEN_BN_READ_ACT") != nullptr;
            const int sm = self_mask && !no_self;
            const float* a = (sm || relu_bits) ? nullptr : act;     // (the one-bit mask replaces the fp32 activation: 1/32 of its bytes, twice)
            float* mx = c->p("t:mxpart" + sfx);
            int nb = 0;
            timed("bn_bwd_reduce_kernel", 0.0, [&] { return bn_bwd_reduce_launch(ga, gb, a, y, bn, npix, C, acc, c->p(redws), s, sm, mx, &nb, relu_bits); });
            timed("bn_bwd_apply_h2_kernel", 0.0, [&] {
                return bn_bwd_apply_h2_launch(ga, gb, a, y, bn, acc, c->B, H, W, C, planes_only ? nullptr : dy, dz, grad(bn_name + "/bn/gamma"), grad(bn_name + "/bn/beta"), s, sm,
                                              planes, mx, nb, planes_a_inv, reinterpret_cast<unsigned*>(c->p("h2s") + 7), relu_bits); });
            return !rc;
        }
        bn_bwd_plain(bn_name, li, ga, gb, act, y, npix, C, dy, dz, self_mask);
        return false;
    }
    // the operands of a stride-1 3x3 weight gradient also as planes (wgrad3h_kernel): G = retained forward planes, D = dy planes
    WgradDesc with_planes(WgradDesc d, const void* gp, const float* gp_a_inv, const void* dp, const float* dpa) {
        d.gp = gp; d.gp_a_inv = gp_a_inv; d.dp = dp; d.dp_a_inv = dpa;
        return d;
    }
    bool h2w() const { return h2w_on(c) && c->tbufs.count("t:h2a" + sfx) != 0; }
    void bn_bwd_plain(const std::string& bn_name, int li, const float* ga, const float* gb, const float* act, const float* y, long npix, int C,
                float* dy, float* dz, bool self_mask = false) {
        if (rc) return;
        const BnRef bn = bn_ref(li, bn_name, npix);
        double* acc = bnb_acc(li);
        layer = "bnbwd:" + bn_name;
        static const bool no_self = getenv("SAGEN_BN_READ_ACT") != nullptr;     // (A/B: read the retained activation for its sign)
        const int sm = self_mask && !no_self;
        const float* a = sm ? nullptr : act;
        timed("bn_bwd_reduce_kernel", 0.0, [&] { return bn_bwd_reduce_launch(ga, gb, a, y, bn, npix, C, acc, c->p(redws), s, sm); });
        timed("bn_bwd_apply_kernel", 0.0, [&] {
            return bn_bwd_apply_launch(ga, gb, a, y, bn, acc, npix, C, dy, dz, grad(bn_name + "/bn/gamma"), grad(bn_name + "/bn/beta"), s, sm); });
    }

    void resnet_bwd(const std::string& scope, const float* gfeat) {
        const int B = c->B;
        const float* ga = gfeat;
        const float* gb = nullptr;
        float* DA = c->p("t:DA" + sfx);
        hipEvent_t done[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int k = 7; k >= 0 && !rc; --k) {
            aux_wait(done[k + 2]);           // the weight gradients of block k+2 have read the buffers this block overwrites
            const int st = k / 2, unit = k % 2 + 1;
            const int cout = RS_C[st], cin = (unit == 1 && st > 0) ? RS_C[st - 1] : cout;
            const bool first = unit == 1 && st > 0;
            const int Ho = RS_H[st], Wo = RS_W[st], H = first ? 2 * Ho : Ho, W = first ? 2 * Wo : Wo;
            const long npix = (long)B * Ho * Wo;
            const std::string pfx = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(unit);
            const std::string ks = std::to_string(k) + sfx;
            const float* xin = k == 0 ? c->p("t:x0" + sfx) : c->p("t:out:" + std::to_string(k - 1) + sfx);
            const float* y1 = c->p("t:y1:" + ks); const float* a1 = c->p("t:a1:" + ks);
            const float* y2 = c->p("t:y2:" + ks); const float* out = c->p("t:out:" + ks);
            float* A = c->p(std::string(k & 1 ? "t:A1" : "t:A0") + sfx);
            float* Z = c->p(std::string(k & 1 ? "t:Z1" : "t:Z0") + sfx);
            float* DY = c->p(std::string(k & 1 ? "t:DYc1" : "t:DYc0") + sfx);       // dy of conv_2
            float* DY1 = c->p(std::string(k & 1 ? "t:DYd1" : "t:DYd0") + sfx);      // dy of conv_1
            const int li1 = 1 + 2 * k, li2 = 2 + 2 * k;
            // out = relu(bn2(y2) + shortcut)
            const bool planes_on = h2d_on(c) && c->tbufs.count("t:DPc0" + sfx) != 0;
            static const bool no_relu_bits = getenv("SAGEN_TRAIN_NO_RELU_BITS") != nullptr;      // (A/B: conv_2's batch-norm backward reads the fp32 block output for its ReLU mask)
            void* DP2 = planes_on ? dp_buf(0, k & 1) : nullptr;
            void* DP1 = planes_on ? dp_buf(1, k & 1) : nullptr;
            const bool P2 = bn_bwd(pfx + "/conv_2", li2, ga, gb, out, y2, npix, cout, DY, Z, false, Ho, Wo, DP2, planes_on ? dp_a_inv(0, k & 1) : nullptr, h2w() && !wgrad_reads_fp32_operands(),
                                   h2w() && !no_relu_bits ? reinterpret_cast<const unsigned char*>(c->p("t:mb:" + ks)) : nullptr);
            {
                WgradDesc w = wdesc(a1, Ho, Wo, cout, cout, DY, Ho, Wo, cout, cout, 3, 3, 1, 1, -1, -1);
                if (P2 && h2w()) w = with_planes(w, pl_buf("a1", k), pl_a_inv(k), DP2, dp_a_inv(0, k & 1));
                wgrad("wgrad:" + pfx + "/conv_2", w, grad(pfx + "/conv_2/weights"));
            }
            dgrad_s1(pfx + "/conv_2", DY, Ho, Wo, cout, cout, DA, P2 ? DP2 : nullptr, P2 ? dp_a_inv(0, k & 1) : nullptr);
            // a1 = relu(bn1(y1))
            const bool P1 = bn_bwd(pfx + "/conv_1", li1, DA, nullptr, a1, y1, npix, cout, DY1, nullptr, true,       // a1 > 0 <=> bn1(y1) > 0: a1 is not read
                                   first ? 0 : Ho, Wo, DP1, planes_on ? dp_a_inv(1, k & 1) : nullptr, h2w() && !wgrad_reads_fp32_operands());
            if (first) {
                wgrad("wgrad:" + pfx + "/conv_1", wdesc(xin, H, W, cin, cin, DY1, Ho, Wo, cout, cout, 3, 3, 2, 2, 0, 0), grad(pfx + "/conv_1/weights"));
                wgrad("wgrad:" + pfx + "/shortcut", wdesc(xin, H, W, cin, cin, Z, Ho, Wo, cout, cout, 1, 1, 2, 2, 0, 0), grad(pfx + "/shortcut/weights"));
                dgrad_strided(pfx + "/conv_1", DY1, Ho, Wo, cout, 3, 3, 2, 2, H, W, cin, A, cin);
                float* S = c->p("t:S" + std::to_string(st) + sfx);
                dgrad_strided(pfx + "/shortcut", Z, Ho, Wo, cout, 1, 1, 2, 2, H, W, cin, S, cin);
                gb = S;
            } else {
                WgradDesc w = wdesc(xin, H, W, cin, cin, DY1, Ho, Wo, cout, cout, 3, 3, 1, 1, -1, -1);
                if (P1 && h2w()) w = with_planes(w, pl_buf("x", k), pl_a_inv(8 + k), DP1, dp_a_inv(1, k & 1));
                wgrad("wgrad:" + pfx + "/conv_1", w, grad(pfx + "/conv_1/weights"));
                dgrad_s1(pfx + "/conv_1", DY1, H, W, cout, cin, A, P1 ? DP1 : nullptr, P1 ? dp_a_inv(1, k & 1) : nullptr);
                gb = Z;
            }
            done[k] = aux_mark();
            ga = A;
            ms_flush();
        }
        if (rc) return;
        // pool + stem: x0 = maxpool(relu(bn0(y0)))  (resnet.py:133-135)
        const std::string name = scope + "/conv1/conv";
        float* dz0 = c->p("t:dz0" + sfx);
        const BnRef bn0 = bn_ref(0, name, (long)B * 112 * 224);
        layer = "poolbwd:" + scope;
        static const bool unfused = getenv("SAGEN_STEM_BWD_UNFUSED") != nullptr;
        if (unfused) {
            timed("maxpool_bwd_kernel", 0.0, [&] { return maxpool_bwd_launch(c->p("y0" + sfx), bn0, c->p("t:x0" + sfx), ga, gb, dz0, B, 112, 224, 64, s); });
            bn_bwd(name, 0, dz0, nullptr, nullptr, c->p("y0" + sfx), (long)B * 112 * 224, 64, dz0, nullptr);       // (in place: elementwise)
        } else {
            timed("maxpool_bn_bwd_kernels", 0.0, [&] {
                return maxpool_bn_bwd_launch(c->p("y0" + sfx), bn0, c->p("t:x0" + sfx), ga, gb, dz0, B, 112, 224, 64, bnb_acc(0), c->p(redws),
                                             grad(name + "/bn/gamma"), grad(name + "/bn/beta"), s,
                                             c->rawpool_last && scope == "video_encoder" ? c->p("rx0" + sfx) : nullptr); });
        }
        // 7x7/2 over the zero-bordered 4-channel frame: the 7 (+1 zero) horizontal taps x 4 channels are 32 contiguous floats
        WgradDesc w = wdesc(c->p("xpad" + sfx), 229, 454, 4, 32, dz0, 112, 224, 64, 64, 7, 1, 2, 2, 0, 0);
        wgrad("wgrad:" + name, w, c->p("t:stemtmp" + sfx));
        Bwd& u = wg ? *wg : *this;              // (on the stream the stem's weight gradient ran on)
        u.layer = "wgrad:" + name;
        u.timed("stem_wgrad_unpack_kernel", 0.0, [&] { return stem_wgrad_unpack_launch(c->p("t:stemtmp" + sfx), grad(name + "/weights"), u.s); });
        if (u.rc) rc = u.rc;
        ms_flush();
    }

    // ---- everything after the loss ----
    void run() {
        const int B = c->B, nsep = c->nsep, Cb = c->Cb;
        const int ldc = ceil4(3 * (nsep + 1)), ncoef = 3 * (nsep + 1);
        const int nloc = c->cfg.n_loc_units;
        float* dcoeffs = c->p("t:dcoeffs");
        // mask -> iSTFT -> mix adjoint: d(deconv1 output rows 44..66) into the zero-bordered buffer (virtual rows 40..70), d(coeffs)
        layer = "separation/mask-istft-mix:bwd";
        timed("mask_istft_bwd_kernel", 0.0, [&] {
            return mask_istft_mix_bwd_launch(c->p("dmask"), 23L * 1024 * nsep, 1, c->p("spec"), c->p("coeffs"), c->p("t:dpred"), B, nsep,
                                             c->p("t:ddmask"), 31L * 1024 * nsep, -3, dcoeffs, ldc, c->p("t:mbw"), s); });

        // localization FCs (model.py:241-271): coeffs = fc_last(relu(fc..(bott)))
        {
            const std::string last = "localization/fc" + std::to_string(nloc + 1);
            const float* dy = dcoeffs;
            int lddy = ldc, N = ncoef;
            relu_bwd("bias:" + last, dcoeffs, ldc, nullptr, 0, nullptr, 0, nullptr, 0, (long)B * 3, ncoef, last + "/biases");
            for (int i = nloc; i >= 0; --i) {
                const std::string name = "localization/fc" + std::to_string(i + 1);
                const float* x = i == 0 ? c->p("bott") : c->p("loc" + std::to_string(i));
                const int K = i == 0 ? Cb : c->cfg.loc_units[i - 1];
                float* dx = i == 0 ? c->p("t:g:bott_loc") : c->p("t:g:loc" + std::to_string(i));
                fc_bwd(name, x, K, B * 3, K, dy, lddy, N, dx, K);
                if (i > 0) {
                    float* dyn = c->p("t:dy:loc" + std::to_string(i));
                    relu_bwd("relu:localization/fc" + std::to_string(i), dx, K, nullptr, 0, x, K, dyn, K, (long)B * 3, K,
                             "localization/fc" + std::to_string(i) + "/biases");
                    dy = dyn; lddy = K; N = K;
                }
            }
        }

        // separation decoder (model.py:282-311): deconv1 is linear; deconv5..2 are ReLU'd and concatenated with the encoder skips
        {
            const std::string name = "separation/deconv1";
            const float* ddm = c->p("t:ddmask");
            // (only buffer rows 4..26 = mask frames 1..23 carry a gradient: the border rows stay zero)
            relu_bwd("bias:" + name, ddm, nsep, nullptr, 0, nullptr, 0, nullptr, 0, (long)B * 23 * 1024, nsep, name + "/biases", 23L * 1024, 31L * 1024, 4L * 1024);
            // live grid rows 10..16 of cat1 <-> buffer rows 4*i' + p (virtual rows 40..70)
            // (with fewer than 64 tracks, 64 / nsep neighbouring horizontal taps are folded into the channel index: neighbouring
            // pixels of the gradient are contiguous, [tw][track] is the variable's own order, and the 64-row tile is full)
            const int fold = (nsep < 64 && 64 % nsep == 0 && 16 % (64 / nsep) == 0) ? 64 / nsep : 1;
            WgradDesc w = wdesc(ddm, 31, 1024, nsep, nsep * fold, c->p("cat1") + (size_t)10 * 127 * 64, 7, 127, 64, 64, 7, 16 / fold, 4, 8, 0, 0);
            w.tsw = fold;
            w.d_bstride = 31u * 127u * 64u;
            wgrad("wgrad:" + name, w, grad(name + "/weights"));
            int Ho, Wo;
            IgemmDesc d = conv_desc(ddm, 31, 1024, nsep, nsep, c->p("pkd:" + name + "/weights"), 7, 16, 4, 8, false, 64,
                                    c->p("t:dcat1") + (size_t)10 * 127 * 64, 64, Ho, Wo);
            d.y_bstride = 31L * 127 * 64;              // rows outside 10..16 keep the zeros written at bind
            layer = "dgrad:" + name;
            contract(d);
        }
        for (int l = 1; l <= 4 && !rc; ++l) {          // deconv_{l+1}: cat_{l+1} -> decoder half of cat_l
            const std::string name = "separation/deconv" + std::to_string(l + 1);
            const int C = c->enc_c[l], H = c->enc_h[l], W = c->enc_w[l];
            const int Cn = 2 * c->enc_c[l + 1], Hn = c->enc_h[l + 1], Wn = c->enc_w[l + 1];
            const float* dcat = c->p("t:dcat" + std::to_string(l));
            float* dyd = c->p("t:dydec" + std::to_string(l));
            const int kh = AENC_K[l][0], sh = AENC_S[l][0];
            // The forward of this step read rows [lo, hi) of cat_l (model.hip: only those reach the cropped window) from rows [lon, hin) of
            // cat_(l+1): outside them the gradient is exactly zero.  The ReLU pass writes the band of d(deconv output), the two contractions
            // run over the band's rows of cat_(l+1), and the rows they never write keep the zeros of sagen_train_bind.
            const int lo = c->dec_lo[l], hi = c->dec_hi[l], lon = c->dec_lo[l + 1], hin = c->dec_hi[l + 1];
            const bool band = (lo > 0 || hi < H || lon > 0 || hin < Hn) && hi > lo && hin > lon;
            if (!band) {
                relu_bwd("relu:" + name, dcat, 2 * C, nullptr, 0, c->p("cat" + std::to_string(l)), 2 * C, dyd, C, (long)B * H * W, C, name + "/biases");
                wgrad("wgrad:" + name, wdesc(dyd, H, W, C, C, c->p("cat" + std::to_string(l + 1)), Hn, Wn, Cn, Cn, AENC_K[l][0], AENC_K[l][1],
                                             AENC_S[l][0], AENC_S[l][1], 0, 0), grad(name + "/weights"));
                int Ho, Wo;
                IgemmDesc d = conv_desc(dyd, H, W, C, C, c->p("pkd:" + name + "/weights"), AENC_K[l][0], AENC_K[l][1], AENC_S[l][0], AENC_S[l][1],
                                        false, Cn, c->p("t:dcat" + std::to_string(l + 1)), Cn, Ho, Wo);
                if (Ho != Hn || Wo != Wn) { rc = fail(SAGEN_ERR_SHAPE, "deconv%d backward geometry", l + 1); return; }
                layer = "dgrad:" + name;
                contract(d);
                continue;
            }
            const int Hb = hin - lon;                         // rows of cat_(l+1) in the band
            const int g0 = lon * sh;                          // ... whose taps start at this row of d(deconv output)
            const int Hgb = (Hb - 1) * sh + kh;               // ... and span this many rows (all inside the tensor: the encoder's conv was VALID)
            if (g0 > lo || g0 + Hgb < hi || g0 + Hgb > H) { rc = fail(SAGEN_ERR_SHAPE, "deconv%d backward band", l + 1); return; }
            relu_bwd("relu:" + name, dcat, 2 * C, nullptr, 0, c->p("cat" + std::to_string(l)), 2 * C, dyd, C, (long)B * (hi - lo) * W, C, name + "/biases",
                     (long)(hi - lo) * W, (long)H * W, (long)lo * W);
            const float* gband = dyd + (size_t)g0 * W * C;
            const float* xband = c->p("cat" + std::to_string(l + 1)) + (size_t)lon * Wn * Cn;
            WgradDesc w = wdesc(gband, Hgb, W, C, C, xband, Hb, Wn, Cn, Cn, AENC_K[l][0], AENC_K[l][1], AENC_S[l][0], AENC_S[l][1], 0, 0);
            w.g_bstride = (unsigned)((long)H * W * C);
            w.d_bstride = (unsigned)((long)Hn * Wn * Cn);
            wgrad("wgrad:" + name, w, grad(name + "/weights"));
            int Ho, Wo;
            IgemmDesc d = conv_desc(gband, Hgb, W, C, C, c->p("pkd:" + name + "/weights"), AENC_K[l][0], AENC_K[l][1], AENC_S[l][0], AENC_S[l][1],
                                    false, Cn, c->p("t:dcat" + std::to_string(l + 1)) + (size_t)lon * Wn * Cn, Cn, Ho, Wo);
            if (Ho != Hb || Wo != Wn) { rc = fail(SAGEN_ERR_SHAPE, "deconv%d backward geometry (band)", l + 1); return; }
            d.x_bstride = (long)H * W * C;
            d.y_bstride = (long)Hn * Wn * Cn;
            layer = "dgrad:" + name;
            contract(d);
        }
        // fc-feats (model.py:287-294): tiled over the 6 frequency columns of cat5
        {
            float* g = c->p("t:g:fcfeats");
            float* dy = c->p("t:dy:fcfeats");
            layer = "tile:separation/fc-feats";
            timed("sum_rows_kernel", 0.0, [&] { return sum_rows_launch(c->p("t:dcat5") + 512, 1024, nullptr, 0, 6, (long)B * 3, 512, g, 512, s); });
            relu_bwd("relu:separation/fc-feats", g, 512, nullptr, 0, c->p("cat5") + 512, 6 * 1024, dy, 512, (long)B * 3, 512, "separation/fc-feats/biases");
            fc_bwd("separation/fc-feats", c->p("bott"), Cb, B * 3, Cb, dy, 512, 512, c->p("t:g:bott_sep"), Cb);
        }
        const float* gbl = c->p("t:g:bott_loc");
        const float* gbs = c->p("t:g:bott_sep");
        // bottleneck (model.py:203-239), audio part: audio-fc over (w, c) of conv5
        {
            float* dy = c->p("t:dy:audiofc");
            relu_bwd("relu:bottleneck/audio-fc", gbl, Cb, gbs, Cb, c->p("bott"), Cb, dy, 1024, (long)B * 3, 1024, "bottleneck/audio-fc/biases");
            WgradDesc w = wdesc(c->p("cat5"), 3, 6, 1024, 512, dy, 3, 1, 1024, 1024, 1, 6, 1, 1, 0, 0);
            wgrad("wgrad:bottleneck/audio-fc", w, grad("bottleneck/audio-fc/weights"));
            dgrad_fc("bottleneck/audio-fc", dy, 1024, B * 3, 1024, 3072, c->p("t:g:conv5_fc"), 3072);
        }
        // audio encoder (model.py:161-187), conv5 .. conv1.  Nothing on the way to the visual trunk depends on it: with a second
        // stream the whole chain (ReLU / bias passes, weight gradients, strided data gradients - small launches that fill a fraction of
        // the chip) runs there, under the trunk's backward on the main stream.
        {
            Bwd& a = wg ? *wg : *this;
            if (wg) {
                hipEvent_t e = next_event();
                if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(wg->s, e, 0) != hipSuccess) { rc = fail(SAGEN_ERR_HIP, "stream fork failed"); return; }
            }
            for (int l = 5; l >= 1 && !rc && !a.rc; --l) {
                const std::string name = "audio_encoder/conv" + std::to_string(l);
                const int C = c->enc_c[l], H = c->enc_h[l], W = c->enc_w[l];
                const int off = l == 5 ? 0 : C;
                const float* ga = c->p("t:dcat" + std::to_string(l)) + off;
                const float* gb2 = l == 5 ? c->p("t:g:conv5_fc") : c->p("t:g:conv" + std::to_string(l));
                float* dy = c->p("t:dy:conv" + std::to_string(l));
                a.relu_bwd("relu:" + name, ga, 2 * C, gb2, C, c->p("cat" + std::to_string(l)) + off, 2 * C, dy, C, (long)B * H * W, C, name + "/biases");
                const int kh = AENC_K[l - 1][0], kw = AENC_K[l - 1][1], sh = AENC_S[l - 1][0], sw = AENC_S[l - 1][1];
                if (l == 1) {
                    // Cin = 1: the 16 taps along frequency are 16 contiguous floats of the magnitude row (pixel stride 1 float)
                    WgradDesc w = wdesc(c->p("mag"), 127, 1024, 1, kw, dy, H, W, C, C, kh, 1, sh, sw, 0, 0);
                    a.wgrad_here("wgrad:" + name, w, grad(name + "/weights"));
                } else {
                    const int Cp = c->enc_c[l - 1], Hp = c->enc_h[l - 1], Wp = c->enc_w[l - 1];
                    const float* x = c->p("cat" + std::to_string(l - 1)) + Cp;            // encoder half of cat_{l-1}
                    a.wgrad_here("wgrad:" + name, wdesc(x, Hp, Wp, 2 * Cp, Cp, dy, H, W, C, C, kh, kw, sh, sw, 0, 0), grad(name + "/weights"));
                    a.dgrad_strided(name, dy, H, W, C, kh, kw, sh, sw, Hp, Wp, Cp, c->p("t:g:conv" + std::to_string(l - 1)), Cp);
                }
            }
            if (a.rc) rc = a.rc;
        }
        ms_flush();               // (localisation, mask decoder, audio bottleneck and encoder: everything but the visual branches)
        // visual encoders: bottleneck FCs (video-fc tiled over the 3 steps), then the trunk
        int choff = 1024;
        for (int e = 0; e < 2 && !rc; ++e) {
            const bool on = e == 0 ? c->has_video : c->has_flow;
            if (!on) continue;
            const std::string enc = e == 0 ? "video" : "flow";
            sfx = (e == 1 && c->has_video) ? "_b" : "";
            float* g = c->p("t:g:vfc" + sfx);
            float* dy = c->p("t:dy:vfc" + sfx);
            layer = "tile:bottleneck/" + enc + "-fc";
            timed("sum_rows_kernel", 0.0, [&] { return sum_rows_launch(gbl + choff, Cb, gbs + choff, Cb, 3, (long)B, 512, g, 512, s); });
            relu_bwd("relu:bottleneck/" + enc + "-fc", g, 512, nullptr, 0, c->p("bott") + choff, 3 * Cb, dy, 512, (long)B, 512,
                     "bottleneck/" + enc + "-fc/biases");
            fc_bwd("bottleneck/" + enc + "-fc", c->p("fcred" + sfx), 98 * 128, B, 98 * 128, dy, 512, 512, c->p("t:g:fcred" + sfx), 98 * 128);
            float* dyr = c->p("t:dy:fcred" + sfx);
            relu_bwd("relu:bottleneck/" + enc + "-fc-red", c->p("t:g:fcred" + sfx), 128, nullptr, 0, c->p("fcred" + sfx), 128, dyr, 128,
                     (long)B * 98, 128, "bottleneck/" + enc + "-fc-red/biases");
            fc_bwd("bottleneck/" + enc + "-fc-red", c->p("t:out:7" + sfx), 512, B * 98, 512, dyr, 128, 128, c->p("t:g:feat" + sfx), 512);
            ms_flush();
            resnet_bwd(enc + "_encoder", c->p("t:g:feat" + sfx));
            choff += 512;
        }
    }
};

}  // namespace sagen

size_t sagen_train_workspace_bytes_impl(sagen_ctx* c) {
    train_carve(c);
    return c->tws_floats * sizeof(float);
}

int sagen_train_bind_impl(sagen_ctx* c, const sagen_tensor* grads, int n_grads, const sagen_tensor* moving, int n_moving, void* tws,
                          size_t tws_bytes, hipStream_t s) {
    if (!c || !grads || !tws) return fail(SAGEN_ERR_NULL, "sagen_train_bind: null argument");
    if (!c->bound) return fail(SAGEN_ERR_WEIGHTS, "sagen_train_bind: bind the weights first");
    if (c->G != 1) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_train_bind: a grouped context (sagen_create_grouped) runs inference only");
    if (!c->freq_mask) return fail(SAGEN_ERR_UNSUPPORTED, "the training step implements separation 'unet_mask' (the configuration train.py trains)");
    train_carve(c);
    if (tws_bytes < c->tws_floats * sizeof(float))
        return fail(SAGEN_ERR_WORKSPACE, "train workspace has %zu bytes, need %zu", tws_bytes, c->tws_floats * sizeof(float));
    if (((uintptr_t)tws) % 256) return fail(SAGEN_ERR_WORKSPACE, "train workspace must be 256-byte aligned");
    c->tws = (float*)tws;
    c->pack_jobs_bwd.clear();
    c->h2d_jobs.clear();
    std::vector<float*> gp(c->vars.size(), nullptr), mp(c->vars.size(), nullptr);
    for (int pass = 0; pass < 2; ++pass) {
        const sagen_tensor* ts = pass ? moving : grads;
        const int n = pass ? n_moving : n_grads;
        for (int i = 0; i < n && ts; ++i) {
            const sagen_tensor& t = ts[i];
            if (!t.name || !t.data) return fail(SAGEN_ERR_NULL, "tensor %d has a null name or data pointer", i);
            auto it = c->var_index.find(t.name);
            if (it == c->var_index.end()) continue;
            const VarSpec& vs = c->vars[it->second];
            bool ok = t.ndim == vs.ndim;
            for (int k = 0; ok && k < vs.ndim; ++k) ok = t.shape[k] == vs.shape[k];
            if (!ok) return fail(SAGEN_ERR_WEIGHTS, "gradient / moving-average tensor %s has the wrong shape", t.name);
            if (((uintptr_t)t.data) % 16) return fail(SAGEN_ERR_WEIGHTS, "tensor %s is not 16-byte aligned", t.name);
            (pass ? mp : gp)[it->second] = const_cast<float*>(t.data);
        }
    }
    for (size_t i = 0; i < c->vars.size(); ++i) {
        const std::string& nm = c->vars[i].name;
        if (nm.find("/moving_") != std::string::npos) continue;
        if (!gp[i]) return fail(SAGEN_ERR_WEIGHTS, "no gradient buffer for variable %s", nm.c_str());
    }
    c->grad_ptr = gp;
    c->mov_ptr = mp;
    // buffers whose untouched parts must be zero: the border rows of d(mask), the pad column of d(coeffs), the dead rows of d(cat1)
    // ... and of d(cat2..5) / d(deconv5..2 outputs) (the decoder's bands)
    for (const char* nm : {"t:ddmask", "t:dcoeffs", "t:dcat1", "t:dcat2", "t:dcat3", "t:dcat4", "t:dcat5", "t:dydec1", "t:dydec2", "t:dydec3", "t:dydec4"})
        if (c->tbufs.count(nm)) SAGEN_HIP_CHECK(hipMemsetAsync(c->p(nm), 0, c->tbufs.at(nm).n * sizeof(float), s));
    // ... and the three output phases a 1x1 stride-2 shortcut never reaches (its phase-wise data gradient only writes phase (0, 0))
    for (const char* nm : {"t:S1", "t:S2", "t:S3", "t:S1_b", "t:S2_b", "t:S3_b"})
        if (c->tbufs.count(nm)) SAGEN_HIP_CHECK(hipMemsetAsync(c->p(nm), 0, c->tbufs.at(nm).n * sizeof(float), s));
    c->train_ready = true;
    return SAGEN_OK;
}

int sagen_train_step_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, const float* target,
                          const float* mask, float* pred_out, double* loss_out, int update_moving, hipStream_t s);
// uint8 frames (sagen_train_step_u8): the pad pass normalises them for the stem's weight gradient, the stem's forward reads u - 128
int sagen_train_step_u8_impl(sagen_ctx* c, const float* audio, const uint8_t* video_u8, const float* flow, const float* target,
                             const float* mask, float* pred_out, double* loss_out, int update_moving, hipStream_t s) {
    if (!c) return fail(SAGEN_ERR_NULL, "sagen_train_step_u8: null ctx");
    if (!c->has_video) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_train_step_u8: the video encoder is not enabled");
    c->video_u8 = true;
    const int rc = sagen_train_step_impl(c, audio, reinterpret_cast<const float*>(video_u8), flow, target, mask, pred_out, loss_out, update_moving, s);
    c->video_u8 = false;
    return rc;
}
int sagen_train_step_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, const float* target,
                          const float* mask, float* pred_out, double* loss_out, int update_moving, hipStream_t s) {
    if (!c || !audio || !target) return fail(SAGEN_ERR_NULL, "sagen_train_step: null argument");
    if (!c->train_ready) return fail(SAGEN_ERR_WORKSPACE, "sagen_train_step: call sagen_train_bind first");
    // the optimiser updated the variables in place: re-pack the filters.  With two streams only the stems' packs run here; the rest
    // (123 MB of variables -> 430 MB of packs, ~190 us) runs on the second stream behind the STFT, under pad + stem + pool of this one
    static const bool no_split_repack = getenv("SAGEN_NO_SPLIT_REPACK") != nullptr;
    const bool split_repack = !no_split_repack && c->ev_pack && c->ev_fork && !c->profiling && sagen_forward_forks(c) && c->train_ready;
    int rc = split_repack ? sagen_repack_part(c, s, 1) : sagen_repack_impl(c, s);
    c->late_repack = split_repack && !rc;
    // the data-gradient packs are first needed after the forward: on the second stream, under it
    static const bool one_stream_pack = getenv("SAGEN_BWD_ONE_STREAM") != nullptr;
    const bool pack_aux = c->aux && c->ev_fork && !one_stream_pack && !c->tuning && !c->profiling;
    // ... enqueued BEHIND the forward's own work on that stream: the main stream's first matrix launch waits for the STFT of the
    // second stream, and with the packs in front of it the stem started 0.2 ms late
    static const bool packs_first = getenv("SAGEN_DGRAD_PACKS_FIRST") != nullptr;      // (A/B: the round-3 order)
    if (!rc && pack_aux && packs_first) {
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_fork, s));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        rc = repack_dgrad(c, c->aux);
    } else if (!rc && !pack_aux) {
        rc = repack_dgrad(c, s);
    } else if (!rc) {
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_fork, s));     // the optimiser's update of the variables (a forward that does not fork never re-records it)
    }
    if (rc) return rc;
    float* pred = pred_out ? pred_out : c->p("t:pred");
    c->train_mode = true;
    rc = sagen_forward_impl(c, audio, video, flow, pred, s);
    c->train_mode = false;
    if (rc) return rc;
    if (pack_aux) {                                    // (the forward joins the second stream only when it forked)
        if (!packs_first) {
            SAGEN_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));      // (a forked forward has long passed a later record of it)
            rc = repack_dgrad(c, c->aux);
            if (rc) return rc;
        }
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_join, c->aux));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    double* loss = reinterpret_cast<double*>(c->p("t:loss"));
    Bwd b(c, s);
    b.layer = "loss";
    b.timed("stft_loss_grad_kernel", 0.0, [&] { return stft_loss_grad_launch(pred, target, mask, c->B, c->p("t:dpred"), loss, s); });
    if (b.rc) return b.rc;
    if (loss_out) SAGEN_HIP_CHECK(hipMemcpyAsync(loss_out, loss, sizeof(double), hipMemcpyDeviceToDevice, s));
    static const bool one_stream = getenv("SAGEN_BWD_ONE_STREAM") != nullptr;
    Bwd w(c, c->aux);
    w.redws = "t:redws2"; w.cslot = CACC_SLOTS / 2; w.wsname = "splitk_aux";      // its own scratch: it runs concurrently with `b`
    // (with the per-launch profiler on, everything stays on one stream: the events then time unoverlapped launches)
    if (c->aux && !one_stream && !c->tuning && !c->profiling) b.wg = &w;
    if (!c->ms_bucket.empty()) {
        c->ms_left = c->ms_count;
        std::fill(c->ms_done.begin(), c->ms_done.end(), 0);
        c->ms_touched.clear();
    }
    b.run();
    b.ms_flush();
    if (b.wg) b.aux_wait(b.aux_mark());          // join: the gradients are complete when the caller's stream gets here
    if (b.rc) return b.rc;
    // (a bucket none of whose milestones fired - a variable this build's backward never asked for - completes here, never early)
    for (size_t k = 0; k < c->ms_event.size(); ++k)
        if (c->ms_left[k] > 0) SAGEN_HIP_CHECK(hipEventRecord(c->ms_event[k], s));
    if (update_moving) {
        // contrib batch_norm update ops (core.py:210, decay 0.99; run with the train op through UPDATE_OPS, train.py:147-148)
        std::vector<BnMovingJob> jobs;
        for (int e = 0; e < 2; ++e) {
            if (!(e == 0 ? c->has_video : c->has_flow)) continue;
            const std::string scope = e == 0 ? "video_encoder" : "flow_encoder";
            b.sfx = (e == 1 && c->has_video) ? "_b" : "";
            for (int li = 0; li < 17; ++li) {
                std::string name;
                int C;
                long count;
                if (li == 0) { name = scope + "/conv1/conv"; C = 64; count = (long)c->B * 112 * 224; }
                else {
                    const int k = (li - 1) / 2, st = k / 2;
                    name = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(k % 2 + 1) + ((li - 1) % 2 ? "/conv_2" : "/conv_1");
                    C = RS_C[st]; count = (long)c->B * RS_H[st] * RS_W[st];
                }
                float* mm = c->mov_ptr[c->var_index.at(name + "/bn/moving_mean")];
                float* mv = c->mov_ptr[c->var_index.at(name + "/bn/moving_variance")];
                if (!mm || !mv) continue;
                const BnRef r = b.bn_ref(li, name, count);
                BnMovingJob j;
                j.acc = r.acc; j.inv_count = r.inv_count; j.mm = mm; j.mv = mv; j.C = C;
                jobs.push_back(j);
            }
        }
        rc = bn_moving_update_multi_launch(jobs.data(), (int)jobs.size(), 0.99f, s);
        if (rc) return rc;
    }
    return SAGEN_OK;
}

int sagen_train_set_grad_events_impl(sagen_ctx* c, const char* const* names, const int32_t* bucket, int n, void* const* events, int n_buckets) {
    if (!c->train_ready) return fail(SAGEN_ERR_WORKSPACE, "sagen_train_set_grad_events: call sagen_train_bind first");
    c->ms_bucket.clear(); c->ms_count.clear(); c->ms_left.clear(); c->ms_event.clear(); c->ms_done.clear(); c->ms_touched.clear();
    if (n == 0 || n_buckets == 0) return SAGEN_OK;             // (off)
    if (!names || !bucket || !events) return fail(SAGEN_ERR_NULL, "sagen_train_set_grad_events: null argument");
    std::vector<int> vb(c->vars.size(), -1), cnt(n_buckets, 0);
    for (int i = 0; i < n; ++i) {
        auto it = c->var_index.find(names[i]);
        if (it == c->var_index.end()) return fail(SAGEN_ERR_WEIGHTS, "sagen_train_set_grad_events: unknown variable %s", names[i]);
        if (bucket[i] < 0 || bucket[i] >= n_buckets) return fail(SAGEN_ERR_SHAPE, "sagen_train_set_grad_events: bucket %d of %s out of range", bucket[i], names[i]);
        if (!c->grad_ptr[it->second]) return fail(SAGEN_ERR_WEIGHTS, "sagen_train_set_grad_events: %s has no gradient bound", names[i]);
        if (vb[it->second] < 0) ++cnt[bucket[i]];
        vb[it->second] = bucket[i];
    }
    for (int k = 0; k < n_buckets; ++k)
        if (!events[k]) return fail(SAGEN_ERR_NULL, "sagen_train_set_grad_events: event %d is null", k);
    if (!c->ms_tmp) SAGEN_HIP_CHECK(hipEventCreateWithFlags(&c->ms_tmp, hipEventDisableTiming));
    c->ms_bucket = vb; c->ms_count = cnt; c->ms_left = cnt;
    c->ms_done.assign(c->vars.size(), 0);
    for (int k = 0; k < n_buckets; ++k) c->ms_event.push_back((hipEvent_t)events[k]);
    return SAGEN_OK;
}

int sagen_train_get_buffer_impl(const sagen_ctx* c, const char* name, const float** data, size_t* n) {
    if (!c->tws) return fail(SAGEN_ERR_WORKSPACE, "no train workspace bound");
    auto it = c->tbufs.find(name);
    if (it == c->tbufs.end()) {
        auto jt = c->bufs.find(name);              // buffers of the forward's own workspace ("fcred", "y0", "bott", ...)
        if (jt == c->bufs.end()) return fail(SAGEN_ERR_SHAPE, "unknown train buffer %s", name);
        *data = c->ws + jt->second.off;
        *n = jt->second.n;
        return SAGEN_OK;
    }
    *data = c->tws + it->second.off;
    *n = it->second.n;
    return SAGEN_OK;
}

// Times every (tile, split-K) candidate of every contraction of the step - the forward's and the data gradients' - on these
// inputs and stores the plan (same mechanism as sagen_autotune).  The gradients this call leaves behind are not meaningful
// (candidates overwrite each other's outputs; the L2-cooling memset reuses the mask buffer).
int sagen_train_autotune_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, const float* target,
                              const float* mask, hipStream_t s) {
    if (!c) return fail(SAGEN_ERR_NULL, "sagen_train_autotune: null ctx");
    if (getenv("SAGEN_NO_AUTOTUNE")) return SAGEN_OK;
    if (!c->tune_e0) {
        SAGEN_HIP_CHECK(hipEventCreate(&c->tune_e0));
        SAGEN_HIP_CHECK(hipEventCreate(&c->tune_e1));
    }
    c->tuning = true;
    const int rc = sagen_train_step_impl(c, audio, video, flow, target, mask, nullptr, nullptr, 0, s);
    c->tuning = false;
    if (rc) return rc;
    SAGEN_HIP_CHECK(hipStreamSynchronize(s));
    return SAGEN_OK;
}
