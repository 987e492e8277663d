// Native runtime of the path, shared by model.hip (context, bind, forward) and train_model.hip (training step):
// the context (variable inventory, workspace map, launch plan) and the launch helper `Fwd`.
#pragma once
#include "kernels.h"
#include "fcm.h"
#include <algorithm>
#include <cstdlib>
#include <functional>
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

constexpr int SAGEN_MAX_GROUPS = 32;      // batches per grouped launch (sagen_create_grouped)

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
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_stft = nullptr, ev_pack = nullptr;
    int pack_early_jobs = 0, pack_early_blocks = 0;     // the stems' jobs come first in the pack table: the training step packs them on the caller's stream, the rest on the second stream under the stem (late_repack)
    bool late_repack = false;
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

    // grouped launch (common.h: GroupInfo; sagen_create_grouped): G independent batches per forward.  The workspace is
    // [shared: packed filters, job tables][per-batch region of group 0][... of group 1] ...: `grp_off` floats of shared buffers, then
    // G copies of `grp_floats` floats; bufs / p() describe group 0, group g's copy of a per-batch buffer lies g * grp_floats further on
    int G = 1;
    int tune_groups = 0;                   // groups a tuning pass launches per candidate (SAGEN_TUNE_GROUPS; 0 = all, the default: same box, 15 per call: tuned on all 3 153 - 3 160 ambisonic-s/s in a 14 s process, on 4 groups 3 122 - 3 136 in 10 s, on 2 3 063 - 3 073)
    int inter_group = 0;                   // sagen_set_option("intermediate_group"): the group sagen_get_intermediate reads
    size_t grp_off = 0, grp_floats = 0;
    GroupInfo group_info() const {
        GroupInfo gi;
        if (G > 1 && ws) {
            gi.lo = reinterpret_cast<const char*>(ws + grp_off);
            gi.span = (unsigned long long)grp_floats * sizeof(float);
            gi.stride = gi.span;
            gi.G = G;
        }
        return gi;
    }
    // workspace
    size_t ws_floats = 0;
    float* ws = nullptr;
    std::map<std::string, Buf> bufs;
    std::map<std::string, Named> named;
    // training (train_model.hip): a second caller-provided workspace holds what the backward pass needs - retained activations,
    // gradient activations, data-gradient filter packs, fp64 accumulators.  `train_mode` makes the forward retain.
    // batched filter packs: job tables (host copies; the device copies live in the workspaces)
    std::vector<PackJob> pack_jobs, pack_jobs_bwd;
    int pack_blocks = 0, pack_blocks_bwd = 0;
    bool train_mode = false;
    bool train_ready = false;
    bool stem_fused = true;                // inference: 7x7/2 stem + max-pool as one kernel (stempool.hip); SAGEN_NO_STEMPOOL=1 disables
    bool materialize_mask = false;         // sagen_set_option("materialize_mask"): keep deconv1 -> mask as two kernels so that the logits exist
    bool mask_fused_last = false;          // the last forward ran the fused decoder tail: "separation/deconv1" holds no logits
    bool video_u8 = false;                 // this call's video frames are uint8 (sagen_forward_u8): normalisation fused into the pad pass
    bool train_h2 = true;                  // the training step's forward also runs the trunk's stride-1 3x3 convs on the fp16x2 planes (SAGEN_TRAIN_NO_H2=1: bf16x3)
    bool use_h2 = true;                    // inference: the planes of the trunk are two fp16 planes (conv3h.hip: three products per multiply) instead of three bf16 planes; SAGEN_NO_H2=1 / sagen_set_option("fp16x2", 0)
    bool train_h2d = true;                 // ... and its backward runs the stride-1 3x3 data gradients on fp16x2 planes of dy, written by the batch-norm backward (SAGEN_TRAIN_NO_H2D=1: bf16x3 on fp32 dy)
    bool no_aud_planes = false;            // SAGEN_NO_AUDIO_PLANES=1: conv2 .. conv5 of the audio encoder stay on the register-staged kernels (fp32 operand, in-loop split) instead of conv3g_kernel on fp16x2 planes of cat_l's encoder half (round 6)
    bool no_dh_split = false;              // SAGEN_NO_DH_SPLIT=1: the tuner does not consider conv3h_kernel's dh-split (one filter row per workgroup, round 6)
    bool sk_fused = false;                 // SAGEN_SK_FUSED=1: split-K partials are combined inside the contraction (last-arriver, igemm_epilogue) instead of by a reducer launch - bit-identical, measured no faster
    bool train_h2w = true;                 // ... and the weight gradients of those layers run on the planes too (wgrad3h.hip): the forward retains its activation planes (SAGEN_TRAIN_NO_H2W=1: bf16x3 on the fp32 tensors)
    std::map<std::string, int> h2d_slot;   // per data-gradient filter: index of its 2^-kw in the "t:h2d" table
    std::vector<H2Job> h2d_jobs;
    int h2d_blocks = 0;
    std::map<std::string, int> h2_slot;    // per layer: index of its 2^-kw in the "h2s" table
    std::vector<H2Job> h2_jobs;            // the batched fp16x2 filter pack (host copy of the job table)
    int h2_blocks = 0;
    bool use_p3g = true;                   // the block merges of stages 3, 4 also write planes for the NEXT stage's stride-2 conv_1 + shortcut (conv3g.hip); opt-in: SAGEN_P3G=1 / sagen_set_option("plane_gather", 1) - measured no faster than igemm3_kernel
    bool rawpool_last = false;             // the last training forward kept the pooled raw stem output in "rx0" (stem8pool_kernel<raw+pool>)
    bool train_rawpool = true;             // the training forward's stem writes the raw output AND its pooled extremum (SAGEN_TRAIN_NO_RAWPOOL=1: a pool pass of its own)
    bool train_bands = true;               // the training step's decoder on the live rows only, forward and backward (SAGEN_TRAIN_NO_BANDS=1: full tensors)
    int dec_lo[7] = {0, 0, 0, 0, 0, 0, 0}, dec_hi[7] = {0, 0, 0, 0, 0, 0, 0};      // rows of cat_l the last forward's decoder read (the backward of the same step follows them)
    bool no_scatter = false, no_d1_planes = false, no_lean_trunk = false;      // SAGEN_NO_DECONV_SCATTER / SAGEN_NO_DECONV1_PLANES / SAGEN_NO_LEAN_TRUNK, read when the context is created
    int dec_planes_min_batch = 16;         // the scatter-form decoder contracts fp16x2 planes from this batch size on (sagen_set_option("decoder_planes", 1 / 0): always / never)
    bool use_fcm = false;                  // inference: the skinny FC layers (bottleneck / localisation / fc-feats) run fcm_kernel (fcm.hip) chained through partials; SAGEN_NO_FCM=1: the round-4 contraction + reducer launches
    std::map<std::string, int> fcm_slices; // per FC layer: K slices of its fcm launch
    bool stem8h = true;                    // uint8 frames: one fp16 plane x two fp16 filter planes (stem8.hip, MODE 2); sagen_set_option("u8_stem_h2", 0) / SAGEN_NO_STEM8_H2=1: the bf16 plane x three bf16 filter planes of round 4
    bool stem16 = true;                    // float frames run the fp16x2 variant of that kernel (stem8.hip, F16: two planes of the frame scaled by its exact maximum); sagen_set_option("f16_fast_stem", 0) / SAGEN_NO_STEM16=1: igemm3s2_kernel + the pool pass
    bool stem8 = true;                     // uint8 frames run the one-operand-plane stem (stem8.hip); sagen_set_option("u8_fast_stem", 0) / SAGEN_NO_STEM8=1: the general kernels
    size_t tws_floats = 0;
    float* tws = nullptr;
    std::map<std::string, Buf> tbufs;
    std::vector<float*> grad_ptr;          // per variable: where its gradient is written (caller's gradient buckets)
    // gradient-bucket milestones (sagen_train_set_grad_events): event b is recorded as soon as the last kernel writing into bucket b
    // has been enqueued (after both streams) - the host starts that bucket's all-reduce under the rest of the backward pass
    std::vector<int> ms_bucket;            // per variable: its bucket, -1 = none
    std::vector<int> ms_count, ms_left;    // per bucket: variables in it / not yet written in this step
    std::vector<hipEvent_t> ms_event;
    std::vector<char> ms_done;             // per variable, this step
    std::vector<int> ms_touched;           // variables whose producer was enqueued since the last flush
    hipEvent_t ms_tmp = nullptr;
    std::vector<float*> mov_ptr;           // per variable: writable pointer of a BN moving average (or null)
    Buf talloc(const std::string& name, size_t n) {
        Buf b;
        b.off = tws_floats;
        b.n = n;
        tws_floats += (n + 63) / 64 * 64;
        tbufs[name] = b;
        return b;
    }
    size_t cap(const std::string& name) const { auto it = bufs.find(name); return it != bufs.end() ? it->second.n : tbufs.at(name).n; }

    Buf alloc(const std::string& name, size_t n) {
        Buf b;
        b.off = ws_floats;
        b.n = n;
        ws_floats += (n + 63) / 64 * 64;     // 256-byte granules
        bufs[name] = b;
        return b;
    }
    float* p(const std::string& name) const {
        auto it = bufs.find(name);
        if (it != bufs.end()) return ws + it->second.off;
        return tws + tbufs.at(name).off;
    }
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

static inline void add_resnet_vars(sagen_ctx* c, const std::string& scope) {
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

// can conv3g_kernel run this conv given planes of its Cin-channel input?  (igemm_launch fills the fields conv3g_ok reads only later)
static inline bool conv3g_ok_desc(const IgemmDesc& d, int cin) {
    return cin % 16 == 0 && d.Kpad == d.K && d.in_scale == nullptr && d.bn_in.acc == nullptr && d.dsh * d.dsw == 1;
}

static inline size_t packed_floats(long N, long K) { return (size_t)N * ((K + 15) / 16 * 16); }

// choose a split-K factor for low-parallelism contractions (>= ~2 workgroups per CU, >= 8 K tiles per split)
static inline int auto_splitk(const IgemmDesc& d, IgemmTile tile) {
    const int bm = (tile == TILE_32x128) ? 32 : 64, bn = (tile == TILE_32x128) ? 128 : 64;
    const long blocks = (long)cdiv(d.M, bm) * cdiv(d.N, bn);
    const int nk = d.Kpad / 16;
    if (blocks >= 384 || nk < 16) return 1;
    int sk = (int)std::min<long>({(512 + blocks - 1) / blocks, (long)nk / 8, 64L});
    return std::max(sk, 1);
}

}  // namespace sagen


int sagen_repack_impl(sagen_ctx* c, hipStream_t s);
int sagen_repack_part(sagen_ctx* c, hipStream_t s, int part);     // 0: all; 1: the stems' packs; 2: the rest + fp16x2 planes (model.hip)
bool sagen_forward_forks(const sagen_ctx* c);
int sagen_upload_pack_jobs(std::vector<PackJob>& jobs, void* dev, hipStream_t s);
int sagen_forward_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s);

namespace sagen {

struct Fwd {
    sagen_ctx* c;
    hipStream_t s;
    int rc = SAGEN_OK;
    std::string layer;      // label of the layer being launched (profiling only)
    std::string wsname = "splitk";   // split-K scratch of this launch stream
    std::string sfx;                 // suffix of the trunk buffers this stream owns ("" or "_b")
    hipEvent_t wait_before_mfma = nullptr;   // event the first contraction of this stream has to wait for (see forward)
    hipEvent_t wait_packs = nullptr;         // training step: the filter packs of everything behind the stem (second stream, sagen_forward_impl)
    // set around one contract(): the partials [sk][M][N] are consumed by this pass instead of splitk_reduce_kernel (the scatter-form
    // transposed convs: deconv_gather_kernel) - the contraction then ALWAYS writes partials, also with sk = 1
    std::function<int(const float* ws, int sk, hipStream_t s)> custom_reduce;
    const char* custom_reduce_name = "";

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
    // clear a per-batch buffer in EVERY group's copy (grouped contexts; one fill per group)
    hipError_t memset_groups(void* p, size_t bytes) {
        for (int g = 0; g < c->G; ++g) {
            const hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(p) + (size_t)g * c->grp_floats * sizeof(float), 0, bytes, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }

    // launches the contraction with an explicit choice; returns the number of BN partial rows written (0 if none)
    int run_choice(const IgemmDesc& d, int rep, IgemmTile tile, int sk) {
        if (rc) return 0;
        if (sk > 1 || rep > 1 || custom_reduce) {
            IgemmDesc e = d;
            e.splitk = sk;
            e.splitk_ws = c->ws + c->bufs.at(wsname).off;
            e.amax_out = nullptr;                    // (partials are not the tensor: the pass that finishes them publishes the maximum)
            if (custom_reduce) {
                e.bias = nullptr; e.relu_out = 0; e.stats = nullptr;
                if (igemm_tile_p3(tile)) e.y = e.splitk_ws;       // the plane-fed kernels do not split K: their dense output IS partial 0
                timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(e, tile, s); });
                timed(custom_reduce_name, 0.0, [&] { return custom_reduce(e.splitk_ws, sk, s); });
                return 0;
            }
            // the partials are combined by the contraction itself (last-arriver, igemm_epilogue) unless statistics ride on the reducer
            const long tiles = (long)cdiv(d.M, igemm_tile_bm(tile)) * cdiv(d.N, igemm_tile_bn(tile));
            if (c->sk_fused && sk > 1 && !d.stats && !d.amax_out && d.N % 4 == 0 && d.ldy % 4 == 0 && ((uintptr_t)d.y % 16) == 0 && (!d.bias || ((uintptr_t)d.bias % 16) == 0) &&
                tiles <= SK_TICKETS && igemm_tile_fused_splitk(tile) && c->bufs.count(wsname + ":tk")) {
                e.sk_ticket = reinterpret_cast<int*>(c->ws + c->bufs.at(wsname + ":tk").off);
                e.sk_rep = rep;
                e.stats = nullptr;
                timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(e, tile, s); });
                return 0;
            }
            e.bias = nullptr; e.relu_out = 0; e.stats = nullptr;
            timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(e, tile, s); });
            timed("splitk_reduce_kernel", 0.0, [&] {
                return splitk_reduce_launch(e.splitk_ws, sk, d.M, d.N, d.bias, d.relu_out, d.y, d.ldy, rep, d.stats, s, d.amax_out); });
            return 0;
        }
        timed(igemm_tile_name(tile), 2.0 * d.M * d.N * d.K, [&] { return igemm_launch(d, tile, s); });
        return 0;
    }

    Choice heuristic(const IgemmDesc& d, int rep, bool allow_split) const {
        Choice ch;
        const IgemmTile tile = igemm_pick_tile(d);
        ch.tile = (int)tile;
        const bool can_split = allow_split && dense_out(d) && !d.stats && !igemm_tile_p3(tile);     // (the plane-fed kernels do not split K)
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
        // grouped contexts: with SAGEN_TUNE_GROUPS=n the candidates are timed on the first n groups only (the launchers read the group
        // count from cur_group()) - a shorter tuning pass; the default times the launches the forward will run (the choice between
        // tiles does depend on the group count: see sagen_ctx::tune_groups)
        struct FewerGroups {
            GroupInfo& g; int saved;
            explicit FewerGroups(int n) : g(cur_group()), saved(g.G) { if (n > 0 && g.G > n) g.G = n; }
            ~FewerGroups() { g.G = saved; }
        } fewer(c->tune_groups);
        Choice top[3];
        for (auto& t : top) { t = heuristic(d, rep, allow_split); t.us = 1e30f; }
        const bool dense = dense_out(d);
        static const int SKS[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64};
        for (int t = 0; t < (int)TILE_AUTO && !rc; ++t) {
            const IgemmTile tile = (IgemmTile)t;
            const int bm = igemm_tile_bm(tile), bn = igemm_tile_bn(tile);
            if (!igemm_tile_ok(d, tile)) continue;
            // conv3pp_kernel (two phase-locked teams per workgroup): faster than conv3p_kernel on its own (stage 5: 87 vs 93 us,
            // stage 4: 80 vs 83) but its 121 KiB of LDS make it the only workgroup of a CU, and with a second batch in flight the
            // forward gets slower (1 560 vs 1 610-1 637 ambisonic-s/s) - a candidate only for the training step, which has one batch in
            // flight (7.27 -> 7.23 ms per step on one box), or when asked for (SAGEN_P3PP=1)
            static const bool p3pp = getenv("SAGEN_P3PP") != nullptr;
            if (!p3pp && !c->train_mode && (tile == TILE_P3PP_PAIR || tile == TILE_P3PP_SPLITK)) continue;
            // conv3hr_kernel (three-deep activation ring) measured equal to conv3h_kernel on every layer: left out of
            // the tuner so that equal candidates do not split a layer family over two kernel names from run to run (SAGEN_P3HR=1 adds it)
            static const bool p3hr = getenv("SAGEN_P3HR") != nullptr;
            // (round 6: with ten batches per launch the wait for the activation images is no longer hidden - tools/trace_conv3h.py: 500 - 1 400 of a
            //  2 000 - 3 000-cycle group - and the deeper ring measures 0 - 2 % ahead on the headline, same box: a candidate of grouped contexts)
            if (!p3hr && c->G == 1 && (tile == TILE_P3HR_256x64 || tile == TILE_P3HR_128x64 || tile == TILE_P3HR_64x64_C2 || tile == TILE_P3HR_128x128)) continue;
            const int nk = d.Kpad / igemm_tile_bk(tile);
            if (bn > 32 && bn >= 2 * d.N) continue;                     // mostly-empty N tile
            if (bm > 32 && bm >= 4 * d.M) continue;
            for (int sk : SKS) {
                if (sk > 1 && (!allow_split || !dense)) break;
                if (sk > 1 && igemm_tile_p3(tile)) {                     // plane-fed kernels: only conv3h_kernel's dh-split (3)
                    if (!igemm_tile_dh_split(tile) || sk > 3 || c->no_dh_split) break;
                    if (sk != 3) continue;
                }
                if (sk > 1 && ((nk / sk < 4 && !igemm_tile_p3(tile)) || (size_t)sk * d.M * d.N > ws_capacity())) break;
                if (sk == 1 && (rep > 1 || custom_reduce) && (size_t)d.M * d.N > ws_capacity()) continue;
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
            if (d.stats && !rc && memset_groups(d.stats, (size_t)2 * d.N * sizeof(double)) != hipSuccess)
                rc = fail(SAGEN_ERR_HIP, "autotune: memset failed");         // candidates polluted the accumulators
        } else if (it != c->plan.end()) {
            ch = it->second;
            if (!igemm_tile_ok(d, (IgemmTile)ch.tile)) ch = heuristic(d, rep, allow_split);
            const bool dh3 = ch.splitk == 3 && igemm_tile_dh_split((IgemmTile)ch.tile);
            if (ch.splitk > 1 && (!allow_split || !dense_out(d) || (igemm_tile_p3((IgemmTile)ch.tile) && !dh3) || d.Kpad / igemm_tile_bk((IgemmTile)ch.tile) / ch.splitk < 1)) ch.splitk = 1;
        } else {
            ch = heuristic(d, rep, allow_split);
        }
        if ((ch.splitk > 1 || rep > 1 || custom_reduce) && (size_t)std::max(ch.splitk, 1) * d.M * d.N > ws_capacity()) {
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
    void fc(const float* x, int M, int K, int ldx, const std::string& name, int N, bool relu, float* y, int ldy, int rep = 1, float* amax_out = nullptr) {
        layer = name;
        IgemmDesc d;
        d.x = x; d.w = c->p("pk:" + name + "/weights"); d.y = y; d.bias = c->v(name + "/biases");
        d.M = M; d.N = N; d.K = K; d.Kpad = (K + 15) / 16 * 16;
        d.Hg = 1; d.Wg = 1; d.Hin = 1; d.Win = 1; d.Cin = K; d.ldx = ldx; d.x_bstride = ldx;
        d.ntaps = 1; d.Cout = N; d.Hlim = 1; d.Wlim = 1; d.ldy = ldy; d.y_rstride = ldy; d.y_bstride = ldy;
        d.relu_out = relu;
        d.amax_out = amax_out;
        gemm(d, rep);
    }

    // tfw.deconv_2d (core.py:96-153) as a stride-1 conv with a depth-to-space epilogue
    void deconv(const float* x, int Hin, int Win, int Cin, int l, float* y, int ldy, bool relu, int a0, int a1, int Ylim,
                long y_bstride, long y_row0, const float* mm_coeffs = nullptr, float* mm_out = nullptr, int mm_nf = 0,
                const std::function<void(IgemmDesc&)>& tweak = nullptr) {
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
        if (mm_out) { d.mm_coeffs = mm_coeffs; d.mm_out = mm_out; d.mm_row0 = (int)y_row0; d.mm_nf = mm_nf; d.mm_f_lo = 1; }
        if (tweak) tweak(d);
        gemm(d);
    }

    // tfw.deconv_2d in scatter form (igemm.hip: deconv_gather_kernel): one GEMM over the INPUT pixels of the band of rows
    // [in_row0, in_row0 + R) against the filter as [(p, q, o)][c] ("pks:" pack); output rows [y0, y1) are gathered (+ bias, ReLU) into
    // y by the pass that also sums the split-K partials
    // exact maxima of the concat buffers' halves (IgemmDesc::amax_out targets; half 0 = decoder, 1 = encoder) and the scale of the planes
    // packed from cat_l
    float* cat_amax(int l, int half) { return c->p("amax") + (size_t)((l - 1) * 2 + half) * H2_AMAX_FLOATS; }
    float* cat_a_inv(int l) { return c->p("amax") + (size_t)10 * H2_AMAX_FLOATS + l; }
    // x_planes: contract fp16x2 planes of the band (packed here from the fp32 tensor with the exact maximum its producers published)
    // on conv3g_kernel instead of splitting the fp32 operand inside igemm3_kernel's K loop; lx = the index of the concat buffer x is
    void deconv_scatter(const float* x, int Hin, int Win, int Cin, int l, float* y, int ldy, bool relu, int in_row0, int R, int y0, int y1,
                        float* amax_out = nullptr, bool x_planes = false) {
        const std::string name = "separation/deconv" + std::to_string(l + 1);
        layer = name;
        DeconvGather gd;
        gd.B = c->B; gd.Hin = Hin; gd.Win = Win; gd.kh = AENC_K[l][0]; gd.kw = AENC_K[l][1]; gd.sh = AENC_S[l][0]; gd.sw = AENC_S[l][1];
        gd.Cout = l == 0 ? c->nsep : AENC_F[l - 1];
        gd.in_row0 = in_row0; gd.R = R; gd.y0 = y0; gd.y1 = y1; gd.relu = relu ? 1 : 0; gd.ldy = ldy;
        gd.bias = c->v(name + "/biases"); gd.y = y; gd.amax_out = amax_out;
        IgemmDesc d;                           // a one-tap "conv" over the band: rows m = (b, row, column), dense partials [M][N]
        d.x = x + (long)in_row0 * Win * Cin; d.w = c->p("pks:" + name + "/weights");
        d.M = c->B * R * Win; d.N = gd.kh * gd.kw * gd.Cout; d.K = Cin; d.Kpad = Cin;
        d.Hg = R; d.Wg = Win; d.Hin = Hin - in_row0; d.Win = Win; d.Cin = Cin; d.ldx = Cin; d.x_bstride = (long)Hin * Win * Cin;
        d.ntaps = 1; d.Cout = d.N; d.Hlim = R; d.Wlim = Win; d.ldy = d.N; d.y_rstride = (long)Win * d.N; d.y_bstride = (long)R * Win * d.N;
        d.y = c->ws + c->bufs.at(wsname).off;            // (never written: the contraction leaves partials only)
        if (x_planes && !c->tuning) {              // a plan that names a register-staged tile keeps the fp32 operand (no pack pass)
            auto it = c->plan.find(name);
            if (it != c->plan.end() && !igemm_tile_p3((IgemmTile)it->second.tile)) x_planes = false;
        }
        if (x_planes) {
            const int lx = l + 1;
            void* planes = c->p("catp");
            timed("h2_pack_rows_kernel", 0.0, [&] {
                return h2_pack_rows_launch(x, (long)Hin * Win * Cin, (long)Win * Cin, Cin, in_row0, c->B, R, Win, Cin, cat_amax(lx, 0), cat_amax(lx, 1), planes,
                                           cat_a_inv(lx), reinterpret_cast<unsigned*>(c->p("h2s") + 7), s); });
            d.xp3 = planes; d.xp3_fmt = 1; d.xp3_row0 = 0; d.xp3_rows = R;
            d.p3_np = c->B * R * (Win + 1);
            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
            d.xp3_bytes = (unsigned)((size_t)d.xp3_cstride * (Cin / 16));
            d.wh2 = c->p("pkhs:" + name + "/weights"); d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
            d.h2_a_inv = cat_a_inv(lx); d.h2_w_inv = c->p("h2s") + c->h2_slot.at("pks:" + name);
        }
        custom_reduce_name = "deconv_gather_kernel";
        custom_reduce = [gd](const float* ws, int sk, hipStream_t st) { return deconv_gather_launch(ws, sk, gd, st); };
        gemm(d);
        custom_reduce = nullptr;
    }

    // fp16x2 planes (conv3h.hip) for this forward?  (the training step's forward too, unless SAGEN_TRAIN_NO_H2=1: its backward reads the
    // retained fp32 activations, not the planes)
    bool h2() const { return c->use_h2 && c->use_p3 && !c->fp32_only && (!c->train_mode || c->train_h2); }
    int p3_fmt() const { return h2() ? 1 : 0; }
    float* h2_a_inv() { return c->p("h2s") + (sfx.empty() ? 0 : 1); }        // 2^-ka of the planes currently in this trunk's plane buffer
    // lean trunk (round 5, resnet()): the block input / output planes live in buffer "p3" with two alternating scale slots (a merge reads
    // its residual under the old scale while it publishes the new one), the conv_2 input planes in "p3b" with their own
    float* h2_a_inv_x(int parity) { return c->p("h2s") + H2S_A_INV_X + 2 * (parity & 1) + (sfx.empty() ? 0 : 1); }
    float* h2_a_inv_b() { return c->p("h2s") + H2S_A_INV_B + (sfx.empty() ? 0 : 1); }
    // statistical bound of the current block input / output (conv3h.hip, p3.hip: P3hScale), two slots used alternately so that a merge
    // reads its input's bound from one and publishes its output's bound into the other
    float* h2_xbound(int parity) { return c->p("h2s") + 2 + 2 * (parity & 1) + (sfx.empty() ? 0 : 1); }
    P3hScale h2_scale(float* bound_out = nullptr, const float* res_bound = nullptr, const double* res_acc = nullptr, double res_inv_count = 0.0) {
        P3hScale h;
        h.a_inv = h2_a_inv(); h.bound_out = bound_out; h.res_bound = res_bound; h.res_acc = res_acc; h.res_inv_count = res_inv_count;
        h.sat_count = reinterpret_cast<unsigned*>(c->p("h2s") + 7);
        return h;
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
                 const void* planes = nullptr, const float* planes_a_inv = nullptr) {
        if (rc) return;
        IgemmDesc d = conv_desc(x, Hin, Win, Cin, Cin, c->p("pk:" + name + "/weights"), k, k, stride, stride, true, Cout, y,
                                Cout, Hout, Wout);
        if (planes) {                       // the input as pre-split planes (p3.hip); x may be null then
            d.xp3 = planes;
            d.p3_np = c->B * Hin * (Win + 1);
            auto hs = c->h2_slot.find(name);
            if (h2() && (stride == 1 || c->use_p3g) && k == 3 && hs != c->h2_slot.end()) {       // two fp16 planes + the layer's fp16 filter planes
                d.xp3_fmt = 1;
                d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
                d.xp3_bytes = (unsigned)p3h_bytes(c->B, Hin, Win, Cin);
                d.wh2 = c->p("pkh:" + name + "/weights");
                d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
                d.h2_a_inv = planes_a_inv ? planes_a_inv : h2_a_inv();
                d.h2_w_inv = c->p("h2s") + hs->second;
            } else {
                d.xp3_cstride = (unsigned)((size_t)d.p3_np * 96);
                d.xp3_bytes = (unsigned)p3_bytes(c->B, Hin, Win, Cin);
            }
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
        // uint8 frames: the centred bf16 plane u - 128 IS the exact operand (x = (u' + 0.5) / 255): one plane, three products (stem8.hip)
        // (a grouped context has no other stem: its autotune pass runs the fused kernels too - they take no tile from the plan)
        const bool tune_general = c->tuning && c->G == 1;
        const bool fast8 = c->video_u8 && scope == "video_encoder" && c->stem8 && !tune_general && !c->fp32_only && !c->train_mode;
        // ... with the filter as two fp16 planes where they exist: two products per multiply instead of three (stem8.hip, MODE 2)
        const bool fast8h = fast8 && c->stem8h && h2() && c->h2_slot.count(scope + "/conv1/conv") != 0;
        // float frames (the flow encoder; video handed over as float32): the same kernel on two fp16 planes of the frame (stem8.hip, F16)
        const bool fast16 = !fast8 && !(c->video_u8 && scope == "video_encoder") && c->stem16 && !tune_general && !c->fp32_only && !c->train_mode && h2() &&
                            c->use_p3 && c->p3_from_stage <= 2 && c->bufs.count("s16:part" + sfx) != 0 && c->h2_slot.count(scope + "/conv1/conv") != 0;
        // the batch-norm accumulators start at zero: cleared by the trunk's first kernel where that is stem8_prep / stem16_amax, else by a fill
        if (!fast8 && !fast16 && !rc && hipMemsetAsync(c->p("bnacc" + sfx), 0, c->bufs.at("bnacc" + sfx).n * sizeof(float), s) != hipSuccess)
            rc = fail(SAGEN_ERR_HIP, "hipMemsetAsync(bn accumulators) failed");
        layer = scope + "/pad";
        const bool pool_planes = c->use_p3 && c->p3_from_stage <= 2;       // the pooled tensor is also wanted as planes (operand of conv2_1/conv_1)
        // Lean trunk (round 5): with fp16x2 planes from stage 2 on and the plane-fed stride-2 kernels, NOTHING reads a block output as
        // fp32 except the trunk's end: the merges read their identity residual from the block-input planes (in place: buffer "p3"
        // holds block inputs / outputs, "p3b" the conv_2 inputs) and write planes only - 4 bytes per element less per merge, no fp32
        // copy of the pooled tensor.  The planes carry the value conv_1 saw (22 significant bits + the residual's sign).
        const bool no_lean = c->no_lean_trunk;
        const bool lean = !no_lean && h2() && pool_planes && c->use_p3g && c->bufs.count("p3b" + sfx) != 0 && !c->train_mode;
        float* const s16_a_inv = c->p("h2s") + H2S_S16_A_INV + (sfx.empty() ? 0 : 1);        // 2^-ka of the frame planes
        if (fast16)
            timed("stem16_amax_kernel+stem16_prep_kernel", 0.0, [&] { return stem16_prep_launch(img, c->p("xpad" + sfx), c->p("s16:part" + sfx), s16_a_inv, B, s,
                                                                                               c->p("bnacc" + sfx), (long)c->bufs.at("bnacc" + sfx).n); });
        else if (fast8)
            timed("stem8_prep_kernel", 0.0, [&] { return stem8_prep_launch(reinterpret_cast<const unsigned char*>(img), c->p("xpad" + sfx), B, s,
                                                                          c->p("bnacc" + sfx), (long)c->bufs.at("bnacc" + sfx).n, fast8h ? 1 : 0); });
        else if (c->video_u8 && scope == "video_encoder")
            timed("pad_u8_nhwc3to4_kernel", 0.0, [&] { return pad_u8_nhwc3to4_launch(reinterpret_cast<const unsigned char*>(img), c->p("xpad" + sfx), B, 224, 448, 2, 3, 2, 4, s); });
        else
            timed("pad_nhwc3to4_kernel", 0.0, [&] { return pad_nhwc3to4_launch(img, c->p("xpad" + sfx), B, 224, 448, 2, 3, 2, 4, s); });
        // conv1 7x7/2 SAME == VALID 7x8 (8th tap column = zero weights) on the padded 4-channel image
        int H = 0, W = 0;
        if (!rc && wait_before_mfma) {
            if (hipStreamWaitEvent(s, wait_before_mfma, 0) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "hipStreamWaitEvent failed");
            wait_before_mfma = nullptr;
        }
        {
            const std::string name = scope + "/conv1/conv";
            const bool fused = c->stem_fused && !c->tuning && !c->fp32_only && !(c->use_p3 && c->p3_from_stage <= 2);
            if (fast8 || fast16) {
                H = 112; W = 224;
                layer = name + "+pool";
                if (fast16)
                    timed("stem8pool_kernel<f16>", 2.0 * B * H * W * 64 * 224, [&] {
                        return stem16pool_launch(c->p("xpad" + sfx), c->p("pkh:" + name + "/weights"), c->v(name + "/bn/gamma"), c->p("rx0" + sfx), bn_acc(li),
                                                 s16_a_inv, c->p("h2s") + c->h2_slot.at(name), B, s); });
                else if (fast8h)
                    timed("stem8pool_kernel<h2>", 2.0 * B * H * W * 64 * 224, [&] {
                        return stem8pool_h2_launch(c->p("xpad" + sfx), c->p("pk:" + name + "/weights"), c->p("pkh:" + name + "/weights"), c->p("h2s") + c->h2_slot.at(name),
                                                   c->v(name + "/bn/gamma"), c->p("rx0" + sfx), bn_acc(li), B, s); });
                else
                timed("stem8pool_kernel", 2.0 * B * H * W * 64 * 224, [&] {
                    return stem8pool_launch(c->p("xpad" + sfx), c->p("pk:" + name + "/weights"), c->v(name + "/bn/gamma"), c->p("rx0" + sfx), bn_acc(li), B, s); });
                const BnRef bnf = bn_ref(li, name, (long)B * H * W);
                layer = name + "/bn-relu";
                P3hScale hs0 = h2_scale(h2_xbound(0));
                if (lean) hs0.a_inv = h2_a_inv_x(0);
                if (pool_planes)       // BN + ReLU of the pooled tensor as planes (lean trunk: planes only - the residual of conv2_1 is read from them) and in place
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("rx0" + sfx), nullptr, nullptr, bnf, nullptr, 1, lean ? nullptr : c->p("rx0" + sfx), c->p("p3" + sfx), B, 56, 112, 64, s, p3_fmt(), &hs0); });
                else
                    timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(c->p("rx0" + sfx), nullptr, nullptr, bnf, nullptr, c->p("rx0" + sfx), (long)B * 56 * 112, 64, s); });
            } else if (fused) {
                // conv + statistics + pool of the RAW output in one kernel (max or min per channel by the sign of gamma), then BN + ReLU on
                // the pooled tensor in place: relu(bn(.)) is monotone per channel, so this IS maxpool(relu(bn(conv))) (stempool.hip)
                H = 112; W = 224;
                layer = name + "+pool";
                timed("stempool_kernel", 2.0 * B * H * W * 64 * 224, [&] {
                    return stempool_launch(c->p("xpad" + sfx), c->p("pk:" + name + "/weights"), c->v(name + "/bn/gamma"), c->p("rx0" + sfx), bn_acc(li), B, s); });
                const BnRef bnf = bn_ref(li, name, (long)B * H * W);
                layer = name + "/bn-relu";
                timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(c->p("rx0" + sfx), nullptr, nullptr, bnf, nullptr, c->p("rx0" + sfx), (long)B * 56 * 112, 64, s); });
            } else {
            IgemmDesc d = conv_desc(c->p("xpad" + sfx), 229, 454, 4, 4, c->p("pk:" + name + "/weights"), 7, 8, 2, 2, false, 64,
                                    c->p("y0" + sfx), 64, H, W);
            d.stats = bn_acc(li);
            layer = name;
            contract(d);
            const BnRef bn = bn_ref(li, name, (long)B * H * W);
            if (c->use_p3 && c->p3_from_stage <= 2)       // pooled block input as fp32 (residual) AND as planes (operand of conv2_1/conv_1)
            {
                P3hScale hs0 = h2_scale(h2_xbound(0));
                if (lean) hs0.a_inv = h2_a_inv_x(0);
                timed("p3_maxpool_kernel", 0.0, [&] { return p3_maxpool_launch(c->p("y0" + sfx), nullptr, nullptr, bn, lean ? nullptr : c->p("rx0" + sfx), c->p("p3" + sfx), B, H, W, 64, s, p3_fmt(), &hs0); });
            }
            else
                timed("maxpool3x3s2_kernel", 0.0, [&] { return maxpool3x3s2_launch(c->p("y0" + sfx), nullptr, nullptr, bn, c->p("rx0" + sfx), B, H, W, 64, s); });
            }
            ++li;
            H = (H + 1) / 2; W = (W + 1) / 2;
        }
        float* xin = c->p("rx0" + sfx);
        float* xout = c->p("rx1" + sfx);
        int cin = 64;
        const int couts[4] = {64, 128, 256, 512};
        bool x_in_planes = pool_planes;        // the pooled tensor exists as planes (p3_pack after the fused uint8 stem, p3_maxpool otherwise)
        int xb_par = 0;                                                 // which h2_xbound slot holds the current block input's bound
        bool x_fp32_valid = !lean;                                      // false: the previous pass wrote the block input as planes only (lean trunk: already the pool)
        for (int st = 0; st < 4; ++st) {
            const int cout = couts[st];
            for (int unit = 1; unit <= 2; ++unit) {
                const std::string pfx = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(unit);
                const bool first = unit == 1 && cin != cout;
                const int stride = first ? 2 : 1;
                int Ho = 0, Wo = 0;
                const float* shortcut = xin;
                // Pre-split planes pay where the tensors are small next to the contraction: per residual block the plane-writing
                // passes cost 77 / 38 / 27 / 22 us (stage 2..5, batch 32) against ~30 / 15 / 8 / 5 us for the fp32 BN passes they
                // replace, while conv3p saves ~19 us per conv at every stage (profiles/r02_*): stage 2 stays on igemm3dw.
                const bool p3_here = c->use_p3 && st + 2 >= c->p3_from_stage;
                void* planes = p3_here ? (void*)c->p((lean ? "p3b" : "p3") + sfx) : nullptr;          // of conv_2's input
                void* const xplanes = p3_here ? (void*)c->p("p3" + sfx) : nullptr;                     // of the block input / output
                const float* const x_ai = lean ? h2_a_inv_x(xb_par) : nullptr;                        // (null: the trunk's one slot)
                // the block input as planes: written by the previous block's merge (x_in_planes) - stride-1 conv_1 (conv3p_kernel) and,
                // since round 4, the stride-2 conv_1 + 1x1 shortcut of a stage's first block (conv3g_kernel: gathered operand tiles)
                const void* in_planes = (p3_here && x_in_planes) ? xplanes : nullptr;
                if (first) {   // 1x1/2 projection, no bias, no BN (resnet.py:211-212)
                    IgemmDesc d = conv_desc(xin, H, W, cin, cin, c->p("pk:" + pfx + "/shortcut/weights"), 1, 1, 2, 2, true,
                                            cout, c->p("rsc" + sfx), cout, Ho, Wo);
                    if (!x_fp32_valid) d.x = nullptr;                   // the merge wrote this block input as planes only
                    if (in_planes) {
                        d.xp3 = in_planes;
                        d.p3_np = B * H * (W + 1);
                        auto hs = c->h2_slot.find(pfx + "/shortcut");
                        if (h2() && hs != c->h2_slot.end()) {
                            d.xp3_fmt = 1;
                            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
                            d.xp3_bytes = (unsigned)p3h_bytes(B, H, W, cin);
                            d.wh2 = c->p("pkh:" + pfx + "/shortcut/weights");
                            d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
                            d.h2_a_inv = x_ai ? x_ai : h2_a_inv();
                            d.h2_w_inv = c->p("h2s") + hs->second;
                        } else {
                            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 96);
                            d.xp3_bytes = (unsigned)p3_bytes(B, H, W, cin);
                        }
                    }
                    if (h2() && p3_here) d.stats = bn_acc(20 + st);      // (sum, sumsq) of the projection: the residual's magnitude for the merge's fp16 scale
                    layer = pfx + "/shortcut";
                    gemm(d, 1, false);
                    shortcut = c->p("rsc" + sfx);
                }
                conv_bn(x_fp32_valid ? xin : nullptr, H, W, cin, pfx + "/conv_1", 3, stride, cout, BnRef(), c->p("ry1" + sfx), Ho, Wo, li, "", in_planes, x_ai);
                const BnRef bn1 = bn_ref(li, pfx + "/conv_1", (long)B * Ho * Wo);
                ++li;
                int H2, W2;
                if (p3_here) {
                    // relu(bn1(y1)) -> planes (one elementwise pass), conv_2 on the planes, then the residual merge, which also
                    // writes the planes of the block output when the next conv_1 is a stride-1 3x3 (unit 1 of a stage)
                    layer = pfx + "/bn1-relu";
                    P3hScale hs1 = h2_scale();
                    if (lean) hs1.a_inv = h2_a_inv_b();
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("ry1" + sfx), nullptr, nullptr, bn1, nullptr, 1, nullptr, planes, B, Ho, Wo, cout, s, p3_fmt(), &hs1); });
                    conv_bn(nullptr, Ho, Wo, cout, pfx + "/conv_2", 3, 1, cout, BnRef(), c->p("ry2" + sfx), H2, W2, li, "", planes, lean ? h2_a_inv_b() : nullptr);
                    const BnRef bn2 = bn_ref(li, pfx + "/conv_2", (long)B * Ho * Wo);
                    layer = pfx + "/merge";
                    // the block output as planes when the next conv_1 reads planes: the stride-1 3x3 of this stage (unit 1), or the
                    // stride-2 conv_1 + shortcut of the next stage's first block (conv3g_kernel)
                    // ... whose consumers read nothing else: that block output is written as planes ONLY (hi + mid + lo IS the value)
                    const bool to_next_stage = unit == 2 && c->use_p3g && st < 3;
                    const bool next_p3 = unit == 1 || to_next_stage;
                    // the residual's statistical bound: tracked from the previous pass (identity) or from the shortcut conv's statistics
                    P3hScale hs2 = first ? h2_scale(h2_xbound(xb_par ^ 1), nullptr, bn_acc(20 + st), 1.0 / ((double)B * Ho * Wo))
                                         : h2_scale(h2_xbound(xb_par ^ 1), h2_xbound(xb_par));
                    // lean trunk: an identity residual comes from the block-input planes (overwritten in place by the block output,
                    // under the other scale slot); only the trunk's last block writes fp32
                    const bool res_planes = lean && !first;
                    const bool y_fp32 = lean ? !next_p3 : !to_next_stage;
                    if (lean) {
                        hs2.a_inv = h2_a_inv_x(xb_par ^ 1);
                        if (res_planes) { hs2.res_planes = xplanes; hs2.res_a_inv = h2_a_inv_x(xb_par); }
                    }
                    xb_par ^= 1;
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("ry2" + sfx), nullptr, nullptr, bn2, res_planes ? nullptr : shortcut, 1, y_fp32 ? xout : nullptr, next_p3 ? xplanes : nullptr, B, Ho, Wo, cout, s, p3_fmt(), &hs2); });
                    x_in_planes = next_p3;
                    x_fp32_valid = y_fp32;
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
                // the last block of a stage that keeps fp32 activations, in front of a stage on planes: its output feeds only the
                // stride-2 conv_1 + shortcut of that stage's first block (conv3g_kernel) and is written as planes only
                const bool planes_out = unit == 2 && st < 3 && c->use_p3g && c->use_p3 && st + 3 >= c->p3_from_stage;
                if (planes_out)
                    timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("ry2" + sfx), nullptr, nullptr, bn2, shortcut, 1, nullptr, c->p("p3" + sfx), B, Ho, Wo, cout, s); });
                else
                    timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(c->p("ry2" + sfx), nullptr, nullptr, bn2, shortcut, xout, (long)B * Ho * Wo, cout, s); });
                x_in_planes = planes_out;
                x_fp32_valid = !planes_out;
                ++li;
                std::swap(xin, xout);
                H = Ho; W = Wo; cin = cout;
            }
        }
        return xin;
    }

    // The same trunk for the training step: every tensor the backward pass reads is retained in the train workspace -
    // raw conv outputs "t:y1:k" / "t:y2:k" (BN backward), conv_2 inputs "t:a1:k" and block outputs "t:out:k" (weight gradients,
    // ReLU masks), the pool output "t:x0"; the raw stem output "y0" and the BN accumulators are never reused anyway.
    // With train_h2w the planes the forward writes for conv3h_kernel are RETAINED per layer ("t:pl:a1:k": input of block k's conv_2,
    // "t:pl:x:k": input of block k's stride-1 conv_1; scales in "t:h2a": slot k / 8 + k): the weight gradients contract them (wgrad3h.hip).
    bool h2w() const { return c->train_mode && c->train_h2w && c->train_h2d && h2() && c->p3_from_stage <= 2 && wgrad_planes_enabled() && c->tbufs.count("t:h2a" + sfx) != 0; }
    void* pl_buf(const char* what, int k) { return c->p(std::string("t:pl:") + what + ":" + std::to_string(k) + sfx); }
    float* pl_a_inv(int slot) { return c->p("t:h2a" + sfx) + slot; }
    const float* resnet_train(const float* img, const std::string& scope) {
        const int B = c->B;
        const bool keep = h2w();
        int li = 0;
        if (!rc && hipMemsetAsync(c->p("bnacc" + sfx), 0, c->bufs.at("bnacc" + sfx).n * sizeof(float), s) != hipSuccess)
            rc = fail(SAGEN_ERR_HIP, "hipMemsetAsync(bn accumulators) failed");
        layer = scope + "/pad";
        if (c->video_u8 && scope == "video_encoder")
            timed("pad_u8_nhwc3to4_kernel", 0.0, [&] { return pad_u8_nhwc3to4_launch(reinterpret_cast<const unsigned char*>(img), c->p("xpad" + sfx), B, 224, 448, 2, 3, 2, 4, s); });
        else
            timed("pad_nhwc3to4_kernel", 0.0, [&] { return pad_nhwc3to4_launch(img, c->p("xpad" + sfx), B, 224, 448, 2, 3, 2, 4, s); });
        int H = 0, W = 0;
        // uint8 frames (sagen_train_step_u8): the stem's forward on the exact one-plane operand (stem8.hip, RAW: y0 is kept for the
        // backward); the float xpad above is still what the stem's weight gradient contracts
        const bool fast8 = c->video_u8 && scope == "video_encoder" && c->stem8 && !c->tuning && !c->fp32_only && c->tbufs.count("t:s8plane" + sfx) != 0;
        if (fast8) {
            layer = scope + "/plane";
            timed("stem8_prep_kernel", 0.0, [&] { return stem8_prep_launch(reinterpret_cast<const unsigned char*>(img), c->p("t:s8plane" + sfx), B, s); });
        }
        if (!rc && wait_before_mfma) {
            if (hipStreamWaitEvent(s, wait_before_mfma, 0) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "hipStreamWaitEvent failed");
            wait_before_mfma = nullptr;
        }
        {
            const std::string name = scope + "/conv1/conv";
            // ... and, where the pooled tensor goes on as planes, the pool of the raw output in the same kernel (max, or min where gamma < 0:
            // relu(bn(.)) is monotone per channel, so BN + ReLU of the pooled raw tensor IS maxpool(relu(bn(y0))), bit for bit) - the pool
            // pass no longer reads the 205 MB raw tensor a second time
            const bool rawpool = fast8 && keep && c->train_rawpool && c->bufs.count("rx0" + sfx) != 0;
            if (scope == "video_encoder") c->rawpool_last = rawpool;
            if (rawpool) {
                H = 112; W = 224;
                layer = name;
                timed("stem8pool_kernel<raw+pool>", 2.0 * B * H * W * 64 * 224, [&] {
                    return stem8rawpool_launch(c->p("t:s8plane" + sfx), c->p("pk:" + name + "/weights"), c->v(name + "/bn/gamma"), c->p("y0" + sfx), c->p("rx0" + sfx),
                                               bn_acc(li), B, s); });
            } else if (fast8) {
                H = 112; W = 224;
                layer = name;
                timed("stem8pool_kernel<raw>", 2.0 * B * H * W * 64 * 224, [&] {
                    return stem8raw_launch(c->p("t:s8plane" + sfx), c->p("pk:" + name + "/weights"), c->p("y0" + sfx), bn_acc(li), B, s); });
            } else {
                IgemmDesc d = conv_desc(c->p("xpad" + sfx), 229, 454, 4, 4, c->p("pk:" + name + "/weights"), 7, 8, 2, 2, false, 64,
                                        c->p("y0" + sfx), 64, H, W);
                d.stats = bn_acc(li);
                layer = name;
                contract(d);
            }
            const BnRef bn = bn_ref(li, name, (long)B * H * W);
            if (rawpool) {
                P3hScale hs0 = h2_scale(h2_xbound(0));
                hs0.a_inv = pl_a_inv(8);
                layer = name + "/bn-relu";
                timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(c->p("rx0" + sfx), nullptr, nullptr, bn, nullptr, 1, c->p("t:x0" + sfx), pl_buf("x", 0), B, 56, 112, 64, s, 1, &hs0); });
            } else if (keep) {                         // the pooled block input also as (retained) planes: stage 2 runs on planes as well
                P3hScale hs0 = h2_scale(h2_xbound(0));
                hs0.a_inv = pl_a_inv(8);
                timed("p3_maxpool_kernel", 0.0, [&] { return p3_maxpool_launch(c->p("y0" + sfx), nullptr, nullptr, bn, c->p("t:x0" + sfx), pl_buf("x", 0), B, H, W, 64, s, 1, &hs0); });
            } else {
                timed("maxpool3x3s2_kernel", 0.0, [&] { return maxpool3x3s2_launch(c->p("y0" + sfx), nullptr, nullptr, bn, c->p("t:x0" + sfx), B, H, W, 64, s); });
            }
            ++li;
            H = (H + 1) / 2; W = (W + 1) / 2;
        }
        if (!rc && wait_packs) {                 // everything from here on reads filters the second stream packed meanwhile
            if (hipStreamWaitEvent(s, wait_packs, 0) != hipSuccess) rc = fail(SAGEN_ERR_HIP, "hipStreamWaitEvent failed");
            wait_packs = nullptr;
        }
        const float* xin = c->p("t:x0" + sfx);
        int cin = 64;
        int xb_par = 0;
        const int couts[4] = {64, 128, 256, 512};
        for (int st = 0; st < 4; ++st) {
            const int cout = couts[st];
            for (int unit = 1; unit <= 2; ++unit) {
                const std::string pfx = scope + "/conv" + std::to_string(st + 2) + "_" + std::to_string(unit);
                const std::string k = std::to_string(2 * st + unit - 1) + sfx;
                const bool first = unit == 1 && cin != cout;
                const int stride = first ? 2 : 1;
                int Ho = 0, Wo = 0, H2, W2;
                const float* shortcut = xin;
                const int kidx = 2 * st + unit - 1;
                const bool p3_here = c->use_p3 && st + 2 >= c->p3_from_stage && (st + 2 >= 3 || keep);   // (without retained planes the pool writes none)
                void* planes = p3_here ? (keep ? pl_buf("a1", kidx) : (void*)c->p("p3" + sfx)) : nullptr;       // of a1
                const float* planes_ai = keep && p3_here ? pl_a_inv(kidx) : nullptr;
                // planes of the block input (written by the pool / by the previous block's merge) and where this block's merge writes the next one's
                void* xplanes = p3_here ? (keep ? pl_buf("x", kidx) : (void*)c->p("p3" + sfx)) : nullptr;
                const float* xplanes_ai = keep && p3_here ? pl_a_inv(8 + kidx) : nullptr;
                // (with retained planes the last block of stages 2-4 writes them too: the stride-2 conv_1 + 1x1 shortcut of the next
                //  stage's first block run conv3g_kernel on them, as in inference)
                const bool to_next_stage = keep && unit == 2 && st < 3 && c->use_p3g;
                void* nplanes = p3_here && (unit == 1 || to_next_stage) ? (keep ? pl_buf("x", kidx + 1) : (void*)c->p("p3" + sfx)) : nullptr;
                // unit 2 of a plane stage finds the planes of its input written by unit 1's merge (stage 2's unit 1: by the pool)
                const bool x_planes = p3_here && (stride == 1 ? (unit == 2 || (keep && st == 0)) : (keep && c->use_p3g));
                if (first) {
                    IgemmDesc d = conv_desc(xin, H, W, cin, cin, c->p("pk:" + pfx + "/shortcut/weights"), 1, 1, 2, 2, true,
                                            cout, c->p("rsc" + sfx), cout, Ho, Wo);
                    auto hsc = c->h2_slot.find(pfx + "/shortcut");
                    if (x_planes && hsc != c->h2_slot.end()) {       // the projection on the planes of the block input (conv3g_kernel)
                        d.xp3 = xplanes;
                        d.p3_np = B * H * (W + 1);
                        d.xp3_fmt = 1;
                        d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
                        d.xp3_bytes = (unsigned)p3h_bytes(B, H, W, cin);
                        d.wh2 = c->p("pkh:" + pfx + "/shortcut/weights");
                        d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
                        d.h2_a_inv = xplanes_ai;
                        d.h2_w_inv = c->p("h2s") + hsc->second;
                    }
                    if (h2() && p3_here) d.stats = bn_acc(20 + st);      // the projection's (sum, sumsq): the residual bound of the merge's fp16 scale
                    layer = pfx + "/shortcut";
                    gemm(d, 1, false);
                    shortcut = c->p("rsc" + sfx);
                }
                float* y1 = c->p("t:y1:" + k); float* a1 = c->p("t:a1:" + k); float* y2 = c->p("t:y2:" + k); float* xout = c->p("t:out:" + k);
                conv_bn(xin, H, W, cin, pfx + "/conv_1", 3, stride, cout, BnRef(), y1, Ho, Wo, li, "",
                        x_planes ? xplanes : nullptr, x_planes ? xplanes_ai : nullptr);
                const BnRef bn1 = bn_ref(li, pfx + "/conv_1", (long)B * Ho * Wo);
                ++li;
                layer = pfx + "/bn1-relu";
                P3hScale hs1 = h2_scale();
                if (planes_ai) hs1.a_inv = const_cast<float*>(planes_ai);
                // with retained planes nobody reads the fp32 a1: conv_2 and its weight gradient take the planes, conv_1's batch-norm
                // backward re-derives the ReLU mask from y1 (unless a debugging switch asks for the fp32 tensor)
                static const bool a1_wanted = getenv("SAGEN_BN_READ_ACT") != nullptr || getenv("SAGEN_TRAIN_KEEP_A1") != nullptr;
                float* a1w = (keep && p3_here && !a1_wanted && !wgrad_reads_fp32_operands()) ? nullptr : a1;
                if (p3_here) timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(y1, nullptr, nullptr, bn1, nullptr, 1, a1w, planes, B, Ho, Wo, cout, s, p3_fmt(), &hs1); });
                else timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(y1, nullptr, nullptr, bn1, nullptr, a1, (long)B * Ho * Wo, cout, s); });
                conv_bn(p3_here ? a1w : a1, Ho, Wo, cout, pfx + "/conv_2", 3, 1, cout, BnRef(), y2, H2, W2, li, p3_here ? "" : pfx + "/conv_2#mat", planes, planes_ai);
                const BnRef bn2 = bn_ref(li, pfx + "/conv_2", (long)B * Ho * Wo);
                ++li;
                layer = pfx + "/merge";
                P3hScale hs2 = first ? h2_scale(h2_xbound(xb_par ^ 1), nullptr, bn_acc(20 + st), 1.0 / ((double)B * Ho * Wo))
                                     : h2_scale(h2_xbound(xb_par ^ 1), h2_xbound(xb_par));
                if (keep && p3_here) {
                    hs2.a_inv = pl_a_inv(8 + kidx + 1);
                    hs2.relu_bits = reinterpret_cast<unsigned char*>(c->p("t:mb:" + k));       // the block output's ReLU mask, one bit per element
                }
                if (p3_here) xb_par ^= 1;
                if (p3_here) timed("p3_pack_kernel", 0.0, [&] { return p3_pack_launch(y2, nullptr, nullptr, bn2, shortcut, 1, xout, nplanes, B, Ho, Wo, cout, s, p3_fmt(), &hs2); });
                else timed("bn_apply_relu_kernel", 0.0, [&] { return bn_apply_relu_launch(y2, nullptr, nullptr, bn2, shortcut, xout, (long)B * Ho * Wo, cout, s); });
                xin = xout;
                H = Ho; W = Wo; cin = cout;
            }
        }
        return xin;
    }
};

}  // namespace sagen
