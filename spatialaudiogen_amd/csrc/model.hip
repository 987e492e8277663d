// sagen_ctx: the native runtime of the inference path — variable inventory, workspace carving,
// filter repacking and the launch sequence that replaces one sess.run of
// SptAudioGen.inference_ops (reference model.py:356-434; called at deploy.py:141, eval.py:145).
#include "model.h"

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
void sagen_destroy_impl(sagen_ctx* c);
int sagen_groups_impl(const sagen_ctx* c) { return c->G; }
int sagen_create_impl(sagen_ctx** out, const sagen_config* cfg, int groups) {
    if (!out || !cfg) return fail(SAGEN_ERR_NULL, "sagen_create: null argument");
    if (groups < 1 || groups > SAGEN_MAX_GROUPS) return fail(SAGEN_ERR_SHAPE, "sagen_create_grouped: groups=%d (1..%d)", groups, SAGEN_MAX_GROUPS);
    if (groups > 1 && cfg->separation != SAGEN_SEP_FREQ_MASK)
        return fail(SAGEN_ERR_UNSUPPORTED, "sagen_create_grouped: the grouped launch covers the FREQ_MASK path only");
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
    c->G = groups;

    c->fp32_only = getenv("SAGEN_FP32_ONLY") != nullptr;
    c->use_p3 = !c->fp32_only && getenv("SAGEN_NO_P3") == nullptr;
    if (const char* e = getenv("SAGEN_P3_FROM_STAGE")) c->p3_from_stage = atoi(e);
    // fused stem + pool (stempool.hip): +0.8 % for a host that runs one forward at a time; with several batches in flight
    // (SAGEN_ONE_STREAM=1 hosts, bench.py) its one-workgroup-per-CU LDS footprint blocks co-residency and it is a wash (-0.3 %)
    c->stem_fused = getenv("SAGEN_NO_STEMPOOL") == nullptr && (getenv("SAGEN_ONE_STREAM") == nullptr || getenv("SAGEN_STEMPOOL") != nullptr);
    c->stem8 = getenv("SAGEN_NO_STEM8") == nullptr;
    c->stem16 = getenv("SAGEN_NO_STEM16") == nullptr;
    c->stem8h = getenv("SAGEN_NO_STEM8_H2") == nullptr;
    c->use_h2 = getenv("SAGEN_NO_H2") == nullptr;
    c->train_h2 = getenv("SAGEN_TRAIN_NO_H2") == nullptr;
    c->train_h2d = getenv("SAGEN_TRAIN_NO_H2D") == nullptr;
    c->train_h2w = getenv("SAGEN_TRAIN_NO_H2W") == nullptr;
    c->sk_fused = getenv("SAGEN_SK_FUSED") != nullptr;
    c->no_dh_split = getenv("SAGEN_NO_DH_SPLIT") != nullptr;
    c->no_aud_planes = getenv("SAGEN_NO_AUDIO_PLANES") != nullptr;
    if (const char* e = getenv("SAGEN_TUNE_GROUPS")) c->tune_groups = atoi(e);
    c->no_scatter = getenv("SAGEN_NO_DECONV_SCATTER") != nullptr;
    c->train_bands = getenv("SAGEN_TRAIN_NO_BANDS") == nullptr;
    c->train_rawpool = getenv("SAGEN_TRAIN_NO_RAWPOOL") == nullptr;
    c->no_d1_planes = getenv("SAGEN_NO_DECONV1_PLANES") != nullptr;
    c->no_lean_trunk = getenv("SAGEN_NO_LEAN_TRUNK") != nullptr;
    if (getenv("SAGEN_NO_DECODER_PLANES") != nullptr) c->dec_planes_min_batch = 1 << 30;
    // with two fp16 planes a plane pass writes 4 bytes per element - what the fp32 pass it replaces writes - so the planes pay from
    // stage 2 on (measured, same box: 2 034 against 1 943 ambisonic-s/s); with three bf16 planes (6 bytes) only from stage 3
    if (getenv("SAGEN_P3_FROM_STAGE") == nullptr) c->p3_from_stage = (c->use_h2 && c->use_p3) ? 2 : 3;
    // conv3g_kernel for the stride-2 conv_1 + shortcut of the first block of stages 3-5 (the previous stage's last merge then writes its
    // output as planes ONLY).  Measured per layer, batch 32 (profiles/r04_layers_p3g*.txt): on three bf16 planes no faster than
    // igemm3_kernel (56 / 58 / 72 us against 57 / 57 / 66) - on two fp16 planes 43 / 41 / 48 us against 60 / 61 / 74, shortcuts
    // 24 / 14 / 13 against 25 / 19 / 18: on with the fp16x2 planes, opt-in (SAGEN_P3G=1) without them
    c->use_p3g = c->use_p3 && (getenv("SAGEN_P3G") != nullptr || (c->use_h2 && getenv("SAGEN_NO_P3G") == nullptr));
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
    size_t h2_pack_blocks = 0;
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
        if (vs.name.find("/deconv") != std::string::npos) {     // deconv5 .. deconv2 also in scatter form: [(p, q, o)][c] (Fwd::deconv_scatter)
            const int l = vs.name[vs.name.find("/deconv") + 7] - '1';
            if (l >= 1 && vs.shape[3] % 16 == 0) {
                const size_t ns = packed_floats((long)vs.shape[0] * vs.shape[1] * vs.shape[2], vs.shape[3]);
                c->alloc("pks:" + vs.name, packed_split_floats(ns));
                c->alloc("pkhs:" + vs.name, ns);                   // ... and as two fp16 planes (conv3g_kernel on planes of the concat buffer)
                h2_pack_blocks += ns / 1024 + 1;
                c->h2_slot["pks:" + vs.name.substr(0, vs.name.size() - 8)] = -1;     // (slot numbers are dealt below)
            }
        }
        // the 3x3 convs (and 1x1 projections) of the ResNet trunks also as two fp16 planes of w * 2^kw (conv3h.hip): N * Kpad * 2 halves
        // ... and deconv1 of the mask decoder (conv3g_kernel with the fused decoder tail on planes of cat1: sagen_forward_impl)
        if ((vs.ndim == 4 && ((vs.shape[0] == 3 && vs.shape[1] == 3) || (vs.shape[0] == 1 && vs.shape[1] == 1)) && vs.shape[2] % 16 == 0 &&
             vs.name.find("_encoder/conv") != std::string::npos) || vs.name == "separation/deconv1/weights" ||
            (vs.ndim == 4 && vs.shape[2] == 3 && vs.name.find("_encoder/conv1/conv/weights") != std::string::npos) ||      // ... and the stem, for float frames (stem8.hip, F16)
            (vs.ndim == 4 && vs.shape[2] % 16 == 0 && vs.name.compare(0, 18, "audio_encoder/conv") == 0)) {                 // ... and conv2 .. conv5 of the audio encoder (round 6: conv3g_kernel on planes of cat_l's encoder half)
            c->alloc("pkh:" + vs.name, n);
            h2_pack_blocks += n / 1024 + 1;
            c->h2_slot[vs.name.substr(0, vs.name.size() - 8)] = -1;
        }
    }
    {   // 2^-kw slots of the fp16x2 filter planes, h2s[8 ..]
        int slot = H2S_FILTER_FIRST;
        for (auto& kv : c->h2_slot) kv.second = slot++;
        static_assert(H2S_A_INV_X + 4 <= H2S_A_INV_B && H2S_A_INV_B + 2 <= H2S_S16_A_INV && H2S_S16_A_INV + 2 <= H2_RIG_OFF &&
                      2 + H2_RIG_OFF + 4 <= 256, "h2s: the fixed slots overlap");
        if (slot > H2S_FIXED_FIRST) return fail(SAGEN_ERR_UNSUPPORTED, "too many fp16x2 layers for the scale table");
    }
    c->alloc("h2:jobs", (c->h2_slot.size() + 1) * sizeof(H2Job) / sizeof(float) + 64);
    c->alloc("h2:amax", c->h2_slot.size() + h2_pack_blocks + 64);      // per-job maxima + per-workgroup partials
    c->alloc("pk:jobs", (c->vars.size() + 1) * sizeof(PackJob) / sizeof(float) + 64);      // device copy of the pack-job table
    // ---- everything from here on is PER BATCH: a grouped context (sagen_create_grouped) holds this region once per group ----
    c->grp_off = c->ws_floats;
    c->alloc("h2s", 256);                  // fp16x2 scales: [0], [1] = 2^-ka of the planes in the video / flow trunk's plane buffer, [2..5] block-input bounds, [6] scratch, [7] saturation counter, [8..] 2^-kw per layer, [2 + H2_RIG_OFF ..] the rigorous bounds behind [2..5] (p3.hip)
    // activations
    c->alloc("mag", (size_t)B * 127 * 1024);
    c->alloc("spec", (size_t)B * 28 * 513 * 2);
    // concat buffers cat_l: [B, H_l, W_l, C_dec + C_enc]; cat5 = [conv5 | fc-feats]
    for (int l = 1; l <= 5; ++l)
        c->alloc("cat" + std::to_string(l), (size_t)B * c->enc_h[l] * c->enc_w[l] * 2 * c->enc_c[l]);
    c->alloc("bott", (size_t)B * 3 * c->Cb);
    for (int i = 0; i < cfg->n_loc_units; ++i) c->alloc("loc" + std::to_string(i + 1), (size_t)B * 3 * cfg->loc_units[i]);
    c->alloc("coeffs", (size_t)B * 3 * 3 * (c->nsep + 1));
    {   // fcm_kernel (fcm.hip) for the skinny FC layers at inference: partial buffers [slices][rows][N] per layer
        struct L { std::string name; int rows, K, N; };
        std::vector<L> ls;
        ls.push_back({"bottleneck/audio-fc", 3 * B, c->enc_w[5] * c->enc_c[5], 1024});
        if (c->has_video) ls.push_back({"bottleneck/video-fc", B, 7 * 14 * 128, 512});
        if (c->has_flow) ls.push_back({"bottleneck/flow-fc", B, 7 * 14 * 128, 512});
        int fin = c->Cb;
        for (int i = 0; i < cfg->n_loc_units; ++i) { ls.push_back({"localization/fc" + std::to_string(i + 1), 3 * B, fin, cfg->loc_units[i]}); fin = cfg->loc_units[i]; }
        ls.push_back({"localization/fc" + std::to_string(cfg->n_loc_units + 1), 3 * B, fin, 3 * (c->nsep + 1)});
        if (c->freq_mask) ls.push_back({"separation/fc-feats", 3 * B, c->Cb, 512});
        // OPT-IN (SAGEN_FCM=1): measured slower than the general kernels it replaces (audio-fc 33 vs 15 us, video-fc 22 vs 18, fc2 / fc3 17 vs
        // 9 - 10; only fc1 + fc-feats in one launch wins, 19 vs 23): these layers are bound by the latency of a short dependent launch
        // (~6 us, what a reducer takes), not by how their arithmetic is fed
        bool ok = getenv("SAGEN_FCM") != nullptr && !c->sk_fused && 3 * B <= FCM_MAX_M;
        for (const L& l : ls) ok = ok && fcm_pick_slices(l.K, (long)l.K * l.N) > 0;
        c->use_fcm = ok;
        if (ok)
            for (const L& l : ls) {
                int ns = fcm_pick_slices(l.K, (long)l.K * l.N);
                if (l.name == "separation/fc-feats") ns = c->fcm_slices.at("localization/fc1");      // (one launch with fc1: the same slicing)
                c->fcm_slices[l.name] = ns;
                c->alloc("fcm:" + l.name, (size_t)ns * l.rows * l.N);
            }
    }
    c->alloc("splitk", std::max<size_t>((size_t)16 << 20, (size_t)B * 56 * 112 * 64 * 2));   // fp32 split-K partials (up to 2 splits of the largest conv), checked per use
    if (c->freq_mask) {
        // rows 10..16 of cat1 as fp16x2 planes (the operand of deconv1 on conv3g_kernel) + the words around them: H2_AMAX_FLOATS floats (H2_AMAX_SLOTS lines) each for the exact
        // max |y| of cat1's encoder / decoder half (published by the epilogues of conv1 / deconv2's gather), then 2^-ka of the planes
        c->alloc("cat1p", (p3h_bytes(B, 7, c->enc_w[1], 2 * c->enc_c[1]) + 3) / 4 + 64);
        {   // planes of the band of cat_(l+1) the scatter-form deconv(l+1) contracts (one buffer: pack -> contraction, layer after layer)
            size_t mx = 0;
            for (int l = 2; l <= 5; ++l) mx = std::max(mx, p3h_bytes(B, c->enc_h[l], c->enc_w[l], 2 * c->enc_c[l]));
            c->alloc("catp", (mx + 3) / 4 + 64);
        }
        {   // round 6: the encoder half of cat_l (l = 1..4) as fp16x2 planes, the operand of the audio encoder's conv(l+1) on conv3g_kernel
            size_t mx = 0;
            for (int l = 1; l <= 4; ++l) mx = std::max(mx, p3h_bytes(B, c->enc_h[l], c->enc_w[l], c->enc_c[l]));
            c->alloc("aencp", (mx + 3) / 4 + 64);
        }
        c->alloc("amax", 10 * H2_AMAX_FLOATS + 64);       // [cat_l: l = 1..5][decoder half, encoder half][H2_AMAX_FLOATS], then 2^-ka of the planes of cat_l at [.. + l]
        c->alloc("dmask", (size_t)B * 23 * 1024 * c->nsep);
        c->alloc("frames", mask_istft_scratch_bytes(B) / sizeof(float));
    }
    c->alloc("splitk_aux", (size_t)8 << 20);          // split-K scratch of the second stream (audio chain / flow FCs)
    c->alloc("splitk:tk", SK_TICKETS);                // per-tile tickets of the in-launch split-K combine (zero between launches)
    c->alloc("splitk_aux:tk", SK_TICKETS);
    for (int set = 0; set < 2; ++set) {
        const std::string x = set ? "_b" : "";
        if (set == 0 ? !(c->has_video || c->has_flow) : !(c->has_video && c->has_flow)) continue;   // "_b": flow trunk next to the video trunk
        c->alloc("xpad" + x, (size_t)B * 229 * 456 * 4);      // zero-bordered 4-channel fp32 frame [229][454][4] | one bf16 plane [229][456][4] (uint8 frames) | two fp16 planes (float frames)
        c->alloc("s16:part" + x, STEM16_PARTS + 64);
        c->alloc("y0" + x, (size_t)B * 112 * 224 * 64);
        const size_t stage = (size_t)B * 56 * 112 * 64;
        for (const char* nm : {"rx0", "rx1", "ry1", "ry2", "rsc", "ry1n"}) c->alloc(nm + x, stage);
        c->alloc("p3" + x, (p3_bytes(B, 56, 112, 64) + 3) / 4 + 64);   // bf16 planes of the current 3x3 conv input (largest: stage 2)
        c->alloc("p3b" + x, (p3h_bytes(B, 56, 112, 64) + 3) / 4 + 64); // lean trunk (model.h: resnet): the conv_2 input planes, "p3" then holds block inputs / outputs
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
    // SAGEN_AUX_PRIO=low|high: the second stream below / above the default priority (experiment: the dispatcher then prefers one branch)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);       // (least, greatest): numerically lower = higher priority
    const char* aux_prio = getenv("SAGEN_AUX_PRIO");
    const int prio = aux_prio && aux_prio[0] == 'l' ? prio_lo : (aux_prio && aux_prio[0] == 'h' ? prio_hi : 0);
    if ((aux_prio ? hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stft, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->aux = nullptr;      // no device / no stream: forward falls back to a single stream (and fails at launch)
    }
    c->grp_floats = c->ws_floats - c->grp_off;          // (a multiple of 64 floats: every buffer is rounded up to 256 bytes)
    c->ws_floats = c->grp_off + (size_t)c->G * c->grp_floats;
    if (c->G > 1 && (c->use_fcm || c->sk_fused || c->fp32_only || !c->use_h2 || !c->use_p3 || !c->use_p3g || c->p3_from_stage > 2 || c->no_lean_trunk)) {
        sagen_destroy_impl(c);
        return fail(SAGEN_ERR_UNSUPPORTED, "sagen_create_grouped: the grouped launch runs the default kernels only (an environment switch selected others)");
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
    c->pack_jobs.clear();                 // (the table holds the variables' addresses)
    c->h2_jobs.clear();
    for (int g = 0; g < c->G; ++g) {
        const size_t go = (size_t)g * c->grp_floats;
        SAGEN_HIP_CHECK(hipMemsetAsync(c->p("h2s") + go, 0, 256 * sizeof(float), s));       // fp16x2 scales, bounds, saturation counter
        SAGEN_HIP_CHECK(hipMemsetAsync(c->p("splitk:tk") + go, 0, SK_TICKETS * sizeof(int), s));
        SAGEN_HIP_CHECK(hipMemsetAsync(c->p("splitk_aux:tk") + go, 0, SK_TICKETS * sizeof(int), s));
    }
    int rc = fft_tables_ensure(s);
    if (rc) return rc;
    rc = sagen_repack_impl(c, s);
    if (rc) return rc;
    // the filters' 2^-kw (written by the fp16x2 pack into group 0's scale table) are the same for every group
    for (int g = 1; g < c->G; ++g)
        SAGEN_HIP_CHECK(hipMemcpyAsync(c->p("h2s") + (size_t)g * c->grp_floats + H2S_FILTER_FIRST, c->p("h2s") + H2S_FILTER_FIRST,
                                       (H2S_FIXED_FIRST - H2S_FILTER_FIRST) * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->bound = true;
    return SAGEN_OK;
}

// repack every filter from the bound variables into the kernels' layouts (bind; and once per training step, after the optimiser
// has updated the variables in place)
// The pack of one "/weights" variable as a job of the batched launch (same layouts as the single-variable launchers of igemm.hip)
static PackJob forward_pack_job(const sagen_ctx* c, const VarSpec& vs) {
    PackJob j;
    j.src = c->v(vs.name);
    j.dst = c->p("pk:" + vs.name);
    if (vs.name.find("/deconv") != std::string::npos) {
        const int l = vs.name[vs.name.find("/deconv") + 7] - '1';
        const int sh = AENC_S[l][0], sw = AENC_S[l][1];
        const int nth = cdiv(vs.shape[0], sh), ntw = cdiv(vs.shape[1], sw);
        j.kind = PACK_DECONV;
        j.N = sh * sw * (int)vs.shape[2];
        j.Kpad = (nth * ntw * (int)vs.shape[3] + 15) / 16 * 16;
        j.p[0] = (int)vs.shape[0]; j.p[1] = (int)vs.shape[1]; j.p[2] = (int)vs.shape[2]; j.p[3] = (int)vs.shape[3]; j.p[4] = sh; j.p[5] = sw; j.p[6] = ntw;
    } else if (vs.ndim == 4) {
        int cin = (int)vs.shape[2], cinp = cin == 3 ? 4 : cin, taps = (int)(vs.shape[0] * vs.shape[1]);
        if (cin == 1) { cin = cinp = (int)vs.shape[1]; taps = (int)vs.shape[0]; }
        const bool stem = cin == 3;                   // tap rows padded 7 -> 8, K = (dh, dw8, c4)
        if (stem) taps = (int)(vs.shape[0] * (vs.shape[1] + 1));
        j.kind = PACK_CONV;
        j.N = (int)vs.shape[3];
        j.Kpad = (taps * cinp + 15) / 16 * 16;
        j.p[0] = taps; j.p[1] = cin; j.p[2] = cinp; j.p[3] = (int)vs.shape[3];
        j.p[4] = stem ? (int)vs.shape[1] : 0; j.p[5] = stem ? (int)vs.shape[1] + 1 : 0;
    } else {
        const int K = (int)vs.shape[0], N = (int)vs.shape[1];
        j.kind = PACK_CONV;
        j.N = N; j.Kpad = (K + 15) / 16 * 16;
        j.p[0] = 1; j.p[1] = K; j.p[2] = K; j.p[3] = N;
    }
    return j;
}

// uploads a job table into `dev` (capacity checked by the caller) and returns the number of blocks of the launch
int sagen_upload_pack_jobs(std::vector<PackJob>& jobs, void* dev, hipStream_t s) {
    int nb = 0;
    for (auto& j : jobs) { j.first_block = nb; nb += pack_job_blocks(j.kind, j.N, j.Kpad); }
    if (!jobs.empty() && hipMemcpyAsync(dev, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return nb;
}

// part 0: everything; 1: only the stems' packs (first in the table); 2: the rest + the fp16x2 planes (the training step runs 1 on
// the caller's stream and 2 on the second stream, under the stem)
int sagen_repack_part(sagen_ctx* c, hipStream_t s, int part) {
    if (c->pack_jobs.empty()) {          // first call after bind: build the table from the bound variables and ship it
        for (int pass = 0; pass < 2; ++pass) {
            for (const auto& vs : c->vars) {
                if (vs.name.size() < 8 || vs.name.compare(vs.name.size() - 8, 8, "/weights") != 0) continue;
                const bool stem = vs.name.find("/conv1/conv/") != std::string::npos;
                if (stem == (pass == 0)) c->pack_jobs.push_back(forward_pack_job(c, vs));
                if (pass == 1 && c->bufs.count("pks:" + vs.name)) {      // the scatter-form pack of a stride-1 transposed conv: rows (p, q, o), columns c
                    PackJob j;
                    j.src = c->v(vs.name); j.dst = c->p("pks:" + vs.name);
                    j.kind = PACK_ROWS;
                    j.N = (int)(vs.shape[0] * vs.shape[1] * vs.shape[2]); j.Kpad = (int)vs.shape[3];
                    j.p[0] = j.N; j.p[1] = (int)vs.shape[3];
                    c->pack_jobs.push_back(j);
                }
            }
            if (pass == 0) c->pack_early_jobs = (int)c->pack_jobs.size();
        }
        if (c->pack_jobs.size() * sizeof(PackJob) > c->bufs.at("pk:jobs").n * sizeof(float)) return fail(SAGEN_ERR_WORKSPACE, "pack job table too small");
        c->pack_blocks = sagen_upload_pack_jobs(c->pack_jobs, c->p("pk:jobs"), s);
        if (c->pack_blocks < 0) return fail(SAGEN_ERR_HIP, "pack job upload failed");
        c->pack_early_blocks = c->pack_early_jobs < (int)c->pack_jobs.size() ? c->pack_jobs[c->pack_early_jobs].first_block : c->pack_blocks;
    }
    const PackJob* jobs = reinterpret_cast<const PackJob*>(c->p("pk:jobs"));
    const int ne = c->pack_early_jobs, be = c->pack_early_blocks, n = (int)c->pack_jobs.size();
    int rc = SAGEN_OK;
    if (part == 1) return ne > 0 ? pack_multi_launch(jobs, ne, be, s) : SAGEN_OK;
    if (part == 2) rc = n > ne ? pack_multi_launch(jobs + ne, n - ne, c->pack_blocks - be, s, be) : SAGEN_OK;
    else rc = pack_multi_launch(jobs, n, c->pack_blocks, s);
    if (rc || c->fp32_only || !c->use_p3) return rc;
    // the fp16x2 filter planes of the trunks' 3x3 convs and 1x1 projections, from the fp32 packs just written: two launches for all
    // of them (bind; every training step)
    if (c->h2_jobs.empty()) {
        int nb = 0;
        for (const auto& kv : c->h2_slot) {
            const bool scatter = kv.first.compare(0, 4, "pks:") == 0;       // the scatter-form pack of a transposed conv: rows (p, q, o), columns c
            const std::string vname = (scatter ? kv.first.substr(4) : kv.first) + "/weights";
            const VarSpec* vs = nullptr;
            for (const auto& v : c->vars) if (v.name == vname) vs = &v;
            if (!vs) continue;
            H2Job j;
            if (scatter) {
                j.N = (int)(vs->shape[0] * vs->shape[1] * vs->shape[2]); j.Kpad = (int)vs->shape[3];
                j.wp = c->p("pks:" + vs->name); j.w2 = c->p("pkhs:" + vs->name); j.w_inv = c->p("h2s") + kv.second;
                j.first_block = nb;
                nb += (int)(((long)j.N * j.Kpad + 1023) / 1024);
                c->h2_jobs.push_back(j);
                continue;
            }
            j.N = (int)vs->shape[3]; j.Kpad = (int)(vs->shape[0] * vs->shape[1] * vs->shape[2]);
            if (vs->shape[2] == 3) j.Kpad = (int)(vs->shape[0] * (vs->shape[1] + 1) * 4);      // the stem's pack: tap rows padded 7 -> 8, channels 3 -> 4
            if (vs->name.find("/deconv") != std::string::npos) {      // depth-to-space pack: N = (ry, rx, o), K = (dp, dq, c)
                const int l = vs->name[vs->name.find("/deconv") + 7] - '1';
                j.N = AENC_S[l][0] * AENC_S[l][1] * (int)vs->shape[2];
                j.Kpad = cdiv(vs->shape[0], AENC_S[l][0]) * cdiv(vs->shape[1], AENC_S[l][1]) * (int)vs->shape[3];
            }
            j.wp = c->p("pk:" + vs->name); j.w2 = c->p("pkh:" + vs->name); j.w_inv = c->p("h2s") + kv.second;
            j.first_block = nb;
            nb += (int)(((long)j.N * j.Kpad + 1023) / 1024);
            c->h2_jobs.push_back(j);
        }
        c->h2_blocks = nb;
        if (c->h2_jobs.size() * sizeof(H2Job) > c->bufs.at("h2:jobs").n * sizeof(float)) return fail(SAGEN_ERR_WORKSPACE, "fp16x2 job table too small");
        if (!c->h2_jobs.empty() && hipMemcpyAsync(c->p("h2:jobs"), c->h2_jobs.data(), c->h2_jobs.size() * sizeof(H2Job), hipMemcpyHostToDevice, s) != hipSuccess)
            return fail(SAGEN_ERR_HIP, "fp16x2 job upload failed");
    }
    return h2_filter_pack_multi_launch(reinterpret_cast<const H2Job*>(c->p("h2:jobs")), (int)c->h2_jobs.size(), c->h2_blocks,
                                       reinterpret_cast<unsigned*>(c->p("h2:amax")), s);
}
int sagen_repack_impl(sagen_ctx* c, hipStream_t s) { return sagen_repack_part(c, s, 0); }
// does sagen_forward_impl fork onto the context's second stream?  (the training step's split repack relies on it)
bool sagen_forward_forks(const sagen_ctx* c) {
    static const bool one_stream = getenv("SAGEN_ONE_STREAM") != nullptr;
    return c->aux && !c->tuning && !one_stream && (c->has_video || c->has_flow);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------

int sagen_forward_impl(sagen_ctx* c, const float* audio, const float* video, const float* flow, float* out, hipStream_t s) {
    if (!c || !audio || !out) return fail(SAGEN_ERR_NULL, "sagen_forward: null argument");
    if (!c->bound) return fail(SAGEN_ERR_WEIGHTS, "sagen_forward: weights are not bound");
    if (c->has_video && !video) return fail(SAGEN_ERR_NULL, "sagen_forward: video encoder enabled but video is NULL");
    if (c->has_flow && !flow) return fail(SAGEN_ERR_NULL, "sagen_forward: flow encoder enabled but flow is NULL");
    const int B = c->B;
    c->events_used = 0;
    c->prof.clear();
    // grouped context: every launch of this call carries the group dimension (common.h); the launchers read it from cur_group()
    struct GroupScope {
        GroupInfo saved;
        explicit GroupScope(const GroupInfo& gi) : saved(cur_group()) { cur_group() = gi; }
        ~GroupScope() { cur_group() = saved; }
    } group_scope(c->group_info());
    if (c->G > 1) {
        if (c->train_mode) return fail(SAGEN_ERR_UNSUPPORTED, "the training step has no grouped launch");
        if (!c->freq_mask || c->fp32_only || !c->use_h2 || !c->use_p3 || !c->use_p3g || c->p3_from_stage > 2 || c->no_lean_trunk || c->use_fcm || c->sk_fused ||
            (c->has_video && !c->video_u8 && !c->stem16) || (c->has_video && c->video_u8 && !c->stem8) || (c->has_flow && !c->stem16))
            return fail(SAGEN_ERR_UNSUPPORTED, "grouped forward: an option moved the path off the kernels that take a group dimension "
                        "(fp16x2 / planes_from_stage / plane_gather / u8_fast_stem / f16_fast_stem must keep their defaults)");
    }

    // Two launch streams: `f` (the caller's) carries the video trunk and everything after the bottleneck; `g`
    // (context-owned) carries the independent audio chain and, with three encoders, the flow trunk.  They fork at
    // entry and join before the localisation FCs.  While autotuning, or without visual encoders, everything
    // stays on the caller's stream.
    const bool forked = sagen_forward_forks(c);
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

    // deconv1 on fp16x2 planes of cat1 (conv3g_kernel + fused decoder tail): the planes' scale is the EXACT maximum of cat1, published by
    // the epilogues of its two producers (conv1: encoder half, deconv2: decoder half) into words zeroed here
    const bool no_d1p = c->no_d1_planes;
    // (the opt-in in-launch split-K combine, SAGEN_SK_FUSED=1, has no reducer to publish a maximum from: it keeps the round-4 decoder)
    bool d1_planes = !no_d1p && !c->sk_fused && c->freq_mask && !c->train_mode && !c->materialize_mask && !c->fp32_only && c->nsep == 32 && f.h2() &&
                     c->bufs.count("cat1p") != 0 && c->h2_slot.count("separation/deconv1") != 0 && getenv("SAGEN_NO_MASKFUSE") == nullptr;
    if (d1_planes && !c->tuning) {             // a plan that names a register-staged tile for deconv1 keeps the round-4 path (no pack pass)
        auto it = c->plan.find("separation/deconv1");
        if (it != c->plan.end() && !igemm_tile_p3((IgemmTile)it->second.tile)) d1_planes = false;
    }
    // ... and deconv5 .. deconv2 (scatter form) on planes of the band of cat_(l+1) they contract, the same way
    const bool no_decp = c->dec_planes_min_batch > (1 << 29);
    // (from 16 windows on: at deploy.py's batch of 10 the four pack launches cost what the plane-fed GEMMs save - 6 450 against 6 630 ambisonic-s/s)
    // (the training step's forward too, on its live bands - its backward reads the fp32 concat buffers the gather passes write either way;
    //  SAGEN_TRAIN_NO_DEC_PLANES=1: the register-staged kernels on fp32 operands, as up to round 5)
    static const bool train_no_decp = getenv("SAGEN_TRAIN_NO_DEC_PLANES") != nullptr;
    const bool dec_planes = !no_decp && c->B >= c->dec_planes_min_batch && !c->sk_fused && c->freq_mask && !(c->train_mode && (train_no_decp || !c->train_bands)) &&
                            !c->fp32_only && f.h2() && c->bufs.count("catp") != 0 && !c->no_scatter;
    const bool want_amax = d1_planes || dec_planes;
    // (the words are cleared by the STFT launch below - the first kernel of the stream that carries the audio chain - not by a fill)

    // ---- stream g: STFT (myutils.py:119-147) -> |.| of frames 46:173 (model.py:166-178) + spectrum of frames 89:117
    g.layer = "stft";
    g.timed("stft_kernel", 0.0, [&] { return stft_launch(audio, B, c->snd_size, 46, 173, c->p("mag"), 89, 117, c->p("spec"), g.s,
                                                         want_amax ? c->p("amax") : nullptr, 10 * H2_AMAX_FLOATS + 64); });
    if (forked) {
        // The LDS FFT kernels give wrong results when bf16x3 contraction waves of ANOTHER stream share their CUs
        // (open): the first matrix launch of the main stream waits for the STFT (it overlaps the pad kernel).
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_stft, c->aux));
        f.wait_before_mfma = c->ev_stft;
    }
    if (c->late_repack) {            // training step: every filter pack but the stems', here - under the stem of the main stream
        c->late_repack = false;
        if (!forked) return bail(fail(SAGEN_ERR_UNSUPPORTED, "late repack without a second stream"));
        const int rc = sagen_repack_part(c, c->aux, 2);
        if (rc) return bail(rc);
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_pack, c->aux));
        f.wait_packs = c->ev_pack;
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
        if (want_amax) d.amax_out = g.cat_amax(l + 1, 1);        // max |conv(l+1)| = the encoder half of cat_(l+1)
        g.layer = name;
        // Round 6: conv2 .. conv5 on fp16x2 planes of cat_l's encoder half (conv3g_kernel: three products per multiply, no operand split in
        // the K loop) - the planes' scale is that half's EXACT maximum, published by conv_l's epilogue (as for the decoder, round 5).  With
        // several batches per launch these layers were the register-staged family's largest share (40 - 76 TFLOP/s: 11 % of a grouped call)
        bool aud_planes = l >= 1 && want_amax && !c->no_aud_planes && f.h2() && c->bufs.count("aencp") != 0 && c->h2_slot.count(name) != 0 &&
                          conv3g_ok_desc(d, c->enc_c[l]);
        if (aud_planes && !c->tuning) {             // a plan that names a register-staged tile keeps the fp32 operand (no pack pass)
            auto it = c->plan.find(name);
            if (it != c->plan.end() ? !igemm_tile_p3((IgemmTile)it->second.tile)
                                    : B < 128)             // untuned: the plane-fed kernel cannot split K and needs many rows to fill the chip - decided by the
                                                           // batch size alone, NOT by the groups, so that a grouped and an ungrouped context without a plan compute bit-identically
                aud_planes = false;
        }
        if (aud_planes) {
            const int cp = c->enc_c[l], Hl = c->enc_h[l], Wl = c->enc_w[l];
            float* const a_inv = c->p("amax") + (size_t)10 * H2_AMAX_FLOATS + 8 + l;
            g.layer = name + "/planes";
            g.timed("h2_pack_rows_kernel", 0.0, [&] {
                return h2_pack_rows_launch(c->p("cat" + std::to_string(l)) + cp, (long)Hl * Wl * 2 * cp, (long)Wl * 2 * cp, 2 * cp, 0, B, Hl, Wl, cp,
                                           g.cat_amax(l, 1), nullptr, c->p("aencp"), a_inv, reinterpret_cast<unsigned*>(c->p("h2s") + 7), g.s); });
            g.layer = name;
            d.xp3 = c->p("aencp"); d.xp3_fmt = 1; d.xp3_row0 = 0; d.xp3_rows = 0;
            d.p3_np = B * Hl * (Wl + 1);
            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
            d.xp3_bytes = (unsigned)((size_t)d.xp3_cstride * (cp / 16));
            d.wh2 = c->p("pkh:" + name + "/weights"); d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
            d.h2_a_inv = a_inv; d.h2_w_inv = c->p("h2s") + c->h2_slot.at(name);
        }
        g.gemm(d);
    }

    // bottleneck (model.py:203-239)
    float* bott = c->p("bott");
    const bool fcm = c->use_fcm && !c->train_mode && !c->tuning;
    // one fcm launch: `segs` describe the input rows, `names` the layers (their variables, their "fcm:" partial buffers)
    auto fcm_layer = [&](Fwd& w, const std::string& label, int M, int K, const std::vector<FcmSeg>& segs, const std::vector<std::string>& names, const std::vector<int>& Ns) {
        FcmDesc fd;
        fd.M = M; fd.K = K; fd.nseg = (int)segs.size(); fd.njobs = (int)names.size(); fd.nslices = c->fcm_slices.at(names[0]);
        for (size_t q = 0; q < segs.size(); ++q) fd.seg[q] = segs[q];
        double fl = 0.0;
        for (size_t j = 0; j < names.size(); ++j) {
            fd.job[j].w = c->v(names[j] + "/weights"); fd.job[j].N = Ns[j]; fd.job[j].out = c->p("fcm:" + names[j]);
            fl += 2.0 * M * K * Ns[j];
        }
        w.layer = label;
        w.timed(M <= 32 ? "fcm_kernel<1>" : (M <= 64 ? "fcm_kernel<2>" : "fcm_kernel<3>"), fl, [&] { return fcm_launch(fd, w.s); });
    };
    // ... and the reducer behind it: y[(m * rep + r) * ldy + n] = act(bias + sum of the slices)
    auto fcm_finish = [&](Fwd& w, const std::string& name, int rows, int N, bool relu, float* y, int ldy, int rep, float* amax_out = nullptr) {
        w.layer = name;
        w.timed("splitk_reduce_kernel", 0.0, [&] {
            return splitk_reduce_launch(c->p("fcm:" + name), c->fcm_slices.at(name), rows, N, c->v(name + "/biases"), relu ? 1 : 0, y, ldy, rep, nullptr, w.s, amax_out); });
    };
    if (fcm) {  // audio-fc: the six frequency columns of conv5 (the first 512 channels of cat5's pixels) are six plain input ranges
        std::vector<FcmSeg> segs(6);
        for (int wq = 0; wq < 6; ++wq) {
            segs[wq].p = c->p("cat5") + (size_t)wq * 1024; segs[wq].k0 = wq * 512; segs[wq].k1 = (wq + 1) * 512; segs[wq].ld = 6 * 1024;
        }
        fcm_layer(g, "bottleneck/audio-fc", B * 3, 6 * 512, segs, {"bottleneck/audio-fc"}, {1024});
        fcm_finish(g, "bottleneck/audio-fc", B * 3, 1024, true, bott, c->Cb, 1);
    } else {   // audio-fc over (w, c) of conv5: 6 taps along W, 512 channels each
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
        const float* feat = c->train_mode ? w.resnet_train(e == 0 ? video : flow, enc + "_encoder")      // (retains every activation)
                                          : w.resnet(e == 0 ? video : flow, enc + "_encoder");          // [B,7,14,512]
        w.fc(feat, B * 98, 512, 512, "bottleneck/" + enc + "-fc-red", 128, true, c->p("fcred" + w.sfx), 128);
        if (fcm) {
            FcmSeg sg;
            sg.p = c->p("fcred" + w.sfx); sg.k0 = 0; sg.k1 = 98 * 128; sg.ld = 98 * 128;
            fcm_layer(w, "bottleneck/" + enc + "-fc", B, 98 * 128, {sg}, {"bottleneck/" + enc + "-fc"}, {512});
            fcm_finish(w, "bottleneck/" + enc + "-fc", B, 512, true, bott + choff, c->Cb, 3);      // tile x3 (model.py:230-232)
        } else {
            w.fc(c->p("fcred" + w.sfx), B, 98 * 128, 98 * 128, "bottleneck/" + enc + "-fc", 512, true, bott + choff, c->Cb, 3);   // tile x3 (model.py:230-232)
        }
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
    const int nlast_fcm = 3 * (c->nsep + 1);
    if (fcm) {
        // fc1 and fc-feats read the same rows (the bottleneck): one launch; the feats tile (cat5, six frequency columns: the decoder's
        // input) is finished on the caller's stream, then the streams fork: fc1's reducer, fc2, fc3 run next to the decoder.
        const int M3 = B * 3, nl = c->cfg.n_loc_units;
        auto plain = [&](const float* x, int K) { FcmSeg sg; sg.p = x; sg.k0 = 0; sg.k1 = K; sg.ld = K; return sg; };
        std::vector<std::string> names{"localization/fc1"};
        std::vector<int> Ns{nl > 0 ? c->cfg.loc_units[0] : nlast_fcm};
        if (c->freq_mask) { names.push_back("separation/fc-feats"); Ns.push_back(512); }
        fcm_layer(f, c->freq_mask ? "localization/fc1+separation/fc-feats" : "localization/fc1", M3, c->Cb, {plain(bott, c->Cb)}, names, Ns);
        if (c->freq_mask) fcm_finish(f, "separation/fc-feats", M3, 512, true, c->p("cat5") + 512, 1024, 6, want_amax ? f.cat_amax(5, 0) : nullptr);
        if (fork2) {
            SAGEN_HIP_CHECK(hipEventRecord(c->ev_fork, s));
            SAGEN_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        }
        Fwd& w = fork2 ? g : f;
        int K = Ns[0];
        for (int i = 1; i <= nl; ++i) {
            const std::string prev = "localization/fc" + std::to_string(i), cur = "localization/fc" + std::to_string(i + 1);
            const int N = i < nl ? c->cfg.loc_units[i] : nlast_fcm;
            float* y = c->p("loc" + std::to_string(i));
            fcm_finish(w, prev, M3, K, true, y, K, 1);
            fcm_layer(w, cur, M3, K, {plain(y, K)}, {cur}, {N});
            K = N;
        }
        fcm_finish(w, "localization/fc" + std::to_string(nl + 1), M3, nlast_fcm, false, c->p("coeffs"), nlast_fcm, 1);
        if (f.rc || g.rc) return bail(f.rc ? f.rc : g.rc);
    } else {
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
    }

    if (!c->freq_mask) {
        f.layer = "decoder";
        f.timed("nosep_mix_kernel", 0.0, [&] { return nosep_mix_launch(audio, c->p("coeffs"), out, B, c->snd_size, c->snd_contx, c->snd_dur, 3, s); });
        return f.rc;
    }

    // separation (model.py:282-348)
    if (!fcm)
        f.fc(bott, B * 3, c->Cb, c->Cb, "separation/fc-feats", 512, true, c->p("cat5") + 512, 1024, 6,     // tile over 6 freq columns
             want_amax ? f.cat_amax(5, 0) : nullptr);
    // Inference runs deconv5 .. deconv2 in SCATTER form (Fwd::deconv_scatter): every input pixel is contracted once against the whole
    // filter and a gather pass assembles the output - the conv form over the output grid multiplies padding for most taps of the
    // stride-1 layers (5.4 of 15 taps of deconv5 land inside its 3x6 input), and the strided layers' depth-to-space form cannot split K.
    // Only rows 10..16 of cat1 reach deconv1's live grid rows (11..16, two vertical taps), hence only rows 4..8 of cat2 (deconv2's
    // taps) and rows 1..4 of cat3: each layer contracts the band of input rows it needs and gathers the output rows that are read.
    // SAGEN_FP32_ONLY keeps the round-4 form on the full tensors.  The training step follows the same bands, forward AND backward
    // (train_model.hip: the dead rows carry no gradient; the buffers behind them are zeroed once, at sagen_train_bind).
    const bool no_scatter = c->no_scatter;
    const bool lean = (!c->train_mode || c->train_bands) && !c->fp32_only && !no_scatter;
    int need_lo[7], need_hi[7];                          // rows of cat_l that the layer below reads
    for (int l = 1; l <= 5; ++l) { need_lo[l] = 0; need_hi[l] = c->enc_h[l]; }
    if (lean) {
        need_lo[1] = 10; need_hi[1] = 17;
        for (int l = 1; l <= 4; ++l) {                   // deconv(l+1) writes rows [need_lo[l], need_hi[l]) of cat_l from these rows of cat_(l+1)
            const int sh = AENC_S[l][0], kh = AENC_K[l][0];
            need_lo[l + 1] = std::max((need_lo[l] - (kh - 1) + sh - 1) / sh, 0);            // smallest y with y*sh + kh - 1 >= need_lo
            need_hi[l + 1] = std::min((need_hi[l] - 1) / sh + 1, c->enc_h[l + 1]);           // largest y with y*sh <= need_hi - 1
        }
    }
    for (int l = 1; l <= 5; ++l) { c->dec_lo[l] = need_lo[l]; c->dec_hi[l] = need_hi[l]; }
    for (int l = 4; l >= 1; --l) {
        const int Cin = 2 * c->enc_c[l + 1];
        float* const am = want_amax ? f.cat_amax(l, 0) : nullptr;        // max |deconv(l+1)| = the decoder half of cat_l
        if (lean && c->bufs.count("pks:separation/deconv" + std::to_string(l + 1) + "/weights")) {
            f.deconv_scatter(c->p("cat" + std::to_string(l + 1)), c->enc_h[l + 1], c->enc_w[l + 1], Cin, l, c->p("cat" + std::to_string(l)),
                             2 * c->enc_c[l], true, need_lo[l + 1], need_hi[l + 1] - need_lo[l + 1], need_lo[l], need_hi[l], am, dec_planes);
            continue;
        }
        f.deconv(c->p("cat" + std::to_string(l + 1)), c->enc_h[l + 1], c->enc_w[l + 1], Cin, l, c->p("cat" + std::to_string(l)),
                 2 * c->enc_c[l], true, 0, 0, 0, 0, 0, nullptr, nullptr, 0, [am](IgemmDesc& d) { d.amax_out = am; });
    }
    // deconv1: only output rows 44..66 (mask frames 1..23) reach the cropped window -> grid rows a = 11..16.
    // Inference: sigmoid and the track-weighted sums run in its epilogue (igemm_epilogue_maskmix) and the 94 MB of logits are never
    // written; the training step keeps them (the adjoint reads them), and so does sagen_set_option("materialize_mask", 1).
    static const bool no_maskfuse = getenv("SAGEN_NO_MASKFUSE") != nullptr;
    const bool fused_tail = !no_maskfuse && !c->train_mode && !c->materialize_mask && !c->fp32_only && c->nsep == 32;
    c->mask_fused_last = fused_tail;
    if (fork2 && fused_tail) {                           // the epilogue needs the localisation coefficients (three small FCs, long done)
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_join, c->aux));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    float* const ebuf = fused_tail ? c->p("dmask") : nullptr;       // (32 bytes per bin in the buffer of the 128-byte logits)
    std::function<void(IgemmDesc&)> d1_tweak = nullptr;
    if (d1_planes && fused_tail) {
        // rows 10..16 of cat1 (all deconv1's live grid rows read) -> two fp16 planes per element, one zero pixel closing every row
        const int W1 = c->enc_w[1], C1 = 2 * c->enc_c[1];
        f.layer = "separation/cat1-planes";
        f.timed("h2_pack_rows_kernel", 0.0, [&] {
            return h2_pack_rows_launch(c->p("cat1"), (long)c->enc_h[1] * W1 * C1, (long)W1 * C1, C1, 10, B, 7, W1, C1, f.cat_amax(1, 0), f.cat_amax(1, 1), c->p("cat1p"),
                                       f.cat_a_inv(1), reinterpret_cast<unsigned*>(c->p("h2s") + 7), s); });
        const float* w_inv = c->p("h2s") + c->h2_slot.at("separation/deconv1");
        const void* planes = c->p("cat1p");
        const void* wh2 = c->p("pkh:separation/deconv1/weights");
        const float* a_inv = f.cat_a_inv(1);
        d1_tweak = [=](IgemmDesc& d) {
            d.xp3 = planes; d.xp3_fmt = 1; d.xp3_row0 = 10; d.xp3_rows = 7;
            d.p3_np = B * 7 * (W1 + 1);
            d.xp3_cstride = (unsigned)((size_t)d.p3_np * 64);
            d.xp3_bytes = (unsigned)((size_t)d.xp3_cstride * (C1 / 16));
            d.wh2 = wh2; d.wh2_bytes = (unsigned)((size_t)d.N * d.Kpad * 4);
            d.h2_a_inv = a_inv; d.h2_w_inv = w_inv;
        };
    }
    f.deconv(c->p("cat1"), 31, 127, 64, 0, c->p("dmask"), c->nsep, false, 11, 17, 67, 23L * 1024 * c->nsep, 44, c->p("coeffs"), ebuf, 23, d1_tweak);
    if (fork2 && !fused_tail) {                          // the mix needs the localisation coefficients
        SAGEN_HIP_CHECK(hipEventRecord(c->ev_join, c->aux));
        SAGEN_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    f.layer = "separation/mask-istft-mix";
    f.timed("mask_istft_kernel+ola_mix_kernel", 0.0, [&] {
        return mask_istft_mix_launch(c->p("dmask"), 23L * 1024 * c->nsep, 1, c->p("spec"), c->p("coeffs"), B, c->nsep, out,
                                     c->p("frames"), s, ebuf); });
    return f.rc;
}
int sagen_forward_u8_impl(sagen_ctx* c, const float* audio, const uint8_t* video_u8, const float* flow, float* out, hipStream_t s) {
    if (!c) return fail(SAGEN_ERR_NULL, "sagen_forward_u8: null ctx");
    if (!c->has_video) return fail(SAGEN_ERR_UNSUPPORTED, "sagen_forward_u8: the video encoder is not enabled");
    c->video_u8 = true;
    const int rc = sagen_forward_impl(c, audio, reinterpret_cast<const float*>(video_u8), flow, out, s);
    c->video_u8 = false;
    return rc;
}
void sagen_destroy_impl(sagen_ctx* c) {
    if (!c) return;
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->tune_e0) { (void)hipEventDestroy(c->tune_e0); (void)hipEventDestroy(c->tune_e1); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_stft) (void)hipEventDestroy(c->ev_stft);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
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
size_t sagen_workspace_bytes_impl(const sagen_ctx* c) { return c->ws_floats * sizeof(float); }      // (all groups' copies included)
int sagen_num_variables_impl(const sagen_ctx* c) { return (int)c->vars.size(); }
int sagen_variable_spec_impl(const sagen_ctx* c, int i, const char** name, int32_t* ndim, int64_t shape[4]) {
    if (i < 0 || i >= (int)c->vars.size()) return fail(SAGEN_ERR_SHAPE, "variable index %d out of range", i);
    *name = c->vars[i].name.c_str();
    *ndim = c->vars[i].ndim;
    for (int k = 0; k < 4; ++k) shape[k] = c->vars[i].shape[k];
    return SAGEN_OK;
}
int sagen_set_option_impl(sagen_ctx* c, const char* name, int value) {
    const std::string n = name;
    if (n == "materialize_mask") { c->materialize_mask = value != 0; return SAGEN_OK; }
    if (n == "intermediate_group") {        // grouped contexts: which group's tensors sagen_get_intermediate returns (default 0)
        if (value < 0 || value >= c->G) return fail(SAGEN_ERR_SHAPE, "intermediate_group=%d of %d groups", value, c->G);
        c->inter_group = value;
        return SAGEN_OK;
    }
    if (n == "u8_fast_stem") { c->stem8 = value != 0; return SAGEN_OK; }
    if (n == "f16_fast_stem") { c->stem16 = value != 0; return SAGEN_OK; }
    if (n == "u8_stem_h2") { c->stem8h = value != 0; return SAGEN_OK; }
    if (n == "train_bands") { c->train_bands = value != 0; return SAGEN_OK; }
    if (n == "train_rawpool") { c->train_rawpool = value != 0; return SAGEN_OK; }
    if (n == "fp16x2") { c->use_h2 = value != 0; return SAGEN_OK; }
    if (n == "planes_from_stage") { if (value < 2 || value > 6) return fail(SAGEN_ERR_SHAPE, "planes_from_stage in 2..6"); c->p3_from_stage = value; return SAGEN_OK; }
    if (n == "decoder_planes") { c->dec_planes_min_batch = value ? 1 : (1 << 30); return SAGEN_OK; }
    if (n == "plane_gather") { c->use_p3g = c->use_p3 && value != 0; return SAGEN_OK; }
    return fail(SAGEN_ERR_UNSUPPORTED, "sagen_set_option: unknown option %s", name);
}

int sagen_counter_impl(sagen_ctx* c, const char* name, uint64_t* value, hipStream_t s) {
    if (!c->ws) return fail(SAGEN_ERR_WORKSPACE, "no workspace bound");
    if (std::string(name) != "fp16x2_saturations") return fail(SAGEN_ERR_UNSUPPORTED, "sagen_counter: unknown counter %s", name);
    unsigned v[SAGEN_MAX_GROUPS] = {0};      // one counter per group
    for (int g = 0; g < c->G; ++g)
        SAGEN_HIP_CHECK(hipMemcpyAsync(&v[g], c->p("h2s") + (size_t)g * c->grp_floats + 7, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    SAGEN_HIP_CHECK(hipStreamSynchronize(s));
    uint64_t tot = 0;
    for (int g = 0; g < c->G; ++g) tot += v[g];
    *value = tot;
    return SAGEN_OK;
}

int sagen_get_intermediate_impl(const sagen_ctx* c, const char* name, const float** data, int32_t* ndim, int64_t shape[4],
                                int64_t* pixel_stride) {
    if (!c->ws) return fail(SAGEN_ERR_WORKSPACE, "no workspace bound");
    auto it = c->named.find(name);
    if (it == c->named.end()) return fail(SAGEN_ERR_SHAPE, "unknown intermediate %s", name);
    const Named& nm = it->second;
    if (c->mask_fused_last && std::string(name) == "separation/deconv1")
        return fail(SAGEN_ERR_UNSUPPORTED, "separation/deconv1: the last forward fused the mask into the deconvolution's epilogue - "
                    "sagen_set_option(ctx, \"materialize_mask\", 1) keeps the logits");
    *data = c->ws + nm.buf.off + nm.extra_off + (size_t)c->inter_group * c->grp_floats;
    *ndim = nm.ndim;
    for (int k = 0; k < 4; ++k) shape[k] = nm.shape[k];
    *pixel_stride = nm.pixel_stride;
    return SAGEN_OK;
}
