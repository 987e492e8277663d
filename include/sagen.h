/*
 * sagen.h — C ABI of the MI355X-native (gfx950) spatialaudiogen inference path.
 *
 * The reference (pedro-morgado/spatialaudiogen, TF1/Python) has no FFI; its de-facto operator
 * boundary for this path is Python.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference repo).  The shared library
 * libsagen_hip.so implements this header with hand-written HIP kernels; there is NO CPU
 * implementation of it — every compute entry point fails with SAGEN_ERR_HIP when no gfx950
 * device is present.
 *
 * Conventions
 *  - plain C: pointers + sizes only.  All data pointers are DEVICE pointers to fp32 unless the
 *    name ends in _h.  Tensors are NHWC exactly as the TF1 graph lays them out.
 *  - `stream` is a hipStream_t passed as void*.  Every launch is asynchronous on that stream; no
 *    entry point synchronises, allocates or frees device memory.  The caller owns all memory
 *    (weights, inputs, outputs, workspace) — e.g. torch tensors' data_ptr().
 *  - return value: 0 (SAGEN_OK) or a negative sagen_status; sagen_last_error() returns a
 *    thread-local message for the last failure.  No exceptions / aborts cross the ABI.
 *  - one sagen_ctx per (device, stream) pair in flight; contexts are independent.
 */
#ifndef SAGEN_H
#define SAGEN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGEN_VERSION 100   /* 0.1.0 */

typedef enum {
    SAGEN_OK = 0,
    SAGEN_ERR_NULL = -1,         /* null pointer argument */
    SAGEN_ERR_SHAPE = -2,        /* bad / inconsistent shape */
    SAGEN_ERR_UNSUPPORTED = -3,  /* configuration the HIP path does not implement */
    SAGEN_ERR_WEIGHTS = -4,      /* missing / misnamed / mis-shaped variable at bind */
    SAGEN_ERR_WORKSPACE = -5,    /* workspace too small or not bound */
    SAGEN_ERR_HIP = -6           /* HIP runtime error (no device, launch failure, ...) */
} sagen_status;

/* encoder bit mask (reference definitions.py:1-4; model.py:47-51) */
#define SAGEN_ENC_AUDIO 1
#define SAGEN_ENC_VIDEO 2
#define SAGEN_ENC_FLOW  4

/* separation mode (reference definitions.py:6-8) */
#define SAGEN_SEP_NONE      0   /* 'none'      : model.py:274-280 */
#define SAGEN_SEP_FREQ_MASK 1   /* 'unet_mask' : model.py:282-348 */

/* Mirrors SptAudioGen.__init__ + SptAudioGenParams (model.py:10-60) and the batch size the
 * caller fixes at graph-build time (deploy.py:50, eval.py:44, train.py:38). */
typedef struct {
    int32_t batch;            /* windows per forward call (BN statistics are per call) */
    int32_t encoders;         /* SAGEN_ENC_* mask; AUDIO is mandatory (model.py:207) */
    int32_t separation;       /* SAGEN_SEP_* */
    int32_t num_sep_tracks;   /* params.sep_num_tracks (32) */
    int32_t n_loc_units;      /* len(params.loc_fc_units) (2) */
    int32_t loc_units[4];     /* params.loc_fc_units ([512,512]) */
    int32_t ambi_order;       /* 1 */
    int32_t audio_rate;       /* 48000 */
    int32_t video_rate;       /* 10 */
    float   context;          /* 1.0 s */
    float   sample_duration;  /* 0.1 s */
    float   fft_window;       /* 0.025 s -> wind_size 1024 */
} sagen_config;

/* One TF variable: name is the checkpoint key, data is a device pointer in the TF
 * layout (conv HWIO, conv2d_transpose [kh,kw,Cout,Cin], FC [in,out], vectors [C]). */
typedef struct {
    const char*  name;
    const float* data;
    int32_t      ndim;
    int64_t      shape[4];
} sagen_tensor;

typedef struct sagen_ctx sagen_ctx;

int  sagen_version(void);
/* the compiler flags this library was built with (the Python host refuses a build without -fno-slp-vectorize -fno-vectorize:
 * packed-fp32 VALU next to bf16 MFMA waves of another stream returned wrong results on MI355X) */
const char* sagen_build_info(void);
/* sha256 (first 16 hex digits) over every source of this library (every .hip and .h under csrc/, and this header) at the moment it was
 * linked: the prebuilt .so travels to the GPU box with the tree, and the tests compare this with the digest of the sources that
 * travelled with it (spatialaudiogen_amd.build.source_digest) */
const char* sagen_source_digest(void);
const char* sagen_last_error(void);

/* ---- model-level: replaces SptAudioGen.inference_ops (model.py:356-434), as called at
 *      deploy.py:77,141 / eval.py:90,145 ------------------------------------------------- */
int    sagen_create(sagen_ctx** out, const sagen_config* cfg);
void   sagen_destroy(sagen_ctx* ctx);
/* bytes of scratch the caller must provide (activations + packed weights); fixed per ctx */
size_t sagen_workspace_bytes(const sagen_ctx* ctx);
/* number of variables the configuration expects, and the i-th expected name/shape */
int    sagen_num_variables(const sagen_ctx* ctx);
int    sagen_variable_spec(const sagen_ctx* ctx, int i, const char** name, int32_t* ndim, int64_t shape[4]);
/* Borrow the variables (replaces tf.train.Saver.restore, deploy.py:79-87) and repack them into
 * the kernels' layouts inside `workspace`.  Tensors must stay alive only until the stream
 * reaches the end of this call's work; workspace must stay alive and untouched by the caller
 * for the life of the ctx. */
int    sagen_bind_weights(sagen_ctx* ctx, const sagen_tensor* tensors, int n,
                          void* workspace, size_t workspace_bytes, void* stream);
/* audio [B, snd_size] (=[B,52799,1]); video/flow [B,224,448,3] (=[B,1,224,448,3]) or NULL;
 * ambi_yzx [B, snd_dur, 3] (channels Y,Z,X = ACN 1,2,3). */
int    sagen_forward(sagen_ctx* ctx, const float* audio, const float* video, const float* flow,
                     float* ambi_yzx, void* stream);
/* The same with the video frames as they come out of the JPEG decoder: uint8 [B,224,448,3]; the reference's pixel normalisation
 * x/255 - 0.5 (myutils.py:88-89, img_prep of feeder.py:121-132) is applied on the device, bit-identical to the float32 the feeder
 * would have produced.  Host -> device traffic per window: 301 KB instead of 1.2 MB. */
int    sagen_forward_u8(sagen_ctx* ctx, const float* audio, const uint8_t* video_u8, const float* flow,
                        float* ambi_yzx, void* stream);
/* Grouped launch (round 6).  The reference runs ONE sess.run per batch (deploy.py:141, eval.py:145) and its batch-norm couples the
 * windows OF a batch (model.py:197, resnet.py:123: is_training=True), so a batch is the unit of work.  A grouped context runs
 * `groups` INDEPENDENT batches of cfg->batch windows as ONE launch per layer (the group is a grid dimension of every kernel): each
 * batch keeps its own batch-norm statistics, plane scales and maxima, and its output is bit-identical to what sagen_forward[_u8] of
 * an ungrouped context (running the same launch plan: sagen_plan_set / none) gives for that batch alone - the semantics per batch are untouched, the fixed cost of a launch is paid once
 * per `groups` batches.  sagen_create_grouped: as sagen_create; sagen_workspace_bytes then covers `groups` copies of the per-batch
 * region (the packed filters are shared).  sagen_forward_grouped[_u8]: audio [groups*B, snd_size], video / flow [groups*B,224,448,3],
 * ambi_yzx [groups*B, snd_dur, 3] - the batches back to back; `groups` must be the context's.  FREQ_MASK separation and the default
 * arithmetic only (sagen_set_option values that leave it are refused at the next forward); sagen_forward[_u8] on a grouped context
 * (groups > 1) is an error, as is the training step.  sagen_get_intermediate returns the tensors of group
 * sagen_set_option(ctx, "intermediate_group", g) (default 0). */
int    sagen_create_grouped(sagen_ctx** out, const sagen_config* cfg, int groups);
int    sagen_forward_grouped(sagen_ctx* ctx, int groups, const float* audio, const float* video, const float* flow,
                             float* ambi_yzx, void* stream);
int    sagen_forward_grouped_u8(sagen_ctx* ctx, int groups, const float* audio, const uint8_t* video_u8, const float* flow,
                                float* ambi_yzx, void* stream);
/* deploy.py:143-152: out[b, n, :] = [mono[b, snd_contx/2 + n], ambi_yzx[b, n, 0..2]] -> [B,snd_dur,4] WYZX */
int    sagen_assemble_wyzx(const float* audio, const float* ambi_yzx, float* out_wyzx,
                           int batch, int snd_size, int snd_contx, int snd_dur, void* stream);
/* named intermediate of the last forward (device pointer inside the workspace) for parity tests:
 * "mag", "stft", "audio_encoder/conv1".."conv5", "<enc>_encoder/conv5_2", "bottleneck",
 * "localization/coeffs", "separation/deconv1" (needed rows only). Returns dims in shape[4]. */
int    sagen_get_intermediate(const sagen_ctx* ctx, const char* name, const float** data,
                              int32_t* ndim, int64_t shape[4], int64_t* pixel_stride);

/* Per-layer launch planning (no reference analogue; cuDNN-style autotune): runs one forward on the given
 * inputs in which every contraction times its (tile shape, split-K) candidates with hipEvents and keeps the
 * fastest; later sagen_forward calls use the stored plan.  Synchronises the stream.  Without it (or with
 * SAGEN_NO_AUTOTUNE set) shape heuristics are used.  sagen_plan_describe writes
 * "layer\ttile\tsplitk\tmicroseconds\n" per contraction and returns the number of lines. */
int    sagen_autotune(sagen_ctx* ctx, const float* audio, const float* video, const float* flow,
                      float* ambi_yzx, void* stream);
int    sagen_plan_describe(sagen_ctx* ctx, char* buf, size_t buflen);
/* pin one layer's launch: tile = index into the contraction kernels' instantiation table (sagen_num_tiles /
 * sagen_tile_name below), splitk >= 1.  An instantiation that cannot run the layer falls back to the heuristic. */
int    sagen_plan_set(sagen_ctx* ctx, const char* layer, int tile, int splitk);
/* the instantiation table: names as sagen_plan_describe / sagen_profile_report print them (host-only calls, no device
 * needed): "igemm_kernel<...>" = exact fp32 MFMA, "igemm3*_kernel<...>" = fp32-equivalent bf16x3 */
int         sagen_num_tiles(void);
const char* sagen_tile_name(int tile);

/* Measurement aid (the reference's only analogue is the samples/sec printout, myutils.py:15-26):
 * when enabled, every launch of the following sagen_forward calls is bracketed by a pair of
 * hipEvents on the launch stream.  sagen_profile_report waits for the last forward's events and
 * writes one line per launch, "kernel\tlayer\tmicroseconds\tflops\n"; returns the number of lines
 * (>= 0) or a negative sagen_status. */
int    sagen_profile_enable(sagen_ctx* ctx, int on);
/* Per-context switches.  "materialize_mask" (default 0): 1 keeps the mask decoder's last layer (model.py:320-337) and the mask
 * application as two kernels so that the logits exist in the workspace ("separation/deconv1" of sagen_get_intermediate); 0 lets the
 * forward fold sigmoid + track mix into the deconvolution's epilogue (same result, 94 MB less HBM traffic per batch of 32). */
int    sagen_set_option(sagen_ctx* ctx, const char* name, int value);
/* Further switches: "fp16x2" (default 1: the ResNet trunk's convs - resnet.py:141-236 - run on two fp16 planes per operand, three matrix
 * products per multiply; 0: three bf16 planes, six products), "plane_gather", "u8_fast_stem", "planes_from_stage" (INTEGRATION.md 5).
 * sagen_counter reads a diagnostic counter of the context after synchronising `stream`: "fp16x2_saturations" = activation elements the
 * fp16x2 plane passes had to clamp to +-65000 since sagen_bind_weights (scales come from batch statistics with 64 x headroom beyond
 * eight standard deviations: 0 unless an activation is not finite or that far out; > 0 means results of the affected forwards are
 * not fp32-equivalent - rerun with "fp16x2" = 0). */
int    sagen_counter(sagen_ctx* ctx, const char* name, uint64_t* value, void* stream);
int    sagen_profile_report(sagen_ctx* ctx, char* buf, size_t buflen);

/* ---- op-level (unit-testable; same conventions) ----------------------------------------- */

/* myutils.stft (myutils.py:119-147) fused with the crop + tf.abs of audio_encoder_ops
 * (model.py:166-178): hop = wind/4, periodic Hann, 1024-pt FFT.
 * mag [B, f1-f0, 1024] magnitudes of frames [f0,f1); spec [B, c1-c0, 513, 2] (re,im) of frames
 * [c0,c1) (bins 0..512; the rest is the Hermitian mirror).  Either output may be NULL. */
int sagen_stft_mag(const float* audio, int batch, int n_samples, int f0, int f1, float* mag,
                   int c0, int c1, float* spec, void* stream);

/* tfw.conv_2d (core.py:156-220) = tf.nn.convolution NHWC/HWIO + (bias | nothing) + optional ReLU.
 * padding: 0 = VALID, 1 = SAME (TF asymmetric).  Batch-norm is NOT applied here: pass bn_stats
 * (>= sagen_bn_stats_floats(...) floats, 8-byte aligned; holds 2*cout fp64 accumulators that the call
 * zeroes and fills with the per-channel sum and sum of squares of the raw output) and
 * call sagen_bn_finalize; the consumer applies scale/shift (+ReLU) via in_scale/in_shift.
 * in_scale/in_shift [Cin] (nullable): input is relu(x*scale+shift) before padding.
 * scratch: >= sagen_conv2d_scratch_bytes(...) bytes for the repacked filter (and, for cin == 3, the
 * zero-bordered 4-channel copy of the input).  Supported cin: a power of two >= 4; 3; or 1 with
 * padding VALID, kw % 4 == 0 and sw % 4 == 0 (the spectrogram conv). */
size_t sagen_conv2d_scratch_bytes(int batch, int h, int w, int kh, int kw, int cin, int cout);
size_t sagen_bn_stats_floats(int batch, int hout, int wout, int cout);
int sagen_conv2d(const float* x, int batch, int h, int w, int cin,
                 const float* w_hwio, int kh, int kw, int cout, int sh, int sw, int padding,
                 const float* bias, int relu, const float* in_scale, const float* in_shift,
                 float* y, float* bn_stats, void* scratch, size_t scratch_bytes, void* stream);

/* contrib batch_norm in training mode (core.py:6,209-210; eps 1e-3, biased variance):
 * turns the partial sums into scale = gamma/sqrt(var+eps), shift = beta - mean*scale. */
int sagen_bn_finalize(const float* bn_stats, int batch, int hout, int wout, int cout,
                      const float* gamma, const float* beta, float eps,
                      float* scale, float* shift, void* stream);
/* y = relu(x*scale + shift (+ residual)) elementwise over [n_pixels, C] (resnet.py:221,235) */
int sagen_bn_apply_relu(const float* x, const float* scale, const float* shift,
                        const float* residual, float* y, int64_t n_pixels, int c, void* stream);
/* tf.nn.max_pool 3x3 s2 'SAME' (resnet.py:135) of relu(x*scale+shift) (scale NULL = identity) */
int sagen_maxpool3x3s2(const float* x, const float* scale, const float* shift, float* y,
                       int batch, int h, int w, int c, void* stream);

/* tfw.fully_connected (core.py:43-93): y[M,N] = act(x[M,K] @ w[K,N] + b). */
size_t sagen_fc_scratch_bytes(int m, int k, int n);
int sagen_fc(const float* x, int m, int k, const float* w_kn, int n, const float* bias, int relu,
             float* y, void* scratch, size_t scratch_bytes, void* stream);

/* tfw.deconv_2d (core.py:96-153) = tf.nn.conv2d_transpose VALID, w [kh,kw,Cout,Cin], + bias
 * (+ReLU).  y [B, h*sh+kh-sh, w*sw+kw-sw, cout]. */
size_t sagen_deconv2d_scratch_bytes(int kh, int kw, int cin, int cout, int sh, int sw);
int sagen_deconv2d(const float* x, int batch, int h, int w, int cin,
                   const float* w_hwoi, int kh, int kw, int cout, int sh, int sw,
                   const float* bias, int relu, float* y,
                   void* scratch, size_t scratch_bytes, void* stream);

/* separation tail + decoder (model.py:326-347, myutils.istft myutils.py:181-211, model.py:421-434):
 * dmask  [B, 28, 1024, ntracks] : deconv1 output rows 43:71 (pre-sigmoid), NHWC
 * spec   [B, 28, 513, 2]        : STFT frames 89:117 (bins 0..512)
 * coeffs [B, 3, 3, ntracks+1]   : localization output (step, out-channel, track | bias)
 * ambi_yzx [B, 4800, 3].  scratch >= sagen_mask_istft_mix_scratch_bytes(batch). */
size_t sagen_mask_istft_mix_scratch_bytes(int batch);
int sagen_mask_istft_mix(const float* dmask, const float* spec, const float* coeffs, int batch,
                         int ntracks, float* ambi_yzx, void* scratch, size_t scratch_bytes, void* stream);

/* SptAudioGen.evaluation_ops (model.py:110-154) for 0.1 s windows at 48 kHz, first-order (3 predicted channels):
 * per_sample [4][B][3] = stft distance (model.py:62-76; before the x100 and channel masking of :122-127), log-spectral
 * distance on the 1200-point STFT (:78-94), temporal MSE (:96-99; before x5e3) and SNR (:101-108);
 * power_sums[2] (fp64) = sum pred^2, sum target^2 over everything (-> pow/pred, pow/gt after /(3B)).
 * scratch: >= sagen_eval_scratch_bytes(batch) bytes, initialised ONCE with sagen_eval_init (DFT matrix). */
size_t sagen_eval_scratch_bytes(int batch);
int sagen_eval_init(void* scratch, size_t scratch_bytes, int batch, void* stream);
int sagen_eval_metrics(const float* pred_yzx, const float* target_yzx, int batch, float* per_sample,
                       double* power_sums, void* scratch, size_t scratch_bytes, void* stream);

/* AmbiDecoder.decode('projection') + RMS map (pyutils/ambisonics/decoder.py:24-28,
 * distance.py:41-52; SH matrix common.py:151-178, order 1 ACN/SN3D):
 * ambi_wyzx [T,4]; sh [P,4] device matrix; rms [P] = sqrt(mean_t (ambi . sh[p])^2).
 * The rms buffer must hold P + 24 floats (the tail is used for the 4x4 second-moment matrix). */
int sagen_power_map(const float* ambi_wyzx, int64_t t, const float* sh, int p, float* rms, void* stream);

/* One map per chunk: SphericalAmbisonicsVisualizer.loop_frames (pyutils/ambisonics/distance.py:41-59) yields one RMS map
 * per `window` seconds of audio; ambix_emd (distance.py:133-143, called at eval.py:190) compares them frame by frame.
 * ambi_wyzx [nchunks, T, 4] (chunks back to back); rms [nchunks, P]; moments = caller scratch of nchunks*10 doubles. */
int sagen_power_map_batched(const float* ambi_wyzx, int nchunks, int64_t t, const float* sh, int p, float* rms, double* moments,
                            void* stream);

/* ---- training step (reference train.py:137-236) -------------------------------------------------------
 * Loss of the reference: losses['stft/mse'] = metrics['stft/avg'] (model.py:156-159, 122-127; stft_for_loss myutils.py:151-178).
 * pred / target [B,4800,3]; mask [B,3] channel mask (the Y,Z,X columns of the feeder's [B,4] W,Y,Z,X mask, train.py:127) or NULL;
 * grad [B,4800,3] = dL/dpred (or NULL); loss = one fp64 (or NULL). */
int sagen_stft_loss_grad(const float* pred_yzx, const float* target_yzx, const float* mask, int batch, float* grad, double* loss,
                         void* stream);
/* tf.train.AdamOptimizer (myutils.py:214-222) on one flat fp32 bucket of n floats (n % 4 == 0, 16-byte aligned):
 * m,v slots updated in place, params -= lr_t * m / (sqrt(v) + epsilon); lr_t = lr * sqrt(1-beta2^t) / (1-beta1^t) from the host;
 * grads are multiplied by grad_scale first (1 / world_size after a sum all-reduce). */
int sagen_adam_update(float* params, const float* grads, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                      float epsilon, float grad_scale, void* stream);

/* One sess.run(train_op) without the optimiser (train.py:208; opt.minimize = compute_gradients + apply_gradients,
 * myutils.py:220-221): forward with retained activations, loss, backward.  Replaces tf.gradients over the graph of
 * model.py:356-434 for separation 'unet_mask'.
 *  - the ctx must be bound (sagen_bind_weights) to the LIVE parameter tensors: every sagen_train_step re-packs the filters from
 *    them first, so the optimiser may update them in place between steps;
 *  - sagen_train_bind: `grads` names, for every trainable variable, where its gradient is written (same name / shape / layout
 *    as the variable; e.g. views into flat gradient buckets); `moving` (optional) names writable bn/moving_mean and
 *    bn/moving_variance tensors, updated like the contrib batch_norm update ops (decay 0.99) when update_moving_averages != 0;
 *    train_workspace: >= sagen_train_workspace_bytes(ctx) bytes, 256-byte aligned, owned by the caller, used only by this ctx;
 *  - sagen_train_step: target_yzx [B,4800,3], mask [B,3] or NULL, pred_yzx [B,4800,3] or NULL (the prediction of this step),
 *    loss: device pointer to one fp64 or NULL.  Asynchronous on `stream`; gradients are complete when the stream reaches the end
 *    of the call's work.  Weight gradients are reduced in a fixed order (bit-reproducible). */
size_t sagen_train_workspace_bytes(sagen_ctx* ctx);
int sagen_train_bind(sagen_ctx* ctx, const sagen_tensor* grads, int n_grads, const sagen_tensor* moving, int n_moving,
                     void* train_workspace, size_t train_workspace_bytes, void* stream);
int sagen_train_step(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, const float* target_yzx,
                     const float* mask, float* pred_yzx, double* loss, int update_moving_averages, void* stream);
/* The same with the video frames as the training feeder decodes them (feeder.py:222-260: uint8 [B,224,448,3], normalised
 * x / 255 - 0.5 by myutils.py:88-89 - here on the device, as in sagen_forward_u8): the stem's forward runs on the exact one-plane
 * operand u - 128 (three matrix products per multiply instead of six). */
int sagen_train_step_u8(sagen_ctx* ctx, const float* audio, const uint8_t* video_u8, const float* flow, const float* target_yzx,
                        const float* mask, float* pred_yzx, double* loss, int update_moving_averages, void* stream);
/* sagen_autotune for the training step: one step on these inputs in which every contraction (forward and data gradients) times
 * its launch candidates; the plan is stored in the ctx.  The gradients it leaves behind are not meaningful.  Synchronises. */
int sagen_train_autotune(sagen_ctx* ctx, const float* audio, const float* video, const float* flow, const float* target_yzx,
                         const float* mask, void* stream);
/* Data-parallel training (train.py runs one device; here one process per GPU, gradients summed over the ranks): overlap the
 * exchange with the backward pass.  The caller partitions the trainable variables into n_buckets flat buckets (names[i] belongs to
 * bucket[i]) and hands in one event per bucket (hipEvent_t, e.g. torch.cuda.Event.cuda_event); every following sagen_train_step
 * records events[b] as soon as the last kernel writing a gradient of bucket b has been enqueued - behind both of the step's streams -
 * so a communication stream that waits for events[b] can all-reduce bucket b under the rest of the backward.  Every event is
 * recorded by every step (buckets that never complete early are recorded at the end).  n = 0 switches it off. */
int sagen_train_set_grad_events(sagen_ctx* ctx, const char* const* names, const int32_t* bucket, int n, void* const* events, int n_buckets);
/* named buffer of the train workspace (parity tests): e.g. "t:dcoeffs" [B*3][100], "t:ddmask" [B,31,1024,ntracks] (deconv1 output
 * rows 40..70), "t:dpred", "t:g:feat" (dL/d conv5_2) */
int sagen_train_get_buffer(const sagen_ctx* ctx, const char* name, const float** data, size_t* n_floats);

/* ---- backward, op level (unit-testable) ------------------------------------------------------------------------------
 * Filter gradient of tf.nn.convolution / tf.nn.conv2d_transpose / tf.matmul (core.py:206,140,79) in one form:
 *   dw[th,tw,g,d] = sum_{b,i,j} G[b, i*sh + th + h0, j*sw + tw + w0, g] * D[b,i,j,d]      (G zero outside [hg) x [wg))
 * conv (HWIO): G = x [B,hg,wg,cg=cin], D = dy [B,hd,wd,cd=cout], h0/w0 = -pad_before; conv2d_transpose ([kh,kw,Cout,Cin]):
 * G = dy, D = x, h0 = w0 = 0; FC: hg = wg = hd = wd = kh = kw = 1, batch = rows.  cg, cd multiples of 4.  scratch (optional,
 * sagen_wgrad_scratch_bytes) enables pixel-range splitting. */
size_t sagen_wgrad_scratch_bytes(int kh, int kw, int cg, int cd);
int sagen_wgrad(const float* g, int batch, int hg, int wg, int cg, const float* d, int hd, int wd, int cd, int kh, int kw, int sh, int sw,
                int h0, int w0, float* dw, void* scratch, size_t scratch_bytes, void* stream);
/* Input gradient of tfw.conv_2d (core.py:206): dy [B,hout,wout,cout], w HWIO -> dx [B,h,w,cin].  Stride 1: any padding; strided:
 * VALID, or SAME with no padding before (every strided conv of the path).  cout a power of two >= 4. */
size_t sagen_conv2d_bwd_data_scratch_bytes(int kh, int kw, int cin, int cout, int sh, int sw);
int sagen_conv2d_bwd_data(const float* dy, int batch, int hout, int wout, int cout, const float* w_hwio, int kh, int kw, int cin,
                          int sh, int sw, int padding, int h, int w, float* dx, void* scratch, size_t scratch_bytes, void* stream);
/* contrib batch_norm (training mode) backward: dz = (ga + gb) * (act > 0) (gb, act nullable) is the gradient at the BN output;
 * bn_stats = the accumulators sagen_conv2d filled for the raw conv output y.  dy = gradient at y; dz (nullable) = the masked
 * gradient itself; dgamma / dbeta [C].  scratch: sagen_bn_bwd_scratch_bytes(c) bytes, 16-byte aligned. */
size_t sagen_bn_bwd_scratch_bytes(int c);
int sagen_bn_bwd(const float* ga, const float* gb, const float* act, const float* y, const float* bn_stats, const float* gamma,
                 const float* beta, float eps, int64_t n_pixels, int c, float* dy, float* dz, float* dgamma, float* dbeta,
                 void* scratch, size_t scratch_bytes, void* stream);
/* backward of sagen_maxpool3x3s2(relu(bn(y0))): pooled = its output, ga (+ gb) = gradient at the pooled tensor -> dz at bn(y0) */
int sagen_maxpool3x3s2_bwd(const float* y0, const float* bn_stats, const float* gamma, const float* beta, float eps, const float* pooled,
                           const float* ga, const float* gb, float* dz, int batch, int h, int w, int c, void* stream);
/* adjoint of sagen_mask_istft_mix: dpred [B,4800,3] -> d_dmask [B,28,1024,ntracks], d_coeffs [B,3,3,ntracks+1] */
size_t sagen_mask_istft_mix_bwd_scratch_bytes(int batch, int ntracks);
int sagen_mask_istft_mix_bwd(const float* dmask, const float* spec, const float* coeffs, const float* dpred, int batch, int ntracks,
                             float* d_dmask, float* d_coeffs, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAGEN_H */
