// Launcher declarations of every HIP kernel family of the path (gfx950).
#pragma once
#include "common.h"
#include "wave_reduce.h"

namespace sagen {

// -----------------------------------------------------------------------------------------
// Implicit-GEMM convolution on fp32 MFMA (igemm.hip).  One kernel serves every contraction of
// the path: VALID/SAME convs, 1x1 strided shortcuts, FCs, and conv2d_transpose written as a
// stride-1 conv with a depth-to-space epilogue.
//
//   out[m, n] = sum_k A[m, k] * Wp[n, k],   m = (b, a, bb) over the output GRID,
//   k = (tap, c):  A[m,k] = f(x[b, a*in_sh + dh(tap), bb*in_sw + dw(tap), c])  (0 outside),
//   f = identity or relu(v*in_scale[c] + in_shift[c]) (previous layer's batch-norm),
//   n = (ry, rx, o): stored at y[b, a*dsh + ry, bb*dsw + rx, o] (+bias[o], optional ReLU).
// -----------------------------------------------------------------------------------------
// Training-mode batch-norm of a producer layer, by reference to its fp64 (sum, sumsq) accumulators: consumers
// derive scale = gamma/sqrt(var+eps), shift = beta - mean*scale themselves (no finalize launch).
struct BnRef {
    const double* acc = nullptr;      // [2][C]; nullptr = no batch-norm
    const float* gamma = nullptr;
    const float* beta = nullptr;
    double inv_count = 0.0;
    float eps = 1e-3f;
};

// an IgemmDesc::amax_out target: H2_AMAX_SLOTS words, one per 128-byte line (atomics to ONE line serialise in the L2 like atomics to
// one word: with the 64 words adjacent the spectrogram conv still took 31 us instead of 18)
constexpr int H2_AMAX_SLOTS = 64, H2_AMAX_STRIDE = 32, H2_AMAX_FLOATS = H2_AMAX_SLOTS * H2_AMAX_STRIDE;

struct IgemmDesc {
    const float* x = nullptr;         // input, channel offset already applied
    const float* w = nullptr;         // packed filter [N][Kpad], k contiguous, zero padded
    int w_split = 0;                  // 1: the bf16x3 planes, tiled [Kpad/16][3][N][16] (pack_split_launch), follow the fp32 filter at w + N*Kpad
    float* y = nullptr;               // output, channel / row offset already applied
    const float* bias = nullptr;      // [Cout] or null
    const float* in_scale = nullptr;  // [Cin] or null
    const float* in_shift = nullptr;
    BnRef bn_in;                      // alternative to in_scale/in_shift: derive them in-kernel (Cin <= 512)
    double* stats = nullptr;          // fp64 accumulators [2][N] of (sum, sumsq) of the raw output (atomics), or null
    // exact max |y| over everything this launch stores (after bias / ReLU): atomicMax of the bit pattern into H2_AMAX_SLOTS zeroed words
    // (the maximum over the slots is the tensor's), or null - the scale of the fp16x2 planes a consumer's pack pass makes of the tensor
    // (h2_pack_rows_launch): exact, so nothing can saturate
    float* amax_out = nullptr;
    float* splitk_ws = nullptr;       // [splitk][M][N] when splitk > 1
    // in-launch combine of the split-K partials (igemm_common.h: igemm_epilogue): one ticket per output tile, zero between launches;
    // the workgroup that draws the last ticket of its tile adds the partials in the fixed order z = 0 .. splitk-1, applies bias / ReLU
    // and writes y (each row sk_rep times) - no reducer launch.  Needs N % 4 == 0, 16-byte aligned y rows, a dense plain output.
    int* sk_ticket = nullptr;
    int sk_rep = 1;
    int M = 0, N = 0, K = 0, Kpad = 0;
    // output grid
    int Hg = 1, Wg = 1, g_h0 = 0, g_w0 = 0;
    // input
    int Hin = 1, Win = 1, Cin = 0, ldx = 0;
    long x_bstride = 0;
    int in_sh = 1, in_sw = 1;
    // taps: tap -> (th, tw) = (tap / TW, tap % TW); dh = th*tap_sh + tap_h0; dw = tw*tap_sw + tap_w0
    int ntaps = 1, TW = 1, tap_sh = 1, tap_sw = 1, tap_h0 = 0, tap_w0 = 0;
    int log2Cin = -1;                 // required (>=2) when ntaps > 1
    // output (depth-to-space factors dsh x dsw; N = dsh*dsw*Cout)
    int dsh = 1, dsw = 1, Cout = 0;
    int Hlim = 1, Wlim = 1;           // valid output coordinates: Y < Hlim, X < Wlim
    long y_bstride = 0;
    long y_rstride = 0;               // elements per output row (Wout * ldy)
    int ldy = 0;
    int relu_out = 0;
    int splitk = 1;
    // filled by igemm_launch
    unsigned x_bytes = 0, w_bytes = 0;
    unsigned y_bytes = 0;             // conv3h_kernel: extent of y for its buffer stores (set by conv3h_dispatch)
    int no_bounds = 0;
    int uniform_taps = 0;
    // conv3p_kernel (conv3p.hip): the activation as pre-split bf16 planes [Cin/16][p3_np][3][16], p3_np = B*Hin*(Win+1)
    const void* xp3 = nullptr;
    unsigned xp3_bytes = 0, xp3_cstride = 0;
    int p3_np = 0;
    int xp3_fmt = 0;                  // 0: three bf16 planes (96 B per pixel and chunk); 1: two fp16 planes of v * 2^ka (64 B; conv3h.hip)
    int xp3_row0 = 0, xp3_rows = 0;   // conv3g_kernel: the planes hold image rows [xp3_row0, xp3_row0 + xp3_rows) only (xp3_rows = 0: all Hin rows)
    const void* wh2 = nullptr;        // conv3h_kernel: the filter as two fp16 planes of w * 2^kw, [Kpad/16][2][N][16]
    unsigned wh2_bytes = 0;
    const float* h2_a_inv = nullptr;  // device scalars 2^-ka (written by the plane producer) and 2^-kw (written by the filter pack)
    const float* h2_w_inv = nullptr;
    unsigned p3_magic_wp = 0, p3_magic_h = 0;   // filled by conv3p_dispatch: floor(2^32 / d) + 1 for d = Win + 1, Hin (exact quotients by mul-hi)
    // Fused decoder tail (deconv1 of the mask decoder at inference, model.py:326-337 + 421-434): instead of the 32 mask logits of an
    // output pixel the epilogue writes E[c] = sum_j w[step_c][out_c][j] * sigmoid(logit_j + bias_j), c = (step lo/hi, output) -
    // what mask_istft_kernel would compute from them; the logits (94 MB at batch 32) never reach HBM.  igemm3_kernel tiles only.
    const float* mm_coeffs = nullptr; // localisation coefficients [B][3 steps][3 outputs][Cout + 1]
    float* mm_out = nullptr;          // [B][mm_nf][Wlim][8] (6 used)
    int mm_row0 = 0, mm_nf = 0;       // output row y holds mask frame mm_f_lo + (y - mm_row0); rows outside [0, mm_nf) are not produced
    int mm_f_lo = 1;
    // debug builds (-DSAGEN_TRACE): phase timeline of workgroup `trace_block`
    void* trace = nullptr;
    int trace_block = 0;
    // grouped launch (common.h: GroupInfo): filled by igemm_launch from cur_group(); the kernels relocate their per-group pointers
    GroupInfo grp;
};
// every pointer of the descriptor that may live in the per-group region of the workspace, moved to group g's copy (filters, the
// caller's variables and null pointers lie outside the region and stay)
__device__ __forceinline__ void igemm_relocate(IgemmDesc& d, int g) {
    const GroupInfo gi = d.grp;
    d.x = grp_ptr(d.x, gi, g); d.y = grp_ptr(d.y, gi, g);
    d.in_scale = grp_ptr(d.in_scale, gi, g); d.in_shift = grp_ptr(d.in_shift, gi, g);
    d.bn_in.acc = grp_ptr(d.bn_in.acc, gi, g);
    d.stats = grp_ptr(d.stats, gi, g); d.amax_out = grp_ptr(d.amax_out, gi, g);
    d.splitk_ws = grp_ptr(d.splitk_ws, gi, g); d.sk_ticket = grp_ptr(d.sk_ticket, gi, g);
    d.xp3 = grp_ptr(d.xp3, gi, g);
    d.h2_a_inv = grp_ptr(d.h2_a_inv, gi, g); d.h2_w_inv = grp_ptr(d.h2_w_inv, gi, g);
    d.mm_coeffs = grp_ptr(d.mm_coeffs, gi, g); d.mm_out = grp_ptr(d.mm_out, gi, g);
}
// grid.z of an IgemmDesc launch: groups x split-K for igemm_kernel / igemm3_kernel (blockIdx.z = g * splitk + z), groups for the others

// tile configurations (BM x BN, 4 waves)
// The first six are the shape-heuristic set; the rest exist for the autotuner (2-stage LDS ring = less LDS,
// more resident workgroups; extra aspect ratios).
enum IgemmTile {
    TILE_128x128 = 0, TILE_128x64, TILE_256x64, TILE_64x64, TILE_128x32, TILE_32x128,
    TILE_128x128_S2, TILE_128x64_S2, TILE_256x64_S2, TILE_64x64_S2,
    TILE_64x128, TILE_64x128_S2, TILE_64x256, TILE_64x256_S2, TILE_256x32,
    // K tile of 32 (half the barriers per MFMA; needs Kpad % 32 == 0)
    TILE_64x64_K32, TILE_64x128_K32, TILE_128x64_K32, TILE_128x128_K32, TILE_32x128_K32, TILE_128x32_K32,
    // fp32-equivalent bf16x3 operand split on the bf16 matrix cores (needs IgemmDesc::w_split)
    TILE_B3_128x128, TILE_B3_128x64, TILE_B3_256x64, TILE_B3_64x64, TILE_B3_64x128, TILE_B3_64x256, TILE_B3_32x128,
    TILE_B3_128x32,
    // ... with two K tiles of 16 per barrier step
    TILE_B3_128x64_K2, TILE_B3_64x64_K2, TILE_B3_64x128_K2, TILE_B3_32x128_K2, TILE_B3_128x32_K2,
    // bf16x3 for dense 3x3 stride-1 SAME convs: the three horizontal taps share one activation tile (igemm3dw_kernel)
    TILE_B3DW_128x128, TILE_B3DW_128x64, TILE_B3DW_256x64, TILE_B3DW_64x128, TILE_B3DW_64x64, TILE_B3DW_64x256,
    // ... the three taps also share one barrier step (narrow N)
    TILE_B3DWM_128x64, TILE_B3DWM_256x64, TILE_B3DWM_64x64, TILE_B3DWM_64x128,
    // bf16x3 for the 7x7 stride-2 stem over the padded 4-channel image, K ordered (dh, dw padded to 8, c) (igemm3s2_kernel)
    TILE_B3S2_256x64, TILE_B3S2_128x64,
    // bf16x3 for dense 3x3 stride-1 SAME convs over pre-split activation planes (conv3p_kernel; needs IgemmDesc::xp3)
    TILE_P3_128x64, TILE_P3_128x128, TILE_P3_64x64,
    // ... two phase-locked 4-wave teams per workgroup (conv3pp_kernel): two 128x64 tiles / the two K halves of one
    TILE_P3PP_PAIR, TILE_P3PP_SPLITK,
    // bf16x3 for ANY strided / multi-tap conv over pre-split activation planes, operand tiles gathered by LDS-DMA (conv3g_kernel)
    TILE_P3G_128x64_K2, TILE_P3G_64x64_K2, TILE_P3G_64x128_K2, TILE_P3G_128x128_K1,
    // fp16x2 for dense 3x3 stride-1 SAME convs over TWO fp16 planes per operand: three products per multiply (conv3h_kernel)
    TILE_P3H_128x64, TILE_P3H_128x128, TILE_P3H_64x64, TILE_P3H_256x64,
    // ... and for any strided / multi-tap conv over them (conv3g_kernel, gathered operand tiles)
    TILE_P3GH_128x64_K3, TILE_P3GH_64x64_K4, TILE_P3GH_128x128_K2, TILE_P3GH_64x128_K3,
    // conv3h_kernel with two / four 16-channel chunks per barrier step (the deep, latency-bound layers)
    TILE_P3H_128x64_C2, TILE_P3H_64x64_C2, TILE_P3H_64x64_C4,
    // conv3g_kernel on fp16x2 planes with the fused decoder tail as its epilogue (deconv1 at inference: IgemmDesc::mm_out), 2 / 4 K tiles per group
    TILE_P3GH_MM_64x128_K2, TILE_P3GH_MM_64x128_K4, TILE_P3GH_MM_128x128_K2, TILE_P3GH_MM_128x256_K2,
    // conv3hr_kernel: conv3h_kernel with a three-deep ring of activation images beside the two filter stages (conv3h.hip)
    TILE_P3HR_256x64, TILE_P3HR_128x64, TILE_P3HR_64x64_C2,
    TILE_P3HR_128x128,         // (round 6: the 128x128 tile with the three-deep activation ring - 72 KB of LDS, two workgroups per CU)
    TILE_AUTO
};

int igemm_launch(const IgemmDesc& d, IgemmTile tile, hipStream_t s);
int igemm_grid_m(const IgemmDesc& d, IgemmTile tile);       // number of M tiles (stats rows)
IgemmTile igemm_pick_tile(const IgemmDesc& d);
const char* igemm_tile_name(IgemmTile t);
int igemm_tile_bm(IgemmTile t);
int igemm_tile_bn(IgemmTile t);
int igemm_tile_bk(IgemmTile t);
bool igemm_tile_split(IgemmTile t);                   // bf16x3 variant (igemm3.hip)?
int igemm3_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s);   // launch only; igemm_launch validates
int igemm3dw_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s); // (igemm3dw.hip)
int igemm3s2_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s); // (igemm3s2.hip)
int conv3p_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s);   // (conv3p.hip)
int conv3g_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s);   // (conv3g.hip)
int conv3h_dispatch(const IgemmDesc& d, IgemmTile tile, hipStream_t s);   // (conv3h.hip)
int h2_filter_pack_launch(const float* wp, int N, int Kpad, void* w2, unsigned* scratch, float* w_inv, hipStream_t s);
struct H2Job {                       // one layer of the batched fp16x2 filter pack (1024 elements per block)
    const float* wp = nullptr;       // fp32 packed filter [N][Kpad]
    void* w2 = nullptr;              // out: two fp16 planes [Kpad/16][2][N][16] of w * 2^kw
    float* w_inv = nullptr;          // out: 2^-kw
    int N = 0, Kpad = 0, first_block = 0, pad_ = 0;
};
// amax: scratch of njobs + nblocks words (the per-job maxima and the per-workgroup partials behind them)
int h2_filter_pack_multi_launch(const H2Job* jobs_dev, int njobs, int nblocks, unsigned* amax, hipStream_t s);
bool conv3g_ok(const IgemmDesc& d);                   // geometry conv3g_kernel can run (given planes)
constexpr int SK_TICKETS = 8192;                      // tiles an in-launch split-K combine can track (IgemmDesc::sk_ticket)
inline bool igemm_tile_fused_splitk(IgemmTile) { return true; }     // every kernel that writes split-K partials does it through igemm_epilogue
bool igemm_tile_p3(IgemmTile t);                      // conv3p_kernel tile (pre-split activation planes, no split-K)?
bool igemm_tile_grouped(IgemmTile t);                 // does the tile's kernel support the grouped launch (common.h: GroupInfo)?
bool igemm_tile_dh_split(IgemmTile t);                // ... except conv3h_kernel's dh-split: split-K = 3 exactly, one filter row per workgroup
bool igemm_p3_eligible(const IgemmDesc& d);           // dense 3x3 stride-1 SAME conv that conv3p_kernel can run (given planes)
bool igemm_tile_ok(const IgemmDesc& d, IgemmTile t);   // can this instantiation run the problem?              // instantiation name as rocprofv3 prints it
// out[(m*rep + r)*ldy + n] = act(sum_z ws[z][m][n] + bias[n]),  r in [0,rep); with `stats` also the
// per-channel (sum, sumsq) of the raw sums, accumulated into stats[2][N] (fp64 atomics)
constexpr int SPLITK_RB = 16;
int splitk_reduce_launch(const float* ws, int splitk, int M, int N, const float* bias, int relu,
                         float* y, int ldy, int rep, double* stats, hipStream_t s, float* amax_out = nullptr);      // amax_out: IgemmDesc::amax_out of what is written

// stride-1 conv2d_transpose in scatter form (igemm.hip): gathers act(bias + sum_z sum_taps T_z[(b, y'-p, x'-q)][(p, q, o)]) from the
// partials ws[splitk][B*Hin*Win][kh*kw*Cout] of the GEMM over the INPUT pixels into y[b, y', x', o] (pixel stride ldy)
// Strided transposed convs the same way (out[b, y*sh+p, x*sw+q, o]); a band of input rows [in_row0, in_row0 + R) contracted
// (M = B*R*Win), output rows [y0, y1) written (the live rows of the mask decoder); amax_out: IgemmDesc::amax_out of what is written.
struct DeconvGather {
    int B = 0, Hin = 0, Win = 0, kh = 1, kw = 1, sh = 1, sw = 1, Cout = 0;
    int in_row0 = 0, R = 0;           // the GEMM's rows: input rows [in_row0, in_row0 + R) of every image
    int y0 = 0, y1 = 0;               // output rows written
    int relu = 0, ldy = 0;
    const float* bias = nullptr;
    float* y = nullptr;               // [B][Hout][Wout][ldy]
    float* amax_out = nullptr;
};
int deconv_gather_launch(const float* ws, int splitk, const DeconvGather& g, hipStream_t s);

// filter repacking (pack.hip).  All produce [Npad][Kpad] with zero padding.
// conv  : Wp[n][(tap, c)] = W_hwio[tap][c][n],  c < cin_src (cin_pad >= cin_src).  With tw_pad > tw_src > 0 the packed
//         taps are rows of tw_pad (tap = th*tw_pad + tw) over a source of tw_src taps per row; tw >= tw_src packs zeros.
int pack_conv_launch(const float* w_hwio, int ntaps, int cin_src, int cin_pad, int cout,
                     float* wp, int Npad, int Kpad, hipStream_t s, int tw_src = 0, int tw_pad = 0);
// appends the bf16x3 planes of a packed filter: wp must have room for N*Kpad floats + 3*N*Kpad bf16
int pack_split_launch(float* wp, int N, int Kpad, hipStream_t s);
inline size_t packed_split_floats(size_t n_fp32) { return n_fp32 + (3 * n_fp32 + 1) / 2; }
// ALL filter packs of a context in one launch (bind; once per training step): one job per variable, fp32 pack + its bf16x3 planes
// written by the same thread (no read-back of the packed filter).  The single-variable launchers above / below stay for the op level.
enum PackKind { PACK_CONV = 0, PACK_DECONV, PACK_FLIPT, PACK_ROWS, PACK_DECONV_PHASE };
struct PackJob {
    const float* src = nullptr;      // the variable (TF layout)
    float* dst = nullptr;            // fp32 [N][Kpad], followed by the planes [Kpad/16][3][N][16] bf16
    int kind = 0, N = 0, Kpad = 0;
    // conv: ntaps, cin_src, cin_pad, cout, tw_src, tw_pad | deconv: kh, kw, cout, cin, sh, sw, ntw | flipT: ntaps, cin, cout | rows: rows, cols
    // deconv phase (stride 2, ONE output phase (ry, rx), only its non-zero taps): kh, kw, cout, cin, ry, rx, ntw of the phase
    int p[7] = {0, 0, 0, 0, 0, 0, 0};
    int first_block = 0;             // of this job inside the launch (1024 elements per block)
};
// conv jobs (a [K][N] -> [N][K] transpose) run as 32 x 32 tiles through LDS, the others as flat runs of 1024 elements
inline int pack_job_blocks(int kind, long N, long Kpad) {
    return kind == PACK_CONV ? (int)(((N + 31) / 32) * ((Kpad + 31) / 32)) : (int)((N * Kpad + 1023) / 1024);
}
// block0: the launch covers blocks [block0, block0 + nblocks) of the table's numbering (jobs_dev / njobs = the jobs those blocks belong to)
int pack_multi_launch(const PackJob* jobs_dev, int njobs, int nblocks, hipStream_t s, int block0 = 0);
// deconv as depth-to-space conv: n = (ry, rx, o), k = (dp, dq, c);
// Wp[n][k] = W[ry + sh*dp][rx + sw*dq][o][c] (0 when outside the kh x kw kernel)
int pack_deconv_launch(const float* w_hwoi, int kh, int kw, int cout, int cin, int sh, int sw,
                       float* wp, int Npad, int Kpad, hipStream_t s);

// -----------------------------------------------------------------------------------------
// batch-norm / elementwise (elementwise.hip)
// -----------------------------------------------------------------------------------------
// stats: fp64 accumulators [2][C] (sum, sumsq) filled by the producer's atomics
int bn_finalize_launch(const double* stats, long count, int C, const float* gamma,
                       const float* beta, float eps, float* scale, float* shift, hipStream_t s);
// scale/shift arrays OR a BnRef (bn.acc != nullptr takes precedence)
int bn_apply_relu_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, const float* residual,
                         float* y, long n_pixels, int C, hipStream_t s);
int maxpool3x3s2_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, float* y,
                        int B, int H, int W, int C, hipStream_t s);
// [B,H,W,3] -> zero-bordered [B,H+pt+pb,W+pl+pr,4]
int pad_nhwc3to4_launch(const float* x, float* y, int B, int H, int W, int pt, int pb, int pl, int pr,
                        hipStream_t s);
// uint8 frames -> normalised (x/255 - 0.5) zero-bordered 4-channel fp32
int pad_u8_nhwc3to4_launch(const unsigned char* x, float* y, int B, int H, int W, int pt, int pb, int pl, int pr, hipStream_t s);
// ResNet stem + max-pool fused for inference (stempool.hip): 7x7/2 conv over the zero-bordered 4-channel frame, batch statistics
// of the raw output, and the 3x3/2 pool of the RAW output taken as max or min per channel by the sign of gamma; the elementwise
// BN + ReLU then runs on the pooled tensor.  wp = the stem's packed filter + planes; pooled [B,56,112,64]; stats fp64 [2][64].
int stempool_launch(const float* xpad, const float* wp, const float* gamma, float* pooled, double* stats, int B, hipStream_t s);
// the stem for uint8 frames (stem8.hip): centred bf16 plane of the decoded frame, then conv + statistics + raw pool with one operand plane
size_t stem8_plane_bytes(int B);
int stem8_prep_launch(const unsigned char* x, void* plane, int B, hipStream_t s, float* zero_ptr = nullptr, long zero_n = 0, int half = 0);   // half: fp16 plane (stem8pool_h2_launch)
int stem8pool_h2_launch(const void* plane, const float* wp, const void* wh2, const float* w_inv, const float* gamma, float* pooled, double* stats, int B,
                        hipStream_t s);    // uint8 frames, one fp16 plane x two fp16 filter planes: two products per multiply   // zero_ptr: zero_n floats cleared by the same launch (the batch-norm accumulators)
int stem8pool_launch(const void* plane, const float* wp, const float* gamma, float* pooled, double* stats, int B, hipStream_t s);
// float frames on the same kernel structure (stem8.hip, F16 variant): two fp16 planes of the frame scaled by its exact maximum
int stem16_prep_launch(const float* x, void* planes, float* part, float* a_inv, int B, hipStream_t s, float* zero_ptr = nullptr, long zero_n = 0);
int stem16pool_launch(const void* planes, const void* wh2, const float* gamma, float* pooled, double* stats, const float* a_inv, const float* w_inv,
                      int B, hipStream_t s);
constexpr int STEM16_PARTS = 1024;
int stem8raw_launch(const void* plane, const float* wp, float* y0, double* stats, int B, hipStream_t s);
int stem8rawpool_launch(const void* plane, const float* wp, const float* gamma, float* y0, float* pooled, double* stats, int B, hipStream_t s);   // raw output AND its 3x3/2 pool (max, or min where gamma < 0) in one kernel      // no pool: the raw output (training step)
int assemble_wyzx_launch(const float* audio, const float* yzx, float* out, int B, int snd_size,
                         int snd_contx, int snd_dur, hipStream_t s);
int power_map_launch(const float* ambi, long T, const float* sh, int P, float* rms, hipStream_t s);
// one map per chunk of T samples: ambi [nchunks][T][4] -> rms [nchunks][P]; moments = scratch of nchunks*10 doubles
int power_map_batched_launch(const float* ambi, int nchunks, long T, const float* sh, int P, float* rms, double* moments, hipStream_t s);
// NO_SEPARATION decoder (model.py:274-280, 430): out[b,n,o] = w[b,step,o,0]*mono + bias
int nosep_mix_launch(const float* audio, const float* coeffs, float* out, int B, int snd_size,
                     int snd_contx, int snd_dur, int num_out, hipStream_t s);

// -----------------------------------------------------------------------------------------
// P3 activation planes (p3.hip): [C/16][B*H*(W+1)][3][16] bf16, one zero pixel after every image row
// -----------------------------------------------------------------------------------------
size_t p3_bytes(int B, int H, int W, int C);
// y = [relu](x*scale + shift [+ residual]) -> fp32 NHWC `y` (or null) and / or planes `p3` (or null)
// fmt 0: three bf16 planes (conv3p / conv3g); fmt 1: two fp16 planes of v * 2^ka (conv3h.hip) - ka from the statistics in `h2`
// h2s (256 floats) layout: [0], [1] trunk plane 2^-ka; [2..5] tracked block-input bounds; [6] scratch; [7] saturation counter;
// [H2S_FILTER_FIRST .. H2S_FIXED_FIRST) 2^-kw per fp16x2 filter; then the hard-wired a_inv slots; [2 + H2_RIG_OFF ..] rigorous bounds
constexpr int H2S_FILTER_FIRST = 8;
constexpr int H2S_FIXED_FIRST = 232;    // first slot that is NOT a filter scale: filter slots must stay below (model.hip checks)
constexpr int H2S_A_INV_X = 232;        // [232..235] lean trunk: 2^-ka of the block-input planes, two alternating slots x (video, flow)
constexpr int H2S_A_INV_B = 236;        // [236..237] lean trunk: 2^-ka of the conv_2 input planes (video, flow)
constexpr int H2S_S16_A_INV = 238;      // [238..239] float-frame stem: 2^-ka of the frame planes (video, flow)
constexpr int H2_RIG_OFF = 240;           // the rigorous (Samuelson) bound of a tracked tensor lives this many floats behind its statistical bound
struct P3hScale {
    float* a_inv = nullptr;             // out: 2^-ka
    const float* res_bound = nullptr;   // in: bound of the residual tensor (identity shortcut), or null; [H2_RIG_OFF] behind it: its rigorous bound
    const double* res_acc = nullptr;    // in: fp64 (sum, sumsq) [2][C] of the residual tensor (1x1 shortcut conv), or null
    double res_inv_count = 0.0;
    float* bound_out = nullptr;         // out: bound of the tensor written (null: not tracked); [H2_RIG_OFF] behind it: the rigorous bound (p3.hip)
    unsigned* sat_count = nullptr;      // out: incremented once per element that had to be clamped to +-65000 (stays 0 unless the statistics lie)
    // out (nullable, p3_pack only): the ReLU mask of the tensor written, ONE BIT per element - byte (dense pixel * C/8 + channel octet),
    // bit k = [channel 8*octet + k > 0]: the training step's batch-norm backward reads it instead of the fp32 activation (1/32 of the bytes)
    unsigned char* relu_bits = nullptr;
    // in (p3_pack, fmt 1): the residual as fp16x2 planes of the SAME geometry (hi + lo carries the value conv_1 saw) scaled by
    // res_a_inv[0] = 2^-ka of those planes, instead of an fp32 tensor; may be the very buffer the pass writes (each thread reads its
    // own 8 channels of its own pixel before it overwrites them) - the block output then needs no fp32 copy at all
    const void* res_planes = nullptr;
    const float* res_a_inv = nullptr;
};
int p3_pack_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, const float* residual, int relu,
                   float* y, void* p3, int B, int H, int W, int C, hipStream_t s, int fmt = 0, const P3hScale* h2 = nullptr);
size_t p3h_bytes(int B, int H, int W, int C);
// rows [row0, row0 + R) of every image of an fp32 NHWC tensor (pixel stride ldx, row stride x_rstride, image stride x_bstride, all in
// floats) -> fp16x2 planes [C/16][B*R*(W+1)][2][16] of v * 2^ka with ka from the EXACT maximum the producers' epilogues published
// (IgemmDesc::amax_out: H2_AMAX_SLOTS words each; amax1 nullable): nothing can saturate.  2^-ka -> a_inv[0].
int h2_pack_rows_launch(const float* x, long x_bstride, long x_rstride, int ldx, int row0, int B, int R, int W, int C, const float* amax0,
                        const float* amax1, void* planes, float* a_inv, unsigned* sat_count, hipStream_t s);
// 3x3/2 SAME max-pool of relu(bn(x)) -> fp32 NHWC `y` (or null) and planes `p3` (or null) of the pooled tensor
int p3_maxpool_launch(const float* x, const float* scale, const float* shift, const BnRef& bn, float* y, void* p3, int B, int H,
                      int W, int C, hipStream_t s, int fmt = 0, const P3hScale* h2 = nullptr);

// -----------------------------------------------------------------------------------------
// FFT family (fft.hip)
// -----------------------------------------------------------------------------------------
// fills the twiddle / Hann tables once per device (first call synchronises the stream)
int fft_tables_ensure(hipStream_t s);
int stft_launch(const float* audio, int B, int n_samples, int f0, int f1, float* mag,
                int c0, int c1, float* spec, hipStream_t s, float* zero_ptr = nullptr, int zero_n = 0);      // zero_ptr: zero_n floats cleared by the same launch
// frames: scratch [B][NF][3][1024]; see fft.hip
size_t mask_istft_scratch_bytes(int B);
int mask_istft_mix_launch(const float* dmask, long dmask_bstride, int dmask_f0, const float* spec,
                          const float* coeffs, int B, int ntracks, float* out, float* scratch,
                          hipStream_t s, const float* ebuf = nullptr);
// ebuf (then dmask may be null): the weighted sigmoid sums E[B][23][1024][8] written by the fused deconv1 epilogue (IgemmDesc::mm_out)

// -----------------------------------------------------------------------------------------
// evaluation metrics (eval.hip)
// -----------------------------------------------------------------------------------------
size_t eval_scratch_floats(int B);
int eval_init_launch(float* scratch, hipStream_t s);                  // fills the DFT matrix once
// ps [4][B][3] = per-sample stft distance, lsd, temporal mse, snr;  pw[2] = sum pred^2, sum gt^2 (fp64)
int eval_metrics_launch(const float* pred, const float* gt, int B, float* ps, double* pw, float* scratch, hipStream_t s);

// -----------------------------------------------------------------------------------------
// training-step pieces (train.hip): stft loss + gradient w.r.t. the prediction, fused Adam over a flat bucket
// -----------------------------------------------------------------------------------------
int stft_loss_grad_launch(const float* pred, const float* gt, const float* mask, int B, float* grad, double* loss, hipStream_t s);
int adam_update_launch(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2, float eps,
                       float gscale, hipStream_t s);

// -----------------------------------------------------------------------------------------
// weight gradients (wgrad.hip): dW[t][g][d] = sum_{b,i,j} G[b, i*sh + th*tsh + h0, j*sw + tw*tsw + w0, g] * D[b, i, j, d]
// (conv: G = input, D = dL/dy; conv2d_transpose: G = dL/dy, D = input; FC: one tap on a 1x1 grid).  Strides in floats.
// -----------------------------------------------------------------------------------------
struct WgradDesc {
    const float* g = nullptr;         // gathered operand [B][HG][WG][.. Cg ..], pixel stride ldg, row stride g_rstride
    const float* d = nullptr;         // dense-grid operand [B][Hd][Wd][.. Cd ..], pixel stride ldd, row stride d_rstride
    float* out = nullptr;             // [TH*TW][Cg][Cd]
    float* ws = nullptr;              // [splitk][TH*TW][Cg][Cd] partials when splitk > 1
    int B = 1, Hd = 1, Wd = 1;
    int HG = 1, WG = 1, ldg = 0, Cg = 0;
    int ldd = 0, Cd = 0;
    unsigned g_bstride = 0, g_rstride = 0, d_bstride = 0, d_rstride = 0;
    int sh = 1, sw = 1, TH = 1, TW = 1, h0 = 0, w0 = 0;
    int tsh = 1, tsw = 1;             // tap strides: tap (th, tw) reads G at (i*sh + th*tsh + h0, j*sw + tw*tsw + w0)
    int splitk = 1;
    int defer_reduce = 0;             // 1: wgrad_launch leaves the partials in ws; the caller runs wgrad_reduce_launch (timed separately)
    // both operands ALSO as fp16x2 planes [C/16][B*Hd*(Wd+1)][2][16] (h2_planes.h) with their 2^-k scales: a dense 3x3 stride-1 SAME
    // layer then runs wgrad3h_kernel (wgrad3h.hip: three fp16 products per multiply, LDS-DMA fed)
    const void* gp = nullptr;
    const void* dp = nullptr;
    const float* gp_a_inv = nullptr;
    const float* dp_a_inv = nullptr;
    unsigned gp_bytes = 0, dp_bytes = 0;      // (filled by wgrad_launch)
    // filled by wgrad_launch
    int P = 0, fold = 1;
    unsigned g_bytes = 0, d_bytes = 0, magic_w = 0, magic_h = 0;
};
int wgrad_launch(const WgradDesc& d, hipStream_t s);
int wgrad_pick_splitk(const WgradDesc& d, size_t ws_capacity_floats);
int wgrad_reduce_launch(const WgradDesc& d, hipStream_t s);
// "wgrad3h_kernel" (fp16x2 planes of both operands, one filter row per workgroup) |
// "wgrad3r_kernel" (bf16x3, one filter row per workgroup: dense 3x3 stride-1) | "wgrad3_kernel" (bf16x3, one tap) |
// "wgrad_kernel" (exact fp32 MFMA) | "wgrad_ref_kernel"
const char* wgrad_kernel_name(const WgradDesc& d);
bool wgrad_reads_fp32_operands();     // SAGEN_WGRAD_REF: the fp32 dy / activations must exist even where the planes do
bool wgrad_planes_enabled();          // wgrad3h_kernel not switched off by the environment (exact fp32 / reference / SAGEN_WGRAD_NO_H2)

// -----------------------------------------------------------------------------------------
// backward elementwise / reductions (backward.hip)
// -----------------------------------------------------------------------------------------
// scratch of the per-channel reductions below: per-workgroup partial rows, summed in a fixed order (no atomics)
size_t reduce_scratch_floats(int C);
// dy[r][c] = (ga[r][c] + gb[r][c]) * (act[r][c] > 0); gb / act / dy nullable; colsum (fp64 [C]) nullable: sum_r dy (overwritten);
// scratch >= reduce_scratch_floats(C) floats (without it, or for widths that do not divide 256 lanes: fp64 atomics)
// colsum_f32 (nullable): the sums also as fp32 (the bias gradient at its place in the gradient bucket)
int relu_bwd_launch(const float* ga, int lda, const float* gb, int ldb, const float* act, int ldact, float* dy, int lddy, long R,
                    int C, double* colsum, float* scratch, hipStream_t s, float* colsum_f32 = nullptr,
                    long band_rows = 0, long batch_rows = 0, long row0 = 0);      // band_rows > 0: row r stands for row (r / band_rows) * batch_rows + row0 + r % band_rows of every operand
// training-mode batch-norm backward (core.py:6,209-210 under tf.gradients).  dz = (ga + gb) * (act > 0) (act nullable);
// xhat = (y - mean) * invstd from the forward accumulators `bn`;  acc[2][C] (fp64) = (sum dz, sum dz*xhat) (overwritten);
// scratch >= reduce_scratch_floats(C) floats
int bn_bwd_reduce_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, long n_pixels, int C,
                         double* acc, float* scratch, hipStream_t s, int self_mask = 0, float* mx_part = nullptr, int* mx_blocks = nullptr,
                         const unsigned char* relu_bits = nullptr);
// relu_bits (nullable; act must be null then): the ReLU mask as one bit per element (P3hScale::relu_bits) instead of the activation
// mx_part (nullable, >= 2 * 512 floats): per-workgroup (max |dz|, max |xhat|); *mx_blocks = the number of workgroups that wrote it
// dy = gamma*invstd*(dz - acc0/N - xhat*acc1/N) (+ dz to `dz_out`, nullable); writes dgamma = acc1, dbeta = acc0 (nullable)
int bn_bwd_apply_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, const double* acc,
                        long n_pixels, int C, float* dy, float* dz_out, float* dgamma, float* dbeta, hipStream_t s, int self_mask = 0);
// ... and dy also as fp16x2 planes [C/16][B*H*(W+1)][2][16] of dy * 2^kd for conv3h_kernel / wgrad3h_kernel, 2^-kd to a_inv[0]; the
// bound behind kd comes from the reduce pass's mx_part; dy may be null then (no fp32 copy)
int bn_bwd_apply_h2_launch(const float* ga, const float* gb, const float* act, const float* y, const BnRef& bn, const double* acc,
                           int B, int H, int W, int C, float* dy, float* dz_out, float* dgamma, float* dbeta, hipStream_t s, int self_mask,
                           void* planes, const float* mx_part, int mx_blocks, float* a_inv, unsigned* sat_count,
                           const unsigned char* relu_bits = nullptr);
// self_mask = 1 (act must be null): the ReLU mask is relu(bn(y)) > 0, re-derived from y with the forward's own scale / shift - for a
// layer whose activation IS relu(bn(y)) (no residual), e.g. conv_1 of a residual block: one tensor less to read in both passes
// backward of maxpool3x3s2(relu(bn(y0))) (resnet.py:134-135): dz0[b,i,j,c] = (a > 0) * sum over the windows containing (i,j) of
// [a == pooled] * (ga + gb), a = relu(bn(y0)); pooled = the forward's pool output [B,Ho,Wo,C]
int maxpool_bwd_launch(const float* y0, const BnRef& bn, const float* pooled, const float* ga, const float* gb, float* dz,
                       int B, int H, int W, int C, hipStream_t s);
// the same followed by the stem's batch-norm backward, fused: dy0 = d/dy0 of maxpool(relu(bn0(y0))), gamma / beta gradients;
// acc: fp64 [2][C] (overwritten), scratch >= reduce_scratch_floats(1024) floats (4096 partial rows of 2C, C <= 128)
int maxpool_bn_bwd_launch(const float* y0, const BnRef& bn, const float* pooled, const float* ga, const float* gb, float* dy0, int B,
                          int H, int W, int C, double* acc, float* scratch, float* dgamma, float* dbeta, hipStream_t s,
                          const float* pooled_raw = nullptr);      // pooled_raw: each window's raw extremum (stem8rawpool_launch) - the sums then run over the pooled grid
// out[m][c] = sum_{r < rep} (ina[(m*rep + r)*lda + c] + inb[(m*rep + r)*ldb + c]); inb nullable (backward of tf.tile / concat fan-in)
int sum_rows_launch(const float* ina, int lda, const float* inb, int ldb, int rep, long M, int C, float* out, int ldo, hipStream_t s);
int acc_to_f32_launch(const double* acc, float* dst, int n, hipStream_t s);
// contrib batch_norm moving averages (UPDATE_OPS, train.py:147-148): m <- decay*m + (1-decay)*batch stat
int bn_moving_update_launch(const BnRef& bn, float* moving_mean, float* moving_var, int C, float decay, hipStream_t s);
constexpr int BN_MOVING_MAX_JOBS = 40;
struct BnMovingJob { const double* acc = nullptr; double inv_count = 0.0; float* mm = nullptr; float* mv = nullptr; int C = 0; };
int bn_moving_update_multi_launch(const BnMovingJob* jobs, int n, float decay, hipStream_t s);
// filter packs for the data-gradient contractions
// 3x3 (any kh x kw) stride-1 conv: Wp[n = ci][(tap', co)] = W_hwio[ntaps-1-tap'][ci][co]
int pack_conv_flipT_launch(const float* w_hwio, int ntaps, int cin, int cout, float* wp, int Kpad, hipStream_t s);
// row-major matrix [rows][cols] (ld = cols) -> [rows][Kpad] zero padded (FC: dx = dy . W^T)
int pack_rows_launch(const float* w, int rows, int cols, float* wp, int Kpad, hipStream_t s);
// stem: [7][8*4][64] (tap row, (tw, c4), o) -> HWIO [7,7,3,64]
int stem_wgrad_unpack_launch(const float* tmp, float* dw, hipStream_t s);

// adjoint of mask_istft_mix (fft.hip): dpred [B,4800,3] -> dmask (same geometry as dmask, frames MASK_F_LO..), dcoeffs [B*3][ldc]
size_t mask_istft_bwd_scratch_floats(int B, int ntracks);
int mask_istft_mix_bwd_launch(const float* dmask, long dmask_bstride, int dmask_f0, const float* spec, const float* coeffs,
                              const float* dpred, int B, int ntracks, float* d_dmask, long dd_bstride, int dd_f0, float* dcoeffs,
                              int ldc, float* scratch, hipStream_t s);

}  // namespace sagen
