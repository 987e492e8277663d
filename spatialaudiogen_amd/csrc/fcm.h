// fcm_kernel (fcm.hip): the skinny fully-connected layers of the bottleneck / localisation / separation heads (tfw.fully_connected,
// core.py:43-93 as model.py:203-256, 287-294 use it) - M = batch x 3 time steps <= 96 rows against [K][N] matrices of up to 6.4 M
// weights.  See fcm.hip.
#pragma once
#include "common.h"

namespace sagen {

constexpr int FCM_MAX_SEG = 8, FCM_MAX_JOBS = 2, FCM_MAX_M = 96;

// one K range [k0, k1) of the input rows and where its values come from
struct FcmSeg {
    const float* p = nullptr;     // nsplit == 0: x(m, k) = p[(m / row_div) * ld + (k - k0)]
    int k0 = 0, k1 = 0, ld = 0;   // nsplit > 0: x(m, k) = act(bias[k - k0] + sum_z p[z * zstride + (m / row_div) * ld + (k - k0)])  (a producer's partials)
    int nsplit = 0;
    long zstride = 0;
    int row_div = 1;              // tf.tile of the producer's rows (model.py:230-232: one visual feature row per window, three time steps)
    int relu = 0;
    const float* bias = nullptr;
};
struct FcmJob {
    const float* w = nullptr;     // the variable itself, TF layout [K][N] (no pack)
    int N = 0;
    float* out = nullptr;         // partials [nslices][M][N] (summed, + bias / activation, by the consumer or by splitk_reduce_kernel)
};
struct FcmDesc {
    int M = 0, K = 0, nseg = 0, njobs = 0, nslices = 1;
    FcmSeg seg[FCM_MAX_SEG];
    FcmJob job[FCM_MAX_JOBS];
};
int fcm_pick_slices(int K, long weights);     // split of K over workgroups: 0 if the shape does not qualify (K % 8)
int fcm_launch(const FcmDesc& d, hipStream_t s);

}  // namespace sagen
