// Shared host-side helpers of libsagen_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/sagen.h"

namespace sagen {

char* err_buf();                       // thread-local message buffer (api.hip)
int   fail(int code, const char* fmt, ...);

#define SAGEN_HIP_CHECK(expr)                                                                   \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::sagen::fail(SAGEN_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,    \
                                 hipGetErrorString(e_));                                        \
    } while (0)

#define SAGEN_LAUNCH_CHECK() SAGEN_HIP_CHECK(hipGetLastError())

// ---- grouped launch (round 6: sagen_forward_grouped) ------------------------------------------------------------------------
// G INDEPENDENT batches of B windows run as ONE launch per layer: every kernel of the inference path takes the group index from
// its grid (blockIdx.z; igemm_kernel / igemm3_kernel: blockIdx.z / splitk) and sees exactly the single-batch problem - same tiles,
// same statistics, same scales, hence bit-identical results per group.  What differs per group is WHERE its tensors live: the
// workspace holds the per-batch region (activations, batch-norm accumulators, plane scales, split-K scratch, ...) G times at a
// constant stride, and a pointer that falls into group 0's copy [lo, lo + span) is moved g * stride further on (grp_ptr).  Pointers
// outside that range - packed filters, the caller's variables, null - are shared and stay.  The caller's own arrays (audio, frames,
// output) hold the G batches back to back and are moved by the explicit element strides of the three kernels that touch them.
struct GroupInfo {
    const char* lo = nullptr;
    unsigned long long span = 0, stride = 0;       // bytes
    int G = 1;
};
template <class T>
__device__ __forceinline__ T* grp_ptr(T* p, const GroupInfo& gi, int g) {
    const unsigned long long d = (unsigned long long)((const char*)p - gi.lo);
    return d < gi.span ? (T*)((char*)p + (unsigned long long)g * gi.stride) : p;
}
// in a kernel with a `const GroupInfo gi` parameter: pointer p of THIS workgroup's group (blockIdx.z)
#define SAGEN_GRP(p) (gi.G > 1 ? ::sagen::grp_ptr((p), gi, (int)blockIdx.z) : (p))
// the group the launches of THIS thread are issued for (set by sagen_forward_impl around a grouped forward; G = 1 otherwise: the op
// level, the training step and ungrouped contexts never see a group dimension)
GroupInfo& cur_group();

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ilog2_exact(int x) {   // -1 if not a power of two
    if (x <= 0 || (x & (x - 1))) return -1;
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

}  // namespace sagen
