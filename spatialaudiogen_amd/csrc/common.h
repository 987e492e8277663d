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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ilog2_exact(int x) {   // -1 if not a power of two
    if (x <= 0 || (x & (x - 1))) return -1;
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

}  // namespace sagen
