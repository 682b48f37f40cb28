// xh_common.h -- shared internals of libxmipp_hip.so (gfx950 only).
#ifndef XH_COMMON_H
#define XH_COMMON_H
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/xmipp_hip.h"

void xh_set_error(const char *fmt, ...);

#define XH_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            xh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                       \
            return XH_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define XH_CHECK(cond, code, ...)       \
    do {                                \
        if (!(cond)) {                  \
            xh_set_error(__VA_ARGS__);  \
            return code;                \
        }                               \
    } while (0)

#define XH_TRY(call)              \
    do {                          \
        int r_ = (call);          \
        if (r_ != XH_OK) return r_; \
    } while (0)

#define XH_LAUNCH_CHECK() XH_HIP(hipGetLastError())

struct xh_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int num_cus;
};

// RAII-less tracked device allocation helper for handles
struct XhBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
int xh_buf_alloc(xh_ctx *ctx, XhBuf &b, size_t bytes);
void xh_buf_free(XhBuf &b);
// grow-only scratch
int xh_buf_reserve(xh_ctx *ctx, XhBuf &b, size_t bytes);

// a grow-only device buffer that lives with a 2-D transform plan (frame-after-frame callers: dose filter, binning)
int xh_fft2d_user_scratch(xh_fft2d *f, size_t bytes, void **p);
int xh_fft2d_rows_of_real_pairs(xh_fft2d *f, const float *d_frame, const float *d_dark, const float *d_gain, int Y, float *d_work, int *n1, int *n2);
int xh_fft2d_rows_of_real_pairs_kept(xh_fft2d *f, const float *d_frame, const float *d_dark, const float *d_gain, int Y, int nc, float *d_C, int *done);

// A/B and test knobs read from the environment (XH_FA_PB, XH_FA_COPY_PATCHES, XH_FA_WARP_PLAIN, XH_PREFILTER_FORM, XH_FFT2D_NO_SMALL,
// XH_FFT2D_N1, XH_FFT2D_NO_45: other correct forms of a kernel; XH_ES_ORDER: one half of xh_iterative_alignment's compute(), a WRONG
// result by design, for tools/diag_iterative.py) exist only in a library built with -DXH_DEBUG_HOOKS (XH_DEBUG_HOOKS=1 csrc/build.sh);
// the product build never looks at them.  XH_ALLOC_TRACE and XH_FA_TIMING only print.
static inline const char *xh_debug_env(const char *name)
{
#ifdef XH_DEBUG_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

static inline int xh_ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}
static inline bool xh_is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

#endif
