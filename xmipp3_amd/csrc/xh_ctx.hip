// xh_ctx.hip -- context, memory and timer entry points of the C ABI.
#include "xh_common.h"
#include <cstring>
#include <cctype>

static thread_local std::string g_err;

void xh_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int xh_buf_alloc(xh_ctx *ctx, XhBuf &b, size_t bytes)
{
    xh_buf_free(b);
    if (bytes == 0) return XH_OK;
    XH_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        xh_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return XH_ERR_NOMEM;
    }
    b.bytes = bytes;
    if (getenv("XH_ALLOC_TRACE")) fprintf(stderr, "xh_buf_alloc %zu bytes -> %p\n", bytes, b.p);
    return XH_OK;
}
void xh_buf_free(XhBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
int xh_buf_reserve(xh_ctx *ctx, XhBuf &b, size_t bytes)
{
    if (b.bytes >= bytes) return XH_OK;
    // the stream may still be using the old buffer
    XH_HIP(hipStreamSynchronize(ctx->stream));
    return xh_buf_alloc(ctx, b, bytes);
}

extern "C" {

const char *xh_last_error(void) { return g_err.c_str(); }
const char *xh_version(void) { return "xmipp3_amd 0.1 (gfx950)"; }

int xh_device_count(int *count)
{
    XH_CHECK(count, XH_ERR_ARG, "xh_device_count: null pointer");
    XH_HIP(hipGetDeviceCount(count));
    return XH_OK;
}

int xh_device_numa_node(int device, int *node)
{
    XH_CHECK(node, XH_ERR_ARG, "xh_device_numa_node: null pointer");
    *node = -1;
    char bdf[64] = {0};
    XH_HIP(hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device));
    for (char *c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
    std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return XH_OK;
    int v = -1;
    if (fscanf(f, "%d", &v) == 1) *node = v;
    fclose(f);
    return XH_OK;
}

static int ctx_create(int device, void *stream, bool own, xh_ctx **out)
{
    XH_CHECK(out, XH_ERR_ARG, "xh_ctx_create: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        xh_set_error("xh_ctx_create: no usable HIP device (%s); this library has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return XH_ERR_HIP;
    }
    XH_CHECK(device >= 0 && device < n, XH_ERR_ARG, "xh_ctx_create: device %d out of range [0,%d)",
             device, n);
    XH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    XH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        xh_set_error("xh_ctx_create: device %d is %s; this library ships gfx950 (MI355X) code only",
                     device, prop.gcnArchName);
        return XH_ERR_UNSUPPORTED;
    }
    xh_ctx *c = new xh_ctx;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->stream = (hipStream_t)stream;   // NULL = the device's default (null) stream
    c->own_stream = false;
    if (own) {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            xh_set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
            return XH_ERR_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return XH_OK;
}

int xh_ctx_create(int device, void *stream, xh_ctx **out) { return ctx_create(device, stream, false, out); }
int xh_ctx_create_private(int device, xh_ctx **out) { return ctx_create(device, nullptr, true, out); }

int xh_ctx_destroy(xh_ctx *ctx)
{
    if (!ctx) return XH_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return XH_OK;
}

int xh_ctx_sync(xh_ctx *ctx)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    return XH_OK;
}
void *xh_ctx_stream(xh_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int xh_malloc(xh_ctx *ctx, size_t bytes, void **d_ptr)
{
    XH_CHECK(ctx && d_ptr, XH_ERR_ARG, "xh_malloc: null argument");
    XH_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) {
        xh_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return XH_ERR_NOMEM;
    }
    return XH_OK;
}
int xh_free(xh_ctx *ctx, void *d_ptr)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    XH_HIP(hipFree(d_ptr));
    return XH_OK;
}
int xh_memset(xh_ctx *ctx, void *d_ptr, int value, size_t bytes)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return XH_OK;
}
int xh_memcpy_h2d(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    return XH_OK;
}
int xh_memcpy_d2h(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    return XH_OK;
}

int xh_host_alloc(xh_ctx *ctx, size_t bytes, void **h_ptr)
{
    XH_CHECK(ctx && h_ptr, XH_ERR_ARG, "xh_host_alloc: null argument");
    XH_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *h_ptr = nullptr;
        xh_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return XH_ERR_NOMEM;
    }
    return XH_OK;
}
int xh_host_free(xh_ctx *ctx, void *h_ptr)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    if (!h_ptr) return XH_OK;
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    XH_HIP(hipHostFree(h_ptr));
    return XH_OK;
}
int xh_memcpy_h2d_async(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return XH_OK;
}
int xh_memcpy_d2h_async(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    XH_CHECK(ctx, XH_ERR_ARG, "null context");
    XH_HIP(hipSetDevice(ctx->device));
    XH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return XH_OK;
}
int xh_ctx_wait_for(xh_ctx *ctx, xh_ctx *other)
{
    XH_CHECK(ctx && other, XH_ERR_ARG, "null context");
    XH_CHECK(ctx->device == other->device, XH_ERR_ARG, "xh_ctx_wait_for: contexts on devices %d and %d", ctx->device, other->device);
    if (ctx->stream == other->stream) return XH_OK;
    XH_HIP(hipSetDevice(ctx->device));
    hipEvent_t ev;
    XH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, other->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
    (void)hipEventDestroy(ev);      // released when the recorded work has completed
    XH_HIP(e);
    return XH_OK;
}

struct XhTimer {
    hipEvent_t a, b;
};
int xh_timer_create(xh_ctx *ctx, void **timer)
{
    XH_CHECK(ctx && timer, XH_ERR_ARG, "null argument");
    XH_HIP(hipSetDevice(ctx->device));
    XhTimer *t = new XhTimer;
    XH_HIP(hipEventCreate(&t->a));
    XH_HIP(hipEventCreate(&t->b));
    *timer = t;
    return XH_OK;
}
int xh_timer_start(xh_ctx *ctx, void *timer)
{
    XH_HIP(hipEventRecord(((XhTimer *)timer)->a, ctx->stream));
    return XH_OK;
}
int xh_timer_stop(xh_ctx *ctx, void *timer)
{
    XH_HIP(hipEventRecord(((XhTimer *)timer)->b, ctx->stream));
    return XH_OK;
}
int xh_timer_elapsed_ms(xh_ctx *ctx, void *timer, float *ms)
{
    XhTimer *t = (XhTimer *)timer;
    XH_HIP(hipEventSynchronize(t->b));
    XH_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return XH_OK;
}
int xh_timer_destroy(xh_ctx *ctx, void *timer)
{
    XhTimer *t = (XhTimer *)timer;
    if (!t) return XH_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return XH_OK;
}

void xh_ctf_defaults(xh_ctf_params *p)
{
    // data/ctf.cpp:365-388 (readFromMdRow defaults)
    memset(p, 0, sizeof(*p));
    p->Tm = 1;
    p->kV = 100;
    p->K = 1;
}
}
