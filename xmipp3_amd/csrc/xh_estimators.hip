// xh_estimators.hip -- first slice of the batched estimator API: the extrema finder and the shift estimator
// by correlation, the two building blocks xmipp_cuda_align_significant and FlexAlign share.
//
//   ExtremaFinder::SingleExtremaFinder<T>   reconstruction/single_extrema_finder.cpp:146-300 (CUDA twin: reconstruction_cuda/
//                                           cuda_single_extrema_finder.cpp): Max, Lowest, MaxAroundCenter, LowestAroundCenter of
//                                           n signals, positions as element offsets (float) and values
//   Alignment::ShiftCorrEstimator<T>        reconstruction/shift_corr_estimator.cpp:33-300 (CUDA twin: reconstruction_cuda/
//                                           cuda_shift_corr_estimator.cpp), AlignType::OneToN: correlation of n spectra with one
//                                           reference spectrum (optionally centred), shifts of n images against one reference
//                                           = position of the correlation maximum within maxShift of the centre
//   Alignment::PolarRotationEstimator<T>    reconstruction/polar_rotation_estimator.cpp:33-144, AlignType::OneToN: rotation of n images
//                                           against one reference = arg-max of the rotational correlation of their polar Fourier
//                                           transforms over the rings firstRing .. lastRing (data/polar.cpp:99-148,212-231)
// The iterative alignment estimator is not built.
#include <cmath>
#include <cstring>
#include <vector>

#include "xh_common.h"

namespace {
typedef float2 es_cf;

// first extremum in element order: std::max_element / std::min_element return the first of equals, the search around the centre
// replaces its candidate on a strict comparison only
template <bool LOWEST>
__global__ void __launch_bounds__(256) k_es_extrema(const float *__restrict__ data, size_t elems, int ydim, int xdim, int around, int maxDist, int empty,
                                                    float *__restrict__ pos, float *__restrict__ val)
{
    __shared__ float sv[256];
    __shared__ long long si[256];
    const float *d = data + (size_t)blockIdx.x * elems;
    const float start = LOWEST ? 3.402823466e+38f : -3.402823466e+38f;
    float best = start;
    long long bi = -1;
    if (!around) {
        for (size_t i = threadIdx.x; i < elems; i += 256) {
            const float v = d[i];
            // element 0 always becomes the candidate (max_element starts from it), later ones only when strictly better
            if (bi < 0 || (LOWEST ? v < best : v > best)) { best = v; bi = (long long)i; }
        }
    } else if (!empty) {
        const int xHalf = xdim / 2, yHalf = ydim / 2;
        const int x0 = max(0, xHalf - maxDist), x1 = min(xdim - 1, xHalf + maxDist), y0 = max(0, yHalf - maxDist), y1 = min(ydim - 1, yHalf + maxDist);
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        for (int t = threadIdx.x; t < w * h; t += 256) {
            const int y = y0 + t / w, x = x0 + t % w;
            const int ly = y - yHalf, lx = x - xHalf;
            if (ly * ly + lx * lx > maxDist * maxDist) continue;
            const float v = d[(size_t)y * xdim + x];
            if (LOWEST ? v < best : v > best) { best = v; bi = (long long)y * xdim + x; }
        }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v = sv[threadIdx.x + o];
            const long long k = si[threadIdx.x + o];
            const bool mine = si[threadIdx.x] >= 0;
            if (k >= 0 && (!mine || (LOWEST ? v < sv[threadIdx.x] : v > sv[threadIdx.x]) || (v == sv[threadIdx.x] && k < si[threadIdx.x]))) { sv[threadIdx.x] = v; si[threadIdx.x] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (pos) pos[blockIdx.x] = (float)si[0];
        if (val) val[blockIdx.x] = si[0] >= 0 ? sv[0] : start;
    }
}

// sComputeCorrelations2DOneToN (shift_corr_estimator.cpp:163-199): inOut = ref conj(inOut), times (-1)^(x+y) when centred
__global__ void __launch_bounds__(256) k_es_correlate(es_cf *__restrict__ inOut, const es_cf *__restrict__ ref, size_t per, int xdim, size_t total, int center)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const size_t e = t % per;
    const int y = (int)(e / xdim), x = (int)(e - (size_t)y * xdim);
    const es_cf r = ref[e], o = inOut[t];
    es_cf v = es_cf{r.x * o.x + r.y * o.y, r.y * o.x - r.x * o.y};
    if (center && ((x + y) & 1)) { v.x = -v.x; v.y = -v.y; }
    inOut[t] = v;
}

__global__ void __launch_bounds__(256) k_es_to_complex(const float *__restrict__ in, es_cf *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < tot) out[t] = es_cf{in[t], 0.f};
}

__global__ void __launch_bounds__(256) k_es_real(const es_cf *__restrict__ in, float *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < tot) out[t] = in[t].x;
}
}  // namespace

struct xh_shiftcorr {
    xh_ctx *ctx;
    int x, y, maxShift;
    xh_fft2d *plan;
    XhBuf ref, work, map, pos;
    bool refLoaded;
};

extern "C" {

int xh_extrema_find(xh_ctx *ctx, const float *d_data, int32_t n, int32_t zdim, int32_t ydim, int32_t xdim, int32_t search_type, float max_dist, float *h_positions,
                    float *h_values)
{
    XH_CHECK(ctx && d_data && n >= 1 && zdim >= 1 && ydim >= 1 && xdim >= 1 && (h_positions || h_values), XH_ERR_ARG, "xh_extrema_find: bad argument");
    XH_CHECK(search_type >= 0 && search_type <= 3, XH_ERR_ARG, "xh_extrema_find: search type %d (0 Max, 1 Lowest, 2 MaxAroundCenter, 3 LowestAroundCenter)", search_type);
    const int around = search_type >= 2;
    if (around) {
        XH_CHECK(zdim == 1 && ydim > 1, XH_ERR_UNSUPPORTED, "xh_extrema_find: the search around the centre is for 2-D signals (\"Not implemented\", single_extrema_finder.cpp:97-105)");
        XH_CHECK(max_dist > 0, XH_ERR_ARG, "xh_extrema_find: the maximal distance from the centre must be positive");
    }
    XH_HIP(hipSetDevice(ctx->device));
    const int maxDist = (int)max_dist;                      // size_t maxDist of sFindUniversal2DAroundCenter
    // xHalf - maxDist is unsigned in the reference: a distance beyond the centre's coordinate wraps, and nothing is searched
    const int empty = around && (maxDist > xdim / 2 || maxDist > ydim / 2);
    XhBuf bPos, bVal;
    XH_TRY(xh_buf_alloc(ctx, bPos, sizeof(float) * n));
    int rc = xh_buf_alloc(ctx, bVal, sizeof(float) * n);
    if (rc == XH_OK) {
        const size_t elems = (size_t)zdim * ydim * xdim;
        if (search_type & 1)
            hipLaunchKernelGGL((k_es_extrema<true>), dim3(n), dim3(256), 0, ctx->stream, d_data, elems, ydim, xdim, around, maxDist, empty, (float *)bPos.p, (float *)bVal.p);
        else
            hipLaunchKernelGGL((k_es_extrema<false>), dim3(n), dim3(256), 0, ctx->stream, d_data, elems, ydim, xdim, around, maxDist, empty, (float *)bPos.p, (float *)bVal.p);
        if (hipGetLastError() != hipSuccess) rc = XH_ERR_HIP;
        if (rc == XH_OK && h_positions && hipMemcpyAsync(h_positions, bPos.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        if (rc == XH_OK && h_values && hipMemcpyAsync(h_values, bVal.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    }
    xh_buf_free(bPos); xh_buf_free(bVal);
    if (rc == XH_ERR_HIP) xh_set_error("xh_extrema_find: device error");
    return rc;
}

int xh_shiftcorr_destroy(xh_shiftcorr *h)
{
    if (!h) return XH_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->plan) xh_fft2d_destroy(h->plan);
    xh_buf_free(h->ref); xh_buf_free(h->work); xh_buf_free(h->map); xh_buf_free(h->pos);
    delete h;
    return XH_OK;
}

int xh_shiftcorr_create(xh_ctx *ctx, int32_t xdim, int32_t ydim, int32_t max_shift, xh_shiftcorr **out)
{
    XH_CHECK(ctx && out && xdim >= 2 && ydim >= 2, XH_ERR_ARG, "xh_shiftcorr_create: bad argument");
    // the centring of the correlation by (-1)^(x+y) needs even sizes (computeShifts2DOneToN asserts them), the search a maximal
    // shift sharply below half of the size (AShiftCorrEstimator::check)
    XH_CHECK((xdim & 1) == 0 && (ydim & 1) == 0, XH_ERR_ARG, "xh_shiftcorr_create: only even sizes are supported");
    XH_CHECK(max_shift > 0 && max_shift < xdim / 2 && max_shift < ydim / 2, XH_ERR_ARG, "xh_shiftcorr_create: the maximal shift must be positive and sharply less than half of the size");
    XH_HIP(hipSetDevice(ctx->device));
    xh_shiftcorr *h = new xh_shiftcorr;
    h->ctx = ctx; h->x = xdim; h->y = ydim; h->maxShift = max_shift; h->plan = nullptr; h->refLoaded = false;
    int rc = xh_fft2d_create(ctx, ydim, xdim, &h->plan);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->ref, sizeof(es_cf) * (size_t)xdim * ydim);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->work, sizeof(es_cf) * (size_t)xdim * ydim);
    if (rc != XH_OK) { xh_shiftcorr_destroy(h); return rc; }
    *out = h;
    return XH_OK;
}

// load2DReferenceOneToN(const T *ref) (:52-62): the reference image [y][x]; its full spectrum is kept
int xh_shiftcorr_load_reference(xh_shiftcorr *h, const float *d_ref)
{
    XH_CHECK(h && d_ref, XH_ERR_ARG, "xh_shiftcorr_load_reference: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)h->x * h->y;
    hipLaunchKernelGGL(k_es_to_complex, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_ref, (es_cf *)h->ref.p, tot);
    XH_LAUNCH_CHECK();
    XH_TRY(xh_fft2d_exec(h->plan, (float *)h->ref.p, 0));
    h->refLoaded = true;
    return XH_OK;
}

// computeCorrelations2DOneToN(hw, inOut, ref, dims, center) (:143-161), the static form: n spectra [n][fy][fx] complex against one
int xh_shiftcorr_correlate(xh_ctx *ctx, float *d_inout, const float *d_ref, int32_t n, int32_t fy, int32_t fx, int32_t center)
{
    XH_CHECK(ctx && d_inout && d_ref && n >= 1 && fy >= 1 && fx >= 1, XH_ERR_ARG, "xh_shiftcorr_correlate: bad argument");
    XH_CHECK(!center || (fy & 1) == 0, XH_ERR_ARG, "xh_shiftcorr_correlate: centring needs an even number of rows");
    XH_HIP(hipSetDevice(ctx->device));
    const size_t per = (size_t)fy * fx, total = per * n;
    hipLaunchKernelGGL(k_es_correlate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (es_cf *)d_inout, (const es_cf *)d_ref, per, fx, total, center);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

// computeShift2DOneToN (:201-246) + computeShifts2DOneToN (:248-283): n images [n][y][x] -> h_shifts [n][2] = (x, y) of the
// correlation maximum within maxShift of the centre, as the reference returns it (the shift of the image is its negative)
int xh_shiftcorr_compute_shifts(xh_shiftcorr *h, const float *d_others, int32_t n, float *h_shifts)
{
    XH_CHECK(h && d_others && n >= 1 && h_shifts, XH_ERR_ARG, "xh_shiftcorr_compute_shifts: bad argument");
    XH_CHECK(h->refLoaded, XH_ERR_STATE, "xh_shiftcorr_compute_shifts: Not ready to execute. Call init() before (no reference loaded)");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)h->x * h->y;
    XH_TRY(xh_buf_reserve(ctx, h->map, sizeof(float) * tot * (size_t)n));
    XH_TRY(xh_buf_reserve(ctx, h->pos, sizeof(float) * (size_t)n));
    const unsigned grid = (unsigned)((tot + 255) / 256);
    for (int i = 0; i < n; ++i) {
        es_cf *w = (es_cf *)h->work.p;
        hipLaunchKernelGGL(k_es_to_complex, dim3(grid), dim3(256), 0, ctx->stream, d_others + (size_t)i * tot, w, tot);
        XH_TRY(xh_fft2d_exec(h->plan, (float *)w, 0));
        hipLaunchKernelGGL(k_es_correlate, dim3(grid), dim3(256), 0, ctx->stream, w, (const es_cf *)h->ref.p, tot, h->x, tot, 1);
        XH_TRY(xh_fft2d_exec(h->plan, (float *)w, 1));
        hipLaunchKernelGGL(k_es_real, dim3(grid), dim3(256), 0, ctx->stream, (const es_cf *)w, (float *)h->map.p + (size_t)i * tot, tot);
        XH_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((k_es_extrema<false>), dim3(n), dim3(256), 0, ctx->stream, (const float *)h->map.p, tot, h->y, h->x, 1, h->maxShift, 0, (float *)h->pos.p, (float *)nullptr);
    XH_LAUNCH_CHECK();
    std::vector<float> pos(n);
    XH_HIP(hipMemcpyAsync(pos.data(), h->pos.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    const int cX = h->x / 2, cY = h->y / 2;
    for (int i = 0; i < n; ++i) {
        h_shifts[2 * i] = (float)(((int)pos[i] % h->x) - cX);
        h_shifts[2 * i + 1] = (float)(((int)pos[i] / h->x) - cY);
    }
    return XH_OK;
}

// PolarRotationEstimator::load2DReferenceOneToN + computeRotation2DOneToN: best_rotation(reference, image) for n square images.
// Runs on the projection matcher (one reference, rings first_ring .. last_ring, the mirrored particle switched off), i.e. with the
// matcher's cubic B-spline polar sampling and its exact arg-max; the reference samples the rings with BsplineOrder 1
// (polar_rotation_estimator.cpp:60,99) -- the correlation values differ in the last digits, the angle by at most one sample of the
// outer ring in near-ties. best_rotation correlates (reference, image) where the matcher correlates (image, reference):
// the index is mirrored, result = ((N - psi) mod N) 360 / N degrees.
int xh_rotation_estimate(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t D, int32_t first_ring, int32_t last_ring, float *h_rotations)
{
    XH_CHECK(ctx && d_ref && d_others && h_rotations && n >= 1, XH_ERR_ARG, "xh_rotation_estimate: bad argument");
    // RotationEstimationSetting::check + PolarRotationEstimator::check (arotation_estimator.h:80-130, polar_rotation_estimator.cpp:125-141)
    XH_CHECK(D >= 6, XH_ERR_ARG, "xh_rotation_estimate: The input signal is too small.");
    XH_CHECK(first_ring >= 1 && last_ring > first_ring && last_ring < D, XH_ERR_ARG, "xh_rotation_estimate: rings %d .. %d of a %d px image (first >= 1, last > first, last < size)",
             first_ring, last_ring, D);
    XH_CHECK(last_ring <= D / 2 - 1, XH_ERR_ARG, "xh_rotation_estimate: the last ring (%d) needs an edge around it: at most %d for %d px", last_ring, D / 2 - 1, D);
    xh_pm *pm = nullptr;
    XH_TRY(xh_pm_create(ctx, D, first_ring, last_ring, 1, d_ref, nullptr, 0, &pm));
    int rc = xh_pm_set_option(pm, "mirror", 0.0);
    int32_t N = 0;
    if (rc == XH_OK) rc = xh_pm_info(pm, &N, nullptr, nullptr);
    XhBuf bRef, bPsi, bFlip;
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bRef, sizeof(int32_t) * n);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bPsi, sizeof(int32_t) * n);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bFlip, n);
    if (rc == XH_OK) rc = xh_pm_match(pm, d_others, n, nullptr, nullptr, 0, (int32_t *)bRef.p, (int32_t *)bPsi.p, (uint8_t *)bFlip.p);
    std::vector<int32_t> psi(n);
    if (rc == XH_OK && hipMemcpyAsync(psi.data(), bPsi.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    xh_buf_free(bRef); xh_buf_free(bPsi); xh_buf_free(bFlip);
    xh_pm_destroy(pm);
    if (rc != XH_OK) return rc;
    for (int i = 0; i < n; ++i) h_rotations[i] = (float)(((N - psi[i]) % N) * (360.0 / N));
    return XH_OK;
}

}  // extern "C"
