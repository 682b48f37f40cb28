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
//   Alignment::IterativeAlignmentEstimator<T>  reconstruction/iterative_alignment_estimator.cpp:33-176: rotation and shift estimated in turn,
//                                           the images re-interpolated from the originals by the inverse pose after every step
//                                           (BSplineGeoTransformer::interpolate, bspline_geo_transformer.cpp:103-137: applyGeometry
//                                           LINEAR, IS_INV, DONT_WRAP), both orders tried, the better correlationIndex
//                                           (CorrelationComputer, correlation_computer.cpp:30-56) kept per image
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "xh_common.h"
#include "xh_bspline.h"
#include "xh_plan.h"

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

// the shift estimator's own transforms are double precision (below): image -> complex, ref conj(other) (-1)^(x+y), real part -> float
__global__ void __launch_bounds__(256) k_es_to_complex64(const float *__restrict__ in, xh_cd *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < tot) out[t] = xh_cd{(double)in[t], 0.0};
}

__global__ void __launch_bounds__(256) k_es_correlate64(xh_cd *__restrict__ inOut, const xh_cd *__restrict__ ref, size_t per, int xdim, size_t total)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const size_t e = t % per;
    const int y = (int)(e / xdim), x = (int)(e - (size_t)y * xdim);
    const xh_cd r = ref[e], o = inOut[t];
    xh_cd v = xh_cd{r.x * o.x + r.y * o.y, r.y * o.x - r.x * o.y};
    if ((x + y) & 1) { v.x = -v.x; v.y = -v.y; }
    inOut[t] = v;
}

__global__ void __launch_bounds__(256) k_es_real64(const xh_cd *__restrict__ in, float *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < tot) out[t] = (float)in[t].x;
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

// applyGeometry(LINEAR, out, in, A, IS_INV, DONT_WRAP) of xmippCore's 2-D branch, outside value 0: out(x, y) = in at A (x, y, 1) in
// logical (Xmipp origin) coordinates, bilinear, pixels that map outside the image stay 0. A: [n][9] doubles, one matrix per image.
__global__ void __launch_bounds__(256) k_es_apply_geometry(const float *__restrict__ in, const double *__restrict__ A9, float *__restrict__ out, int ydim, int xdim)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)ydim * xdim) return;
    const int i = (int)(t / xdim), j = (int)(t - (size_t)i * xdim);
    const double *A = A9 + 9 * (size_t)blockIdx.y;
    const float *V1 = in + (size_t)blockIdx.y * ydim * xdim;
    const int cen_y = ydim / 2, cen_x = xdim / 2;
    const double eps = 1e-6;                                   // XMIPP_EQUAL_ACCURACY
    const double minxp = -cen_x, minyp = -cen_y, maxxp = xdim - cen_x - 1, maxyp = ydim - cen_y - 1;
    const double x = j - cen_x, y = i - cen_y;
    const double xp = x * A[0] + y * A[1] + A[2], yp = x * A[3] + y * A[4] + A[5];
    double val = 0.0;
    if (!(xp < minxp - eps || xp > maxxp + eps || yp < minyp - eps || yp > maxyp + eps)) {
        double wx = xp + cen_x;
        const int m1 = (int)wx;
        wx = wx - m1;
        const int m2 = m1 + 1;
        double wy = yp + cen_y;
        const int n1 = (int)wy;
        wy = wy - n1;
        const int n2 = n1 + 1;
        const double wx_1 = 1 - wx, wy_1 = 1 - wy;
        double aux2 = wy_1 * wx_1;
        double tmp = aux2 * (double)V1[(size_t)n1 * xdim + m1];
        if (wx != 0 && m2 < xdim) tmp += (wy_1 - aux2) * (double)V1[(size_t)n1 * xdim + m2];
        if (wy != 0 && n2 < ydim) {
            aux2 = wy * wx_1;
            tmp += aux2 * (double)V1[(size_t)n2 * xdim + m1];
            if (wx != 0 && m2 < xdim) tmp += (wy - aux2) * (double)V1[(size_t)n2 * xdim + m2];
        }
        val = tmp;
    }
    out[(size_t)blockIdx.y * ydim * xdim + t] = (float)val;
}

// correlationIndex(ref, other) of xmippCore without a mask (population statistics: its N / (N - 1) is an integer division); block per image
__global__ void __launch_bounds__(256) k_es_corr_index(const float *__restrict__ ref, const float *__restrict__ others, size_t N, float *__restrict__ merit)
{
    __shared__ double red[5][256];
    const float *y = others + (size_t)blockIdx.x * N;
    double mx = 0, my = 0, sx = 0, sy = 0, sxy = 0;
    for (size_t i = threadIdx.x; i < N; i += 256) { const double a = ref[i], b = y[i]; mx += a; my += b; sx += a * a; sy += b * b; sxy += a * b; }
    red[0][threadIdx.x] = mx; red[1][threadIdx.x] = my; red[2][threadIdx.x] = sx; red[3][threadIdx.x] = sy; red[4][threadIdx.x] = sxy;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double n = (double)N;
    mx = red[0][0] / n; my = red[1][0] / n;
    const double f = N > 1 ? (double)(N / (N - 1)) : 0.0;
    sx = sqrt(fabs((red[2][0] / n - mx * mx) * f)); sy = sqrt(fabs((red[3][0] / n - my * my) * f));
    double r = 0;
    if (!(fabs(sx) < 1e-6 || fabs(sy) < 1e-6)) r = (red[4][0] - n * mx * my) / ((sx * sy) * n);          // sum (x - mx)(y - my) = sum xy - n mx my
    merit[blockIdx.x] = (float)r;
}
// ---- PolarRotationEstimator (polar_rotation_estimator.cpp:49-99) as the reference computes it: rings sampled with BsplineOrder 1
// (polar.h:689-693: interpolatedElement2DOutsideZero, bilinear, zero outside the image), not normalised, every ring's
// DFT / nsam (polar.cpp:34-54), the ring-weighted products summed per frequency and brought back over 2 N - 1 angles
// (:58; polar.cpp:99-148), the first maximum (polar.cpp:212-233).  Double precision throughout, as the reference.
struct EsRing { int nsam, soff, coff; double w; };

__global__ void __launch_bounds__(256) k_es_polar_linear(const float *__restrict__ imgs, const float *__restrict__ sx, const float *__restrict__ sy, int nsamples, int D,
                                                         double *__restrict__ rings)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nsamples) return;
    const float *img = imgs + (size_t)blockIdx.y * D * D;
    const int first = -(D / 2), last = first + D - 1;
    const double minp = first, maxp = last;
    double xp = (double)sx[t], yp = (double)sy[t];
    // coordinates outside [min - 1e-6, max + 1e-6] are wrapped (polar.h:683-686)
    if (xp < minp - 1e-6 || xp > maxp + 1e-6) xp = d_realwrap<double>(xp, minp - 0.5, maxp + 0.5);
    if (yp < minp - 1e-6 || yp > maxp + 1e-6) yp = d_realwrap<double>(yp, minp - 0.5, maxp + 0.5);
    const int x0 = (int)floor(xp), y0 = (int)floor(yp);
    const double fx = xp - x0, fy = yp - y0;
    auto at = [&](int i, int j) -> double { return (j < first || j > last || i < first || i > last) ? 0.0 : (double)img[(size_t)(i - first) * D + (j - first)]; };
    const double d00 = at(y0, x0), d01 = at(y0, x0 + 1), d10 = at(y0 + 1, x0), d11 = at(y0 + 1, x0 + 1);
    const double d0 = d00 + (d01 - d00) * fx, d1 = d10 + (d11 - d10) * fx;
    rings[(size_t)blockIdx.y * nsamples + t] = d0 + (d1 - d0) * fy;
}

// block = (ring, image): the ring's samples and the nsam twiddles in LDS, thread k sums sample s against twiddle (s k) mod nsam
__global__ void __launch_bounds__(256) k_es_ring_dft(const double *__restrict__ rings, const EsRing *__restrict__ ringTab, int nsamples, int ncoefs, int conjugate,
                                                     double2 *__restrict__ coefs)
{
    extern __shared__ double es_lds[];
    const EsRing R = ringTab[blockIdx.x];
    const int n = R.nsam;
    double *x = es_lds;
    double2 *tw = (double2 *)(es_lds + n + (n & 1));
    const double *src = rings + (size_t)blockIdx.y * nsamples + R.soff;
    for (int s = threadIdx.x; s < n; s += 256) {
        x[s] = src[s];
        double sn, cs;
        sincospi(2.0 * (double)s / (double)n, &sn, &cs);
        tw[s] = double2{cs, sn};
    }
    __syncthreads();
    const double inv = 1.0 / n;
    for (int k = threadIdx.x; k <= n / 2; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int s = 0; s < n; ++s) {
            const double2 t = tw[idx];
            const double v = x[s];
            re += v * t.x;
            im -= v * t.y;
            idx += k;
            if (idx >= n) idx -= n;
        }
        re *= inv; im *= inv;
        if (conjugate) im = -im;
        coefs[(size_t)blockIdx.y * ncoefs + R.coff + k] = double2{re, im};
    }
}

// Fsum[k] = sum over the rings that have frequency k of 2 pi r . F1[k] F2[k] (F2 arrives conjugated), polar.cpp:122-135
__global__ void __launch_bounds__(256) k_es_rot_fsum(const double2 *__restrict__ Fref, const double2 *__restrict__ F, const EsRing *__restrict__ ringTab, int nrings, int ncoefs,
                                                     int nh, double2 *__restrict__ Fsum)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nh) return;
    const double2 *f2 = F + (size_t)blockIdx.y * ncoefs;
    double re = 0.0, im = 0.0;
    for (int r = 0; r < nrings; ++r) {
        const EsRing R = ringTab[r];
        if (k > R.nsam / 2) continue;
        const double2 a = Fref[R.coff + k], c = f2[R.coff + k];
        re += R.w * (a.x * c.x - a.y * c.y);
        im += R.w * (a.y * c.x + a.x * c.y);
    }
    Fsum[(size_t)blockIdx.y * nh + k] = double2{re, im};
}

// the inverse transform of the Hermitian half Fsum over len = 2 nh - 1 (odd) angles, un-normalised: corr[j] = Re F0 + 2 sum_k Re(F_k e^{2 pi i j k / len})
__global__ void __launch_bounds__(256) k_es_rot_corr(const double2 *__restrict__ Fsum, int nh, int len, double *__restrict__ corr)
{
    extern __shared__ double es_lds[];
    double2 *F = (double2 *)es_lds;
    const double2 *src = Fsum + (size_t)blockIdx.y * nh;
    for (int k = threadIdx.x; k < nh; k += 256) F[k] = src[k];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    double acc = 0.0;
    int idx = 0;
    const double step = 2.0 / (double)len;
    for (int k = 1; k < nh; ++k) {
        idx += j;
        if (idx >= len) idx -= len;
        double sn, cs;
        sincospi(step * (double)idx, &sn, &cs);
        acc += F[k].x * cs - F[k].y * sn;
    }
    corr[(size_t)blockIdx.y * len + j] = F[0].x + 2.0 * acc;
}

// best_rotation (polar.cpp:218-229): the first element that is strictly greater than everything before it
__global__ void __launch_bounds__(256) k_es_first_max(const double *__restrict__ corr, int len, int *__restrict__ imax)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const double *c = corr + (size_t)blockIdx.x * len;
    double best = c[0];
    int bi = 0;
    for (int i = threadIdx.x; i < len; i += 256)
        if (c[i] > best) { best = c[i]; bi = i; }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = sv[threadIdx.x + o];
            const int k = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && k < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) imax[blockIdx.x] = si[0];
}
}  // namespace

// the rotation estimator's plan: ring table and sample coordinates on the device, the reference's polar Fourier transform
struct EsRotation {
    xh_ctx *ctx = nullptr;
    int D = 0, first = 0, last = 0, nrings = 0, nsamples = 0, ncoefs = 0, N = 0, len = 0, maxNsam = 0;
    XhBuf ringTab, sx, sy, Fref, rings, coefs, Fsum, corr, imax;
};

static void es_rotation_free(EsRotation &R)
{
    xh_buf_free(R.ringTab); xh_buf_free(R.sx); xh_buf_free(R.sy); xh_buf_free(R.Fref); xh_buf_free(R.rings); xh_buf_free(R.coefs);
    xh_buf_free(R.Fsum); xh_buf_free(R.corr); xh_buf_free(R.imax);
}

// polarFourierTransform<false>(..., BsplineOrder = 1) of n images [n][D][D] (float) into coefs [n][ncoefs]
static int es_rotation_transform(EsRotation &R, const float *d_imgs, int n, int conjugate, double2 *d_coefs)
{
    xh_ctx *ctx = R.ctx;
    XH_TRY(xh_buf_reserve(ctx, R.rings, sizeof(double) * (size_t)R.nsamples * n));
    hipLaunchKernelGGL(k_es_polar_linear, dim3((unsigned)((R.nsamples + 255) / 256), n), dim3(256), 0, ctx->stream, d_imgs, (const float *)R.sx.p, (const float *)R.sy.p, R.nsamples,
                       R.D, (double *)R.rings.p);
    XH_LAUNCH_CHECK();
    const size_t smem = sizeof(double) * (size_t)(R.maxNsam + (R.maxNsam & 1)) + sizeof(double2) * (size_t)R.maxNsam;
    hipLaunchKernelGGL(k_es_ring_dft, dim3(R.nrings, n), dim3(256), smem, ctx->stream, (const double *)R.rings.p, (const EsRing *)R.ringTab.p, R.nsamples, R.ncoefs, conjugate,
                       d_coefs);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

static int es_rotation_create(xh_ctx *ctx, const float *d_ref, int32_t D, int32_t first_ring, int32_t last_ring, EsRotation &R)
{
    // RotationEstimationSetting::check + PolarRotationEstimator::check (arotation_estimator.h:80-130, polar_rotation_estimator.cpp:125-141)
    XH_CHECK(D >= 6, XH_ERR_ARG, "xh_rotation_estimate: The input signal is too small.");
    XH_CHECK(first_ring >= 1 && last_ring > first_ring && last_ring < D, XH_ERR_ARG, "xh_rotation_estimate: rings %d .. %d of a %d px image (first >= 1, last > first, last < size)",
             first_ring, last_ring, D);
    R.ctx = ctx; R.D = D; R.first = first_ring; R.last = last_ring; R.nrings = last_ring - first_ring + 1;
    std::vector<EsRing> tab(R.nrings);
    int ns = 0, nc = 0;
    for (int r = 0; r < R.nrings; ++r) {
        const float radius = (float)r + first_ring;
        int nsam = 2 * (int)(0.5 * 1.0 * 6.2831853071795864769 * radius);          // getNoOfSamples, polar.h:723-726
        nsam = nsam > 1 ? nsam : 1;
        tab[r].nsam = nsam; tab[r].soff = ns; tab[r].coff = nc;
        tab[r].w = 2. * 3.14159265358979323846 * (double)radius;                     // polar.cpp:123
        ns += nsam; nc += nsam / 2 + 1;
        R.maxNsam = nsam > R.maxNsam ? nsam : R.maxNsam;
    }
    R.nsamples = ns; R.ncoefs = nc; R.N = tab[R.nrings - 1].nsam; R.len = 2 * R.N - 1;
    const size_t smem = sizeof(double) * (size_t)(R.maxNsam + (R.maxNsam & 1)) + sizeof(double2) * (size_t)R.maxNsam;
    XH_CHECK(smem <= 160 * 1024 && sizeof(double2) * (size_t)R.N <= 160 * 1024, XH_ERR_UNSUPPORTED, "xh_rotation_estimate: a ring of %d samples does not fit the 160 KB of LDS", R.maxNsam);
    XH_HIP(hipFuncSetAttribute((const void *)k_es_ring_dft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    XH_HIP(hipFuncSetAttribute((const void *)k_es_rot_corr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double2) * (size_t)R.N)));
    // the angle cache of Polar<T>::ensureAngleCache (polar.cpp:57-83): float angle, float product, stored as floats
    std::vector<float> sx(ns), sy(ns);
    for (int r = 0; r < R.nrings; ++r) {
        const float radius = (float)(r + first_ring);
        const int n = tab[r].nsam;
        const float dphi = (float)(6.2831853071795864769 / (float)n);
        for (int i = 0; i < n; ++i) {
            const float phi = i * dphi;
            sx[tab[r].soff + i] = sinf(phi) * radius;
            sy[tab[r].soff + i] = cosf(phi) * radius;
        }
    }
    XH_TRY(xh_buf_alloc(ctx, R.ringTab, sizeof(EsRing) * tab.size()));
    XH_TRY(xh_buf_alloc(ctx, R.sx, sizeof(float) * ns));
    XH_TRY(xh_buf_alloc(ctx, R.sy, sizeof(float) * ns));
    XH_TRY(xh_buf_alloc(ctx, R.Fref, sizeof(double2) * nc));
    XH_HIP(hipMemcpyAsync(R.ringTab.p, tab.data(), sizeof(EsRing) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipMemcpyAsync(R.sx.p, sx.data(), sizeof(float) * ns, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipMemcpyAsync(R.sy.p, sy.data(), sizeof(float) * ns, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));                // the host vectors go out of scope
    return es_rotation_transform(R, d_ref, 1, 0, (double2 *)R.Fref.p);                // load2DReferenceOneToN: not conjugated
}

// computeRotation2DOneToN: h_rotations [n] in degrees = imax 360 / (2 N - 1), as floats (getRotations2D is a std::vector<float>)
static int es_rotation_run(EsRotation &R, const float *d_others, int n, float *h_rotations)
{
    xh_ctx *ctx = R.ctx;
    XH_TRY(xh_buf_reserve(ctx, R.coefs, sizeof(double2) * (size_t)R.ncoefs * n));
    XH_TRY(xh_buf_reserve(ctx, R.Fsum, sizeof(double2) * (size_t)R.N * n));
    XH_TRY(xh_buf_reserve(ctx, R.corr, sizeof(double) * (size_t)R.len * n));
    XH_TRY(xh_buf_reserve(ctx, R.imax, sizeof(int) * (size_t)n));
    XH_TRY(es_rotation_transform(R, d_others, n, 1, (double2 *)R.coefs.p));
    hipLaunchKernelGGL(k_es_rot_fsum, dim3((unsigned)((R.N + 255) / 256), n), dim3(256), 0, ctx->stream, (const double2 *)R.Fref.p, (const double2 *)R.coefs.p,
                       (const EsRing *)R.ringTab.p, R.nrings, R.ncoefs, R.N, (double2 *)R.Fsum.p);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_es_rot_corr, dim3((unsigned)((R.len + 255) / 256), n), dim3(256), sizeof(double2) * (size_t)R.N, ctx->stream, (const double2 *)R.Fsum.p, R.N, R.len,
                       (double *)R.corr.p);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_es_first_max, dim3(n), dim3(256), 0, ctx->stream, (const double *)R.corr.p, R.len, (int *)R.imax.p);
    XH_LAUNCH_CHECK();
    std::vector<int> imax(n);
    XH_HIP(hipMemcpyAsync(imax.data(), R.imax.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) h_rotations[i] = (float)((double)imax[i] * (360. / R.len));
    return XH_OK;
}

struct xh_shiftcorr {
    xh_ctx *ctx;
    int x, y, maxShift;
    XhPlanBufs<double> planX, planY;      // line transforms of any length (xh_plan.h), double precision
    XhBuf ref, work, map, pos;
    bool refLoaded;
};

// 2-D complex transform of n images [n][y][x] (double) in place, un-normalised: rows then columns
static int es_fft2d64(xh_shiftcorr *h, xh_cd *d, int n, bool inverse)
{
    xh_ctx *ctx = h->ctx;
    const size_t budget = 64 * 1024;
    const int lx = xh_plan_lpb(h->planX.plan, budget, 16), ly = xh_plan_lpb(h->planY.plan, budget, 16);
    const size_t rows = (size_t)n * h->y, cols = (size_t)n * h->x;
    const size_t smx = ((size_t)lx * sizeof(xh_cd)) << h->planX.plan.logM, smy = ((size_t)ly * sizeof(xh_cd)) << h->planY.plan.logM;
    if (!inverse) {
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((rows + lx - 1) / lx)), dim3(256), smx, ctx->stream, d, h->planX.plan, rows, rows, (size_t)0, (size_t)h->x, (size_t)1, lx);
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((cols + ly - 1) / ly)), dim3(256), smy, ctx->stream, d, h->planY.plan, cols, (size_t)h->x, (size_t)h->x * h->y, (size_t)1, (size_t)h->x, ly);
    } else {
        hipLaunchKernelGGL((xh_k_fft_lines<double, true>), dim3((unsigned)((cols + ly - 1) / ly)), dim3(256), smy, ctx->stream, d, h->planY.plan, cols, (size_t)h->x, (size_t)h->x * h->y, (size_t)1, (size_t)h->x, ly);
        hipLaunchKernelGGL((xh_k_fft_lines<double, true>), dim3((unsigned)((rows + lx - 1) / lx)), dim3(256), smx, ctx->stream, d, h->planX.plan, rows, rows, (size_t)0, (size_t)h->x, (size_t)1, lx);
    }
    XH_LAUNCH_CHECK();
    return XH_OK;
}

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
    xh_plan_free(h->planX); xh_plan_free(h->planY);
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
    h->ctx = ctx; h->x = xdim; h->y = ydim; h->maxShift = max_shift; h->refLoaded = false;
    // The reference's ShiftCorrEstimator<float> transforms with fftwf; its test images (one-pixel lines) give correlation maps full of
    // exact ties, which single-precision rounding breaks at random.  The device transforms in double and compares the map as floats,
    // so that the first maximum in raster order is the one exact arithmetic has.
    int rc = xh_plan_create<double>(ctx, xdim, h->planX);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, ydim, h->planY);
    if (rc == XH_OK && ((sizeof(xh_cd) << h->planX.plan.logM) > 64 * 1024 || (sizeof(xh_cd) << h->planY.plan.logM) > 64 * 1024)) {
        xh_set_error("xh_shiftcorr_create: a line of %d x %d does not fit the LDS of the double-precision transform", xdim, ydim);
        rc = XH_ERR_UNSUPPORTED;
    }
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->ref, sizeof(xh_cd) * (size_t)xdim * ydim);
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
    hipLaunchKernelGGL(k_es_to_complex64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_ref, (xh_cd *)h->ref.p, tot);
    XH_LAUNCH_CHECK();
    XH_TRY(es_fft2d64(h, (xh_cd *)h->ref.p, 1, false));
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
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)1 << 30) / (sizeof(xh_cd) * tot)));      // at most 1 GiB of work space
    XH_TRY(xh_buf_reserve(ctx, h->work, sizeof(xh_cd) * tot * (size_t)chunk));
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        const size_t total = tot * (size_t)m;
        const unsigned grid = (unsigned)((total + 255) / 256);
        xh_cd *w = (xh_cd *)h->work.p;
        hipLaunchKernelGGL(k_es_to_complex64, dim3(grid), dim3(256), 0, ctx->stream, d_others + (size_t)i0 * tot, w, total);
        XH_TRY(es_fft2d64(h, w, m, false));
        hipLaunchKernelGGL(k_es_correlate64, dim3(grid), dim3(256), 0, ctx->stream, w, (const xh_cd *)h->ref.p, tot, h->x, total);
        XH_TRY(es_fft2d64(h, w, m, true));
        hipLaunchKernelGGL(k_es_real64, dim3(grid), dim3(256), 0, ctx->stream, (const xh_cd *)w, (float *)h->map.p + (size_t)i0 * tot, total);
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

// PolarRotationEstimator::load2DReferenceOneToN + computeRotation2DOneToN: best_rotation(reference, image) for n square images, the
// reference's arithmetic (kernels above); h_corr (optional, [n][2 N - 1] doubles) receives the correlation rows
int xh_rotation_estimate(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t D, int32_t first_ring, int32_t last_ring, float *h_rotations)
{
    XH_CHECK(ctx && d_ref && d_others && h_rotations && n >= 1, XH_ERR_ARG, "xh_rotation_estimate: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    EsRotation R;
    int rc = es_rotation_create(ctx, d_ref, D, first_ring, last_ring, R);
    if (rc == XH_OK) rc = es_rotation_run(R, d_others, n, h_rotations);
    (void)hipStreamSynchronize(ctx->stream);
    es_rotation_free(R);
    return rc;
}

// BSplineGeoTransformer<T>::interpolate (bspline_geo_transformer.cpp:103-137): image i of d_src through matrix h_matrices[i] (3 x 3, row
// major, as applyGeometry(LINEAR, out, in, M, IS_INV, DONT_WRAP) takes it: out(p) = in(M p))
int xh_apply_geometry2d(xh_ctx *ctx, const float *d_src, int32_t n, int32_t ydim, int32_t xdim, const float *h_matrices, float *d_dst)
{
    XH_CHECK(ctx && d_src && d_dst && h_matrices && n >= 1 && ydim >= 1 && xdim >= 1 && d_src != d_dst, XH_ERR_ARG, "xh_apply_geometry2d: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    std::vector<double> A(9 * (size_t)n);
    for (size_t i = 0; i < A.size(); ++i) A[i] = (double)h_matrices[i];
    XhBuf bA;
    XH_TRY(xh_buf_alloc(ctx, bA, sizeof(double) * A.size()));
    int rc = XH_OK;
    if (hipMemcpyAsync(bA.p, A.data(), sizeof(double) * A.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK) {
        const size_t per = (size_t)ydim * xdim;
        hipLaunchKernelGGL(k_es_apply_geometry, dim3((unsigned)((per + 255) / 256), n), dim3(256), 0, ctx->stream, d_src, (const double *)bA.p, d_dst, ydim, xdim);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    }
    xh_buf_free(bA);
    if (rc == XH_ERR_HIP) xh_set_error("xh_apply_geometry2d: device error");
    return rc;
}

// CorrelationComputer<T>::compute, MeritType::OneToN, normalizeResult (correlation_computer.cpp:30-56): correlationIndex(ref, other)
int xh_correlation_merit(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t ydim, int32_t xdim, float *h_merit)
{
    XH_CHECK(ctx && d_ref && d_others && h_merit && n >= 1 && ydim >= 1 && xdim >= 1, XH_ERR_ARG, "xh_correlation_merit: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    XhBuf b;
    XH_TRY(xh_buf_alloc(ctx, b, sizeof(float) * n));
    int rc = XH_OK;
    hipLaunchKernelGGL(k_es_corr_index, dim3(n), dim3(256), 0, ctx->stream, d_ref, d_others, (size_t)ydim * xdim, (float *)b.p);
    if (hipGetLastError() != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipMemcpyAsync(h_merit, b.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    xh_buf_free(b);
    if (rc == XH_ERR_HIP) xh_set_error("xh_correlation_merit: device error");
    return rc;
}

// IterativeAlignmentEstimator<T>::compute(others, iters) (iterative_alignment_estimator.cpp:96-176) for n square images of D pixels (even)
// against one reference: h_poses [n][9] (3 x 3 float, row major) and h_merit [n]. The estimators underneath are the three above.
int xh_iterative_alignment(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t D, int32_t max_shift, int32_t first_ring, int32_t last_ring,
                           int32_t iters, float *h_poses, float *h_merit)
{
    XH_CHECK(ctx && d_ref && d_others && h_poses && h_merit && n >= 1 && iters >= 1, XH_ERR_ARG, "xh_iterative_alignment: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    xh_shiftcorr *sc = nullptr;
    XH_TRY(xh_shiftcorr_create(ctx, D, D, max_shift, &sc));
    EsRotation rotEst;                   // the rotation estimator: the reference's polar transform once for all rounds
    {
        const int r0 = es_rotation_create(ctx, d_ref, D, first_ring, last_ring, rotEst);
        if (r0 != XH_OK) { es_rotation_free(rotEst); xh_shiftcorr_destroy(sc); return r0; }
    }
    const size_t per = (size_t)D * D;
    XhBuf bDest;
    int rc = xh_buf_alloc(ctx, bDest, sizeof(float) * per * (size_t)n);
    if (rc == XH_OK) rc = xh_shiftcorr_load_reference(sc, d_ref);
    float *dest = (float *)bDest.p;
    std::vector<float> rot(n), sh(2 * (size_t)n), inv(9 * (size_t)n);
    auto applyTransform = [&](const std::vector<float> &poses) {
        // M3x3_INV of every pose (float), then the transformer interpolates the ORIGINAL images with it
        for (int j = 0; j < n; ++j) {
            const float *m = &poses[9 * (size_t)j];
            float *o = &inv[9 * (size_t)j];
            o[0] = m[8] * m[4] - m[7] * m[5]; o[1] = -(m[8] * m[1] - m[7] * m[2]); o[2] = m[5] * m[1] - m[4] * m[2];
            o[3] = -(m[8] * m[3] - m[6] * m[5]); o[4] = m[8] * m[0] - m[6] * m[2]; o[5] = -(m[5] * m[0] - m[3] * m[2]);
            o[6] = m[7] * m[3] - m[6] * m[4]; o[7] = -(m[7] * m[0] - m[6] * m[1]); o[8] = m[4] * m[0] - m[3] * m[1];
            // M3x3_INV: "spduptmp0 = 1.0 / (...)" is a double (SPEED_UP_temps0), the matrix is scaled by it and stored as floats
            const double t = 1.0 / (double)(m[0] * o[0] + m[3] * o[1] + m[6] * o[2]);
            for (int q = 0; q < 9; ++q) o[q] = (float)(o[q] * t);
        }
        return xh_apply_geometry2d(ctx, d_others, n, D, D, inv.data(), dest);
    };
    auto pass = [&](bool rotationFirst, std::vector<float> &poses, std::vector<float> &merit) {
        poses.assign(9 * (size_t)n, 0.f);
        for (int j = 0; j < n; ++j) poses[9 * (size_t)j] = poses[9 * (size_t)j + 4] = poses[9 * (size_t)j + 8] = 1.f;
        int r2 = hipMemcpyAsync(dest, d_others, sizeof(float) * per * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? XH_OK : XH_ERR_HIP;   // copySrcToDest
        auto stepRotation = [&]() {
            int r3 = es_rotation_run(rotEst, dest, n, rot.data());
            if (r3 != XH_OK) return r3;
            for (int j = 0; j < n; ++j) {
                // rotation2DMatrix(angle, r); lhs = r * lhs
                const double a = (double)rot[j] * 3.14159265358979323846 / 180.0;
                const float c = (float)std::cos(a), s = (float)std::sin(a);
                float *m = &poses[9 * (size_t)j];
                const float r[9] = {c, s, 0.f, -s, c, 0.f, 0.f, 0.f, 1.f};
                float o[9];
                for (int p = 0; p < 3; ++p)
                    for (int q = 0; q < 3; ++q) o[3 * p + q] = r[3 * p] * m[q] + r[3 * p + 1] * m[3 + q] + r[3 * p + 2] * m[6 + q];
                for (int q = 0; q < 9; ++q) m[q] = o[q];
            }
            return applyTransform(poses);
        };
        auto stepShift = [&]() {
            int r3 = xh_shiftcorr_compute_shifts(sc, dest, n, sh.data());
            if (r3 != XH_OK) return r3;
            for (int j = 0; j < n; ++j) { poses[9 * (size_t)j + 2] += sh[2 * j]; poses[9 * (size_t)j + 5] += sh[2 * j + 1]; }
            return applyTransform(poses);
        };
        for (int i = 0; i < iters && r2 == XH_OK; ++i) {
            if (rotationFirst) { r2 = stepRotation(); if (r2 == XH_OK) r2 = stepShift(); }
            else { r2 = stepShift(); if (r2 == XH_OK) r2 = stepRotation(); }
        }
        merit.assign(n, 0.f);
        if (r2 == XH_OK) r2 = xh_correlation_merit(ctx, d_ref, dest, n, D, D, merit.data());
        return r2;
    };
    std::vector<float> pRS, mRS, pSR, mSR;
    // test hook (tools/diag_iterative.py): XH_ES_ORDER=RS / SR returns that half of compute() alone
    const char *only = xh_debug_env("XH_ES_ORDER");
    if (rc == XH_OK) rc = pass(!(only && !strcmp(only, "SR")), pRS, mRS);
    if (only) { pSR = pRS; mSR = mRS; }
    else if (rc == XH_OK) rc = pass(false, pSR, mSR);
    if (rc == XH_OK)
        for (int j = 0; j < n; ++j) {
            const bool sr = mRS[j] < mSR[j];
            h_merit[j] = sr ? mSR[j] : mRS[j];
            std::memcpy(h_poses + 9 * (size_t)j, (sr ? pSR : pRS).data() + 9 * (size_t)j, 9 * sizeof(float));
        }
    xh_buf_free(bDest);
    xh_shiftcorr_destroy(sc);
    (void)hipStreamSynchronize(ctx->stream);
    es_rotation_free(rotEst);
    return rc;
}

}  // extern "C"
