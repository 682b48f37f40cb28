// xh_flexalign.hip -- FlexAlign, global alignment of a movie (BASELINE config 5: first slice).
//
// Replaces ProgMovieAlignmentCorrelationGPU<T>::computeGlobalAlignment (reconstruction_adapt_cuda/
// movie_alignment_correlation_gpu.cpp:633-725) with the arithmetic of the CPU program the reference can be compared with
// (ProgMovieAlignmentCorrelation<double>, reconstruction/movie_alignment_correlation.cpp:45-157; base class
// movie_alignment_correlation_base.cpp:152-320,399-418; EquationSystemSolver::solve, eq_system_solver.cpp:35-106):
//
//   per frame   (frame - dark) * gain                       loadFrame, base.cpp:152-176
//               2-D FFT of the whole frame (xh_fft2d: 4096 x 5760 lines in four steps)
//               crop to the reduced size + low-pass filter + the normalisations of the reference's two transforms
//                                                            scaleToSizeFourier + createLPF / scaleLPF (loadData, :79-127):
//               the CPU path goes back to real space between crop and filter; a c2r transform drops what is not Hermitian on the
//               DC and Nyquist columns of the cropped half spectrum, so those two columns are symmetrised here instead
//               (k_fa_reduce) -- same spectrum, no round trip. This is what the reference's scaleFFT2DKernel fuses too.
//   per pair    S_a conj(S_b) dSize, inverse transform, centre, bestShift within --maxShift (computeShifts, :129-157;
//               correlation_matrix + bestShift, data/filters.cpp:1593-1719,1745-1754)
//   host        least squares over all pairs with one round of 3-sigma outlier rejection, reference frame = minimax of the X
//               shifts, total shifts from it (solve / findReferenceImage / computeTotalShift)
//
// The spectra of the reduced frames stay resident ([N][nY][nX] complex<float>: 0.39 GB for 40 K3 frames at the default
// 30 A); transforms in fp32 (reference: double), peak statistics and centre of mass in double.
// The patch (local) alignment of movie_alignment_correlation_gpu.cpp:289-430 is not here: the reference has no CPU form of it
// to compare with (movie_alignment_correlation.cpp:63-76 throw "Not implemented").
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "xh_common.h"

namespace {
typedef float2 fa_cf;

__device__ __forceinline__ fa_cf fa_conj(fa_cf a) { return fa_cf{a.x, -a.y}; }

__global__ void __launch_bounds__(256) k_fa_load(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain,
                                                 fa_cf *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tot) return;
    float v = frame[t];
    if (dark) v -= dark[t];
    if (gain) v *= gain[t];
    out[t] = fa_cf{v, 0.f};
}

// cropped half spectrum of scaleToSizeFourier: rows 0 .. ihalf-1 from the top, the last ihalf-2 rows from the bottom
__device__ __forceinline__ fa_cf d_fa_crop(const fa_cf *__restrict__ B, int Y, int X, int nY, int ihalf, int i, int j)
{
    if (i < ihalf) return B[(size_t)i * X + j];
    const int n = nY - i;                       // 1 .. : row nY - n comes from row Y - n
    if (n >= 1 && n < ihalf - 1) return B[(size_t)(Y - n) * X + j];
    return fa_cf{0.f, 0.f};
}
// the same with what a c2r transform keeps of the DC / Nyquist column
__device__ __forceinline__ fa_cf d_fa_crop_sym(const fa_cf *__restrict__ B, int Y, int X, int nY, int nX, int ihalf, int i, int j)
{
    const fa_cf c = d_fa_crop(B, Y, X, nY, ihalf, i, j);
    if (j == 0 || ((nX & 1) == 0 && j == nX / 2)) {
        const fa_cf m = d_fa_crop(B, Y, X, nY, ihalf, (nY - i) % nY, j);
        return fa_cf{0.5f * (c.x + m.x), 0.5f * (c.y - m.y)};
    }
    return c;
}

// full spectrum of the reduced, filtered frame: S(i, j) = crop_sym(i, j) * lpf(i, j) / (Y X) for j <= nX/2, Hermitian beyond
__global__ void __launch_bounds__(256) k_fa_reduce(const fa_cf *__restrict__ B, int Y, int X, fa_cf *__restrict__ S, int nY, int nX,
                                                   const float *__restrict__ lpf, float inorm)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)nY * nX) return;
    int i = (int)(t / nX), j = (int)(t - (size_t)i * nX);
    const bool mirror = j > nX / 2;
    if (mirror) { j = nX - j; i = (nY - i) % nY; }
    const int ihalf = min(nY / 2 + 1, Y / 2 + 1);
    fa_cf v = d_fa_crop_sym(B, Y, X, nY, nX, ihalf, i, j);
    const float f = lpf[(size_t)i * (nX / 2 + 1) + j] * inorm;
    v = fa_cf{v.x * f, v.y * f};
    S[t] = mirror ? fa_conj(v) : v;
}

__global__ void __launch_bounds__(256) k_fa_pair(const fa_cf *__restrict__ A, const fa_cf *__restrict__ Bs, fa_cf *__restrict__ P, size_t tot, float scale)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tot) return;
    const fa_cf a = A[t], b = Bs[t];
    P[t] = fa_cf{(a.x * b.x + a.y * b.y) * scale, (a.y * b.x - a.x * b.y) * scale};
}

// sum and sum of squares of the correlation map (real part of the inverse transform), one partial per block
__global__ void __launch_bounds__(256) k_fa_stats(const fa_cf *__restrict__ M, size_t tot, double *__restrict__ part)
{
    __shared__ double s1[256], s2[256];
    double a = 0, b = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (size_t)gridDim.x * 256) { const double v = M[t].x; a += v; b += v * v; }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}

// bestShift (data/filters.cpp:1593-1719, mask == nullptr) on the centred map: statisticsAdjust(0, 1), first maximum in raster order
// within the disc of radius maxShift, neighbourhood growth while every value stays above max / 1.414, centre of mass.
// Logical index l of CenterFFT(R, true) <-> transform index l >= 0 ? l : l + n. One block.
__global__ void __launch_bounds__(256) k_fa_bestshift(const fa_cf *__restrict__ M, int nY, int nX, int maxShift, const double *__restrict__ part, int nparts,
                                                      double *__restrict__ out)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    double sum = 0, sum2 = 0;
    for (int p = 0; p < nparts; ++p) { sum += part[2 * p]; sum2 += part[2 * p + 1]; }
    const double N = (double)nY * (double)nX;
    const double avg = sum / N;
    double sd = sqrt(fabs(sum2 / N - avg * avg));
    double a = 0, b = 0;
    if (sd != 0) { a = 1.0 / sd; b = -avg * a; }
    const int starty = -(nY / 2), startx = -(nX / 2), finy = starty + nY - 1, finx = startx + nX - 1;
    auto val = [&](int i, int j) { return a * (double)M[(size_t)(i >= 0 ? i : i + nY) * nX + (j >= 0 ? j : j + nX)].x + b; };
    // maximum within the disc, first in raster order (i outer, j inner) among equals
    const int w = 2 * maxShift + 1;
    double best = -1.79769313486231570815e+308;
    int bestIdx = 0x7fffffff;
    for (int t = threadIdx.x; t < w * w; t += 256) {
        const int i = t / w - maxShift, j = t % w - maxShift;
        if (i * i + j * j > maxShift * maxShift) continue;
        const double v = val(i, j);
        if (v > best) { best = v; bestIdx = t; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bestIdx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = sv[threadIdx.x + o];
            const int k = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && k < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double shiftX = 0, shiftY = 0;
    if (si[0] == 0x7fffffff) { out[0] = 0; out[1] = 0; out[2] = -1; return; }
    const int imax = si[0] / w - maxShift, jmax = si[0] % w - maxShift;
    const double mx = sv[0];
    int n_max = -1;
    bool neighbourhood = true;
    while (neighbourhood) {
        n_max++;
        for (int i = -n_max; i <= n_max && neighbourhood; i++) {
            const int ia = i + imax;
            if (ia < starty || ia > finy) { neighbourhood = false; break; }
            for (int j = -n_max; j <= n_max && neighbourhood; j++) {
                const int ja = j + jmax;
                if (ja < startx || ja > finx) { neighbourhood = false; break; }
                else if (mx / 1.414 > val(ia, ja)) { neighbourhood = false; break; }
            }
        }
    }
    if (imax - n_max < starty) n_max = min(imax - starty, n_max);
    if (imax + n_max > finy) n_max = min(finy - imax, n_max);
    if (jmax - n_max < starty) n_max = min(jmax - startx, n_max);       // (the reference compares with STARTINGY / FINISHINGY here too)
    if (jmax + n_max > finy) n_max = min(finx - jmax, n_max);
    double xs = 0, ys = 0, sc = 0;
    for (int i = -n_max; i <= n_max; i++)
        for (int j = -n_max; j <= n_max; j++) {
            const int ia = i + imax, ja = j + jmax;
            const double v = val(ia, ja);
            ys += ia * v; xs += ja * v; sc += v;
        }
    if (sc != 0) { shiftX = xs / sc; shiftY = ys / sc; }
    out[0] = shiftX; out[1] = shiftY; out[2] = mx;
}

// ---- host: EquationSystemSolver::solve + computeAlignment ------------------------------------------------------------------
void mean_stddev(const std::vector<double> &v, double &mean, double &sd)
{
    const size_t n = v.size();
    double s = 0, s2 = 0;
    for (double x : v) { s += x; s2 += x * x; }
    mean = n ? s / n : 0;
    sd = 0;
    if (n > 1) sd = std::sqrt(std::fabs((s2 / n - mean * mean) * ((double)n / (n - 1))));
}

// normal equations of the row-weighted system for both right-hand sides, Gauss-Jordan with partial pivoting
void weighted_least_squares(std::vector<double> &A, int rows, int cols, const std::vector<double> &w, const std::vector<double> &bx,
                            const std::vector<double> &by, std::vector<double> &sx, std::vector<double> &sy)
{
    std::vector<double> wbx(rows), wby(rows);
    for (int i = 0; i < rows; ++i) {
        const double q = std::sqrt(w[i]);
        wbx[i] = bx[i] * q; wby[i] = by[i] * q;
        for (int j = 0; j < cols; ++j) A[(size_t)i * cols + j] *= q;
    }
    const int W = cols + 2;
    std::vector<double> M((size_t)cols * W, 0.0);
    for (int i = 0; i < rows; ++i) {
        const double *r = &A[(size_t)i * cols];
        for (int p = 0; p < cols; ++p) {
            if (r[p] == 0) continue;
            for (int q = 0; q < cols; ++q) M[(size_t)p * W + q] += r[p] * r[q];
            M[(size_t)p * W + cols] += r[p] * wbx[i];
            M[(size_t)p * W + cols + 1] += r[p] * wby[i];
        }
    }
    for (int k = 0; k < cols; ++k) {
        int piv = k;
        for (int r = k + 1; r < cols; ++r) if (std::fabs(M[(size_t)r * W + k]) > std::fabs(M[(size_t)piv * W + k])) piv = r;
        if (piv != k) for (int c = 0; c < W; ++c) std::swap(M[(size_t)k * W + c], M[(size_t)piv * W + c]);
        const double d = M[(size_t)k * W + k];
        if (d == 0) continue;
        for (int r = 0; r < cols; ++r) {
            if (r == k) continue;
            const double f = M[(size_t)r * W + k] / d;
            if (f == 0) continue;
            for (int c = k; c < W; ++c) M[(size_t)r * W + c] -= f * M[(size_t)k * W + c];
        }
    }
    sx.assign(cols, 0.0); sy.assign(cols, 0.0);
    for (int k = 0; k < cols; ++k) {
        const double d = M[(size_t)k * W + k];
        if (d != 0) { sx[k] = M[(size_t)k * W + cols] / d; sy[k] = M[(size_t)k * W + cols + 1] / d; }
    }
}

void fa_solve(const std::vector<double> &bx, const std::vector<double> &by, int N, int iterations, double *shiftX, double *shiftY, int *refFrame)
{
    const int rows = N * (N - 1) / 2, cols = N - 1;
    std::vector<double> A0((size_t)rows * cols, 0.0), w(rows, 1.0), sx, sy;
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j, ++idx)
            for (int ij = i; ij < j; ++ij) A0[(size_t)idx * cols + ij] = 1;
    int it = 0;
    do {
        std::vector<double> A = A0;
        weighted_least_squares(A, rows, cols, w, bx, by, sx, sy);
        std::vector<double> ex(rows), ey(rows);
        for (int r = 0; r < rows; ++r) {
            double px = 0, py = 0;
            for (int c = 0; c < cols; ++c) { px += A[(size_t)r * cols + c] * sx[c]; py += A[(size_t)r * cols + c] * sy[c]; }
            ex[r] = bx[r] - px; ey[r] = by[r] - py;       // (rows already rejected carry a zeroed A: their residual is b itself)
        }
        double mean, sdx, sdy;
        mean_stddev(ex, mean, sdx);
        mean_stddev(ey, mean, sdy);
        for (int r = 0; r < rows; ++r)
            if (std::fabs(ex[r]) > 3 * sdx || std::fabs(ey[r]) > 3 * sdy) w[r] = 0.0;
        ++it;
    } while (it < iterations);
    auto total = [&](int iref, int j, double &tx, double &ty) {
        tx = ty = 0;
        if (iref < j) for (int jj = j - 1; jj >= iref; --jj) { tx -= sx[jj]; ty -= sy[jj]; }
        else if (iref > j) for (int jj = j; jj <= iref - 1; ++jj) { tx += sx[jj]; ty += sy[jj]; }
    };
    int best = -1;
    double worstEver = std::numeric_limits<double>::max();
    for (int iref = 0; iref < N; ++iref) {
        double worst = -1;
        for (int j = 0; j < N; ++j) {
            double tx, ty;
            total(iref, j, tx, ty);
            if (std::fabs(tx) > worst) worst = std::fabs(tx);         // X only: movie_alignment_correlation_base.cpp:258-261
        }
        if (worst < worstEver) { worstEver = worst; best = iref; }
    }
    *refFrame = best;
    for (int i = 0; i < N; ++i) total(best, i, shiftX[i], shiftY[i]);
}
}  // namespace

struct xh_fa {
    xh_ctx *ctx;
    int Y, X, nY, nX;
    float Ts, maxRes;
    double sizeFactor;
    xh_fft2d *big, *small;
    XhBuf work, spectra, lpf, pair, part, res;
    int capFrames;
};

extern "C" {

int xh_fa_destroy(xh_fa *h)
{
    if (!h) return XH_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->big) xh_fft2d_destroy(h->big);
    if (h->small) xh_fft2d_destroy(h->small);
    xh_buf_free(h->work); xh_buf_free(h->spectra); xh_buf_free(h->lpf); xh_buf_free(h->pair); xh_buf_free(h->part); xh_buf_free(h->res);
    delete h;
    return XH_OK;
}

int xh_fa_create(xh_ctx *ctx, int32_t Y, int32_t X, float sampling_rate, float max_res_for_correlation, xh_fa **out)
{
    XH_CHECK(ctx && out && Y >= 8 && X >= 8 && sampling_rate > 0 && max_res_for_correlation > 0, XH_ERR_ARG, "xh_fa_create: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    // getC, getTsPrime, getScaleFactor (movie_alignment_correlation_base.cpp:286-314), float like the reference
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));
    const float tsPrime = max_res_for_correlation / (8.f * c);
    const float scale = sampling_rate / tsPrime;
    XH_CHECK(scale < 1, XH_ERR_ARG, "xh_fa_create: the correlation scale factor is bigger than one; for this sampling rate use a maximal resolution of %g or higher "
             "(checkSettings, movie_alignment_correlation_base.cpp:74-79)", (double)(sampling_rate * 8 * c));
    xh_fa *h = new xh_fa;
    h->ctx = ctx; h->Y = Y; h->X = X; h->Ts = sampling_rate; h->maxRes = max_res_for_correlation;
    h->sizeFactor = scale;
    h->nX = (int)(X * h->sizeFactor); h->nY = (int)(Y * h->sizeFactor);       // loadData, movie_alignment_correlation.cpp:101-102
    h->big = h->small = nullptr;
    h->capFrames = 0;
    int rc = (h->nX >= 4 && h->nY >= 4) ? XH_OK : XH_ERR_ARG;
    if (rc != XH_OK) xh_set_error("xh_fa_create: reduced frames of %d x %d pixels", h->nY, h->nX);
    if (rc == XH_OK) rc = xh_fft2d_create(ctx, Y, X, &h->big);
    if (rc == XH_OK) rc = xh_fft2d_create(ctx, h->nY, h->nX, &h->small);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->work, sizeof(fa_cf) * (size_t)Y * X);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->pair, sizeof(fa_cf) * (size_t)h->nY * h->nX);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->part, sizeof(double) * 2 * 256);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->res, sizeof(double) * 4);
    if (rc == XH_OK) {
        // createLPF + scaleLPF (:184-227): a 1-D Gaussian profile of nX samples, looked up by |w| nX with linear interpolation
        const int nX = h->nX, nY = h->nY, nxh = nX / 2 + 1;
        const double Tsp = (double)(sampling_rate / (float)h->sizeFactor);       // getPixelResolution (float)
        std::vector<double> prof(nX);
        const double iX = 1 / (double)nX, sigma = (Tsp * c) / max_res_for_correlation;
        for (int x = 0; x < nX; ++x) { const double w = x * iX; prof[x] = std::exp(-0.5 * (w * w) / (sigma * sigma)); }
        std::vector<float> lpf((size_t)nY * nxh);
        for (int i = 0; i < nY; ++i)
            for (int j = 0; j < nxh; ++j) {
                const double wy = nY <= 1 ? 0.0 : (double)(i <= nY / 2 ? i : i - nY) / nY, wx = (double)(j <= nX / 2 ? j : j - nX) / nX;
                const double x = std::sqrt(wx * wx + wy * wy) * nX;
                const int x0 = (int)std::floor(x), x1 = x0 + 1;
                const double fx = x - x0;
                const double d0 = (x0 < 0 || x0 >= nX) ? 0.0 : prof[x0], d1 = (x1 < 0 || x1 >= nX) ? 0.0 : prof[x1];
                lpf[(size_t)i * nxh + j] = (float)((1 - fx) * d0 + fx * d1);
            }
        rc = xh_buf_alloc(ctx, h->lpf, sizeof(float) * lpf.size());
        if (rc == XH_OK && hipMemcpy(h->lpf.p, lpf.data(), sizeof(float) * lpf.size(), hipMemcpyHostToDevice) != hipSuccess) rc = XH_ERR_HIP;
    }
    if (rc != XH_OK) { xh_fa_destroy(h); return rc; }
    *out = h;
    return XH_OK;
}

int xh_fa_info(const xh_fa *h, int32_t *newY, int32_t *newX, double *size_factor)
{
    XH_CHECK(h, XH_ERR_ARG, "xh_fa_info: null handle");
    if (newY) *newY = h->nY;
    if (newX) *newX = h->nX;
    if (size_factor) *size_factor = h->sizeFactor;
    return XH_OK;
}

int xh_fa_global_alignment(xh_fa *h, const float *d_frames, int32_t N, const float *d_dark, const float *d_gain, float max_shift_px,
                           double *h_bX, double *h_bY, double *h_shiftX, double *h_shiftY, int32_t *h_ref)
{
    XH_CHECK(h && d_frames && N >= 2 && h_shiftX && h_shiftY && h_ref, XH_ERR_ARG, "xh_fa_global_alignment: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int nY = h->nY, nX = h->nX;
    const int ms = (int)(max_shift_px * h->sizeFactor);                          // computeShifts, movie_alignment_correlation.cpp:141
    XH_CHECK(ms >= 0 && ms < nY / 2 && ms < nX / 2, XH_ERR_ARG, "xh_fa_global_alignment: --maxShift %g px reaches beyond the reduced frame (%d x %d)", (double)max_shift_px, nY, nX);
    const size_t big = (size_t)h->Y * h->X, small = (size_t)nY * nX;
    XH_TRY(xh_buf_reserve(ctx, h->spectra, sizeof(fa_cf) * small * (size_t)N));
    fa_cf *S = (fa_cf *)h->spectra.p, *wk = (fa_cf *)h->work.p, *P = (fa_cf *)h->pair.p;
    const float inorm = (float)(1.0 / ((double)h->Y * (double)h->X));
    for (int n = 0; n < N; ++n) {
        hipLaunchKernelGGL(k_fa_load, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, ctx->stream, d_frames + (size_t)n * big, d_dark, d_gain, wk, big);
        XH_LAUNCH_CHECK();
        XH_TRY(xh_fft2d_exec(h->big, (float *)wk, 0));
        hipLaunchKernelGGL(k_fa_reduce, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)wk, h->Y, h->X, S + (size_t)n * small, nY, nX,
                           (const float *)h->lpf.p, inorm);
        XH_LAUNCH_CHECK();
    }
    const int rows = N * (N - 1) / 2;
    std::vector<double> bx(rows), by(rows);
    XhBuf resAll;
    XH_TRY(xh_buf_alloc(ctx, resAll, sizeof(double) * 3 * (size_t)rows));
    const double dSize = (double)small;
    const int nparts = 256;
    int idx = 0, rc = XH_OK;
    for (int i = 0; i < N - 1 && rc == XH_OK; ++i)
        for (int j = i + 1; j < N && rc == XH_OK; ++j, ++idx) {
            // FFT1 conj(FFT2) dSize through an un-normalised inverse (correlation_matrix); ours divides by dSize
            hipLaunchKernelGGL(k_fa_pair, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)(S + (size_t)i * small),
                               (const fa_cf *)(S + (size_t)j * small), P, small, (float)(dSize * dSize));
            rc = xh_fft2d_exec(h->small, (float *)P, 1);
            if (rc != XH_OK) break;
            hipLaunchKernelGGL(k_fa_stats, dim3(nparts), dim3(256), 0, ctx->stream, (const fa_cf *)P, small, (double *)h->part.p);
            hipLaunchKernelGGL(k_fa_bestshift, dim3(1), dim3(256), 0, ctx->stream, (const fa_cf *)P, nY, nX, ms, (const double *)h->part.p, nparts,
                               (double *)resAll.p + 3 * (size_t)idx);
            if (hipGetLastError() != hipSuccess) { xh_set_error("xh_fa_global_alignment: kernel launch failed"); rc = XH_ERR_HIP; }
        }
    std::vector<double> res(3 * (size_t)rows);
    if (rc == XH_OK && hipMemcpyAsync(res.data(), resAll.p, sizeof(double) * res.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    xh_buf_free(resAll);
    if (rc != XH_OK) { if (rc == XH_ERR_HIP) xh_set_error("xh_fa_global_alignment: device error"); return rc; }
    for (int r = 0; r < rows; ++r) { bx[r] = res[3 * r] / h->sizeFactor; by[r] = res[3 * r + 1] / h->sizeFactor; }       // scale to the movie's pixels
    if (h_bX) std::memcpy(h_bX, bx.data(), sizeof(double) * rows);
    if (h_bY) std::memcpy(h_bY, by.data(), sizeof(double) * rows);
    int ref = 0;
    fa_solve(bx, by, N, 2, h_shiftX, h_shiftY, &ref);          // solverIterations = 2 (movie_alignment_correlation_base.h:332)
    *h_ref = ref;
    return XH_OK;
}

}  // extern "C"
