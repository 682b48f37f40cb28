// oracle/xo_bspline.cpp -- cubic B-spline prefilter/interpolation, applyGeometry,
// correlation_matrix, bestShift, correlationIndex.  TEST INFRASTRUCTURE ONLY.
//
// xmippCore is absent from the reference tree; these follow its published
// algorithms (Thevenaz/Unser cubic B-spline with "mirror off bounds" boundary)
// and are pinned by the reference's unit tests:
//   rotate(BSPLINE3, 3x3, 10deg)      test_transformation_main.cpp:76-95
//   correlation_matrix 3x3            test_filters_main.cpp:71-92
//   bestShift(x,x) = 0                test_filters_main.cpp:59-69
//   correlationIndex(x,x) = 1         test_filters_main.cpp:94-103
// In-tree copies used as evidence: reconstruction_cuda/cuda_gpu_bilib.cu:16-25,
// cuda_gpu_multidim_array.cu:78-157, cuda_gpu_iirconvolve.cu:28-41,
// cuda_gpu_geo_transformer.cu:7-60.
#include "xo.h"
#include "xo_internal.h"
#include <cfloat>
#include <climits>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>
#include <vector>

namespace xo {

// 1-D in-place conversion samples -> cubic B-spline coefficients,
// half-sample-symmetric ("MirrorOffBounds") boundary, exact initialisation.
static void prefilter1d(double *c, int n, int stride)
{
    if (n == 1) return;
    const double z = std::sqrt(3.0) - 2.0;
    const double lambda = (1.0 - z) * (1.0 - 1.0 / z);  // = 6
    for (int i = 0; i < n; ++i) c[(size_t)i * stride] *= lambda;
    // causal initialisation: sum over the 2n-periodic half-sample-symmetric extension
    // c+(0) = [ s0 + sum_{k=1..n} z^k s(k-1) + sum_{k=n+1..2n-1} z^k s(2n-k) ] / (1 - z^{2n})
    double sum = c[0];
    double zk = z;
    for (int k = 1; k <= n; ++k) {
        sum += zk * c[(size_t)(k - 1) * stride];
        zk *= z;
        if (std::fabs(zk) < 1e-300) break;
    }
    if (std::fabs(zk) >= 1e-300) {
        for (int k = n + 1; k <= 2 * n - 1; ++k) {
            sum += zk * c[(size_t)(2 * n - k) * stride];
            zk *= z;
        }
        // zk == z^{2n}
        sum /= (1.0 - zk);
    }
    c[0] = sum;
    for (int i = 1; i < n; ++i) c[(size_t)i * stride] += z * c[(size_t)(i - 1) * stride];
    // anticausal initialisation (bilib MirrorOffBounds)
    c[(size_t)(n - 1) * stride] = (z / (z - 1.0)) * c[(size_t)(n - 1) * stride];
    for (int i = n - 2; i >= 0; --i)
        c[(size_t)i * stride] = z * (c[(size_t)(i + 1) * stride] - c[(size_t)i * stride]);
}

void prefilter2d(const double *in, int ydim, int xdim, double *coef)
{
    if (coef != in) std::memcpy(coef, in, sizeof(double) * (size_t)ydim * xdim);
    for (int i = 0; i < ydim; ++i) prefilter1d(coef + (size_t)i * xdim, xdim, 1);
    for (int j = 0; j < xdim; ++j) prefilter1d(coef + j, ydim, xdim);
}

// 3-D version (produceSplineCoefficients on a volume): x, then y, then z lines, in place
void prefilter3d(double *c, int zdim, int ydim, int xdim)
{
#pragma omp parallel for collapse(2)
    for (int k = 0; k < zdim; ++k)
        for (int i = 0; i < ydim; ++i) prefilter1d(c + ((size_t)k * ydim + i) * xdim, xdim, 1);
#pragma omp parallel for collapse(2)
    for (int k = 0; k < zdim; ++k)
        for (int j = 0; j < xdim; ++j) prefilter1d(c + (size_t)k * ydim * xdim + j, ydim, xdim);
#pragma omp parallel for collapse(2)
    for (int i = 0; i < ydim; ++i)
        for (int j = 0; j < xdim; ++j) prefilter1d(c + (size_t)i * xdim + j, zdim, ydim * xdim);
}

static inline double bspline03(double x)
{
    // cuda_gpu_bilib.cu:16-25 (copy of xmippCore Bspline03)
    double a = std::fabs(x);
    if (a < 1.0) return a * a * (a - 2.0) * 0.5 + 2.0 / 3.0;
    if (a < 2.0) { a -= 2.0; return a * a * a * (-1.0 / 6.0); }
    return 0.0;
}

// interpolatedElementBSpline2D(x, y, 3): x,y logical coordinates
double interp2d(const double *coef, int ydim, int xdim, int starty, int startx, double x, double y)
{
    // cuda_gpu_multidim_array.cu:78-157 after "x -= STARTINGX; y -= STARTINGY"
    x -= startx;
    y -= starty;
    const int l1 = (int)std::ceil(x - 2);
    const int m1 = (int)std::ceil(y - 2);
    int eql[4];
    double wx[4];
    for (int t = 0; t < 4; ++t) {
        int l = l1 + t;
        wx[t] = bspline03(x - (double)l);
        int e = l;
        if (l < 0) e = -l - 1;
        else if (l >= xdim) e = 2 * xdim - l - 1;
        eql[t] = e;
    }
    double columns = 0.0;
    for (int t = 0; t < 4; ++t) {
        int m = m1 + t;
        int e = m;
        if (m < 0) e = -m - 1;
        else if (m >= ydim) e = 2 * ydim - m - 1;
        const double *ref = coef + (size_t)e * xdim;
        double rows = 0.0;
        for (int s = 0; s < 4; ++s) rows += ref[eql[s]] * wx[s];
        columns += rows * bspline03(y - (double)m);
    }
    return columns;
}

double realWRAP(double x, double x0, double xF)
{
    // xmippCore xmipp_macros.h realWRAP
    if (x >= x0 && x <= xF) return x;
    if (x < x0) return x - (int)((x - x0) / (xF - x0) - 1) * (xF - x0);
    return x - (int)((x - xF) / (xF - x0) + 1) * (xF - x0);
}

static void inv3x3(const double *A, double *B)
{
    const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7],
                 i = A[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    B[0] = (e * i - f * h) * id; B[1] = (c * h - b * i) * id; B[2] = (b * f - c * e) * id;
    B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
    B[6] = (d * h - e * g) * id; B[7] = (b * g - a * h) * id; B[8] = (a * e - b * d) * id;
}

static bool is_identity3(const double *A)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (std::fabs(A[r * 3 + c] - (r == c ? 1.0 : 0.0)) > XO_EQUAL_ACCURACY) return false;
    return true;
}

// xmippCore applyGeometry (2-D branch), outside value 0.
void apply_geometry2d(int degree, const double *V1, int ydim, int xdim, const double *At, bool inv,
                      bool wrap, double *V2)
{
    if (is_identity3(At)) {
        std::memcpy(V2, V1, sizeof(double) * (size_t)ydim * xdim);
        return;
    }
    double A[9];
    if (!inv) inv3x3(At, A);
    else std::memcpy(A, At, sizeof(A));
    const int cen_y = ydim / 2, cen_x = xdim / 2;
    const int cen_yp = ydim / 2, cen_xp = xdim / 2;
    const double minxp = -cen_xp, minyp = -cen_yp;
    const double minxpp = minxp - XO_EQUAL_ACCURACY, minypp = minyp - XO_EQUAL_ACCURACY;
    const double maxxp = xdim - cen_xp - 1, maxyp = ydim - cen_yp - 1;
    const double maxxpp = maxxp + XO_EQUAL_ACCURACY, maxypp = maxyp + XO_EQUAL_ACCURACY;
    std::vector<double> B;
    if (degree > 1) {
        B.resize((size_t)ydim * xdim);
        prefilter2d(V1, ydim, xdim, B.data());
    }
    for (int i = 0; i < ydim; ++i) {
        const double x = -cen_x, y = i - cen_y;
        double xp = x * A[0] + y * A[1] + A[2];
        double yp = x * A[3] + y * A[4] + A[5];
        for (int j = 0; j < xdim; ++j) {
            bool interp = true;
            if (wrap) {
                if (xp < minxpp || xp > maxxpp) xp = realWRAP(xp, minxp - 0.5, maxxp + 0.5);
                if (yp < minypp || yp > maxypp) yp = realWRAP(yp, minyp - 0.5, maxyp + 0.5);
            } else {
                if (xp < minxpp || xp > maxxpp) interp = false;
                if (yp < minypp || yp > maxypp) interp = false;
            }
            double val = 0.0;
            if (interp) {
                if (degree == 1) {
                    double wx = xp + cen_xp;
                    int m1 = (int)wx;
                    wx = wx - m1;
                    int m2 = m1 + 1;
                    double wy = yp + cen_yp;
                    int n1 = (int)wy;
                    wy = wy - n1;
                    int n2 = n1 + 1;
                    if (wrap) {
                        if (m2 >= xdim) m2 = 0;
                        if (n2 >= ydim) n2 = 0;
                    }
                    const double wx_1 = 1 - wx, wy_1 = 1 - wy;
                    double aux2 = wy_1 * wx_1;
                    double tmp = aux2 * V1[(size_t)n1 * xdim + m1];
                    if (wx != 0 && m2 < xdim) tmp += (wy_1 - aux2) * V1[(size_t)n1 * xdim + m2];
                    if (wy != 0 && n2 < ydim) {
                        aux2 = wy * wx_1;
                        tmp += aux2 * V1[(size_t)n2 * xdim + m1];
                        if (wx != 0 && m2 < xdim) tmp += (wy - aux2) * V1[(size_t)n2 * xdim + m2];
                    }
                    val = tmp;
                } else {
                    val = interp2d(B.data(), ydim, xdim, (int)minyp, (int)minxp, xp, yp);
                }
            }
            V2[(size_t)i * xdim + j] = val;
            xp += A[0];
            yp += A[3];
        }
    }
}

void rotation2DMatrix(double ang_deg, double *A)
{
    // xmippCore rotation2DMatrix(ang, A, homogeneous=true)
    const double a = ang_deg * M_PI / 180.0;
    const double c = std::cos(a), s = std::sin(a);
    A[0] = c; A[1] = s; A[2] = 0;
    A[3] = -s; A[4] = c; A[5] = 0;
    A[6] = 0; A[7] = 0; A[8] = 1;
}

// correlation_matrix(m1, m2, R): R = IFFT(FFT(m1) . conj(FFT(m2))) * N, CenterFFT(R, true)
void correlation_matrix(const double *m1, const double *m2, int ydim, int xdim, double *R)
{
    const int xh = xdim / 2 + 1;
    std::vector<double> F1((size_t)ydim * xh * 2), F2((size_t)ydim * xh * 2);
    xo_fft2d_r2c(m1, ydim, xdim, F1.data());
    xo_fft2d_r2c(m2, ydim, xdim, F2.data());
    const double dSize = (double)xdim * ydim;
    for (size_t n = 0; n < (size_t)ydim * xh; ++n) {
        const double a = F1[2 * n], b = F1[2 * n + 1];
        const double c = F2[2 * n] * dSize, d = F2[2 * n + 1] * (-dSize);
        F2[2 * n] = a * c - b * d;
        F2[2 * n + 1] = b * c + a * d;
    }
    std::vector<double> r((size_t)ydim * xdim);
    xo_fft2d_c2r(F2.data(), ydim, xdim, r.data());
    // CenterFFT(R, true): circular shift by +dim/2 along each axis
    const int sy = ydim / 2, sx = xdim / 2;
    for (int i = 0; i < ydim; ++i)
        for (int j = 0; j < xdim; ++j)
            R[(size_t)((i + sy) % ydim) * xdim + (j + sx) % xdim] = r[(size_t)i * xdim + j];
}

// FIL:1593-1719 with mask == nullptr. Mcorr is modified (statisticsAdjust(0,1)).
double best_shift_mcorr(double *Mcorr, int ydim, int xdim, int maxShift, double &shiftX,
                        double &shiftY)
{
    const size_t N = (size_t)ydim * xdim;
    const int starty = -(ydim / 2), startx = -(xdim / 2);
    const int finy = starty + ydim - 1, finx = startx + xdim - 1;
#define MC(i, j) Mcorr[(size_t)((i) - starty) * xdim + ((j) - startx)]
    // statisticsAdjust(0,1): a = 1/stddev, b = -avg*a  (xmippCore; the sigma
    // convention only scales the map and cannot change the outputs used)
    {
        double avg = 0, sd = 0;
        for (size_t n = 0; n < N; ++n) { avg += Mcorr[n]; sd += Mcorr[n] * Mcorr[n]; }
        avg /= N;
        sd = sd / N - avg * avg;
        sd = std::sqrt(std::fabs(sd));
        double a, b;
        if (sd != 0) { a = 1.0 / sd; b = -avg * a; }
        else { a = 0; b = 0; }
        for (size_t n = 0; n < N; ++n) Mcorr[n] = a * Mcorr[n] + b;
    }
    int imax = INT_MIN, jmax = 0;
    if (maxShift == -1) {
        // maxIndex: first maximum in raster order
        double best = Mcorr[0];
        imax = starty; jmax = startx;
        for (int i = 0; i < ydim; ++i)
            for (int j = 0; j < xdim; ++j)
                if (Mcorr[(size_t)i * xdim + j] > best) {
                    best = Mcorr[(size_t)i * xdim + j];
                    imax = i + starty;
                    jmax = j + startx;
                }
    } else {
        const int maxShift2 = maxShift * maxShift;
        double bestCorr = std::numeric_limits<double>::lowest();
        for (int i = -maxShift; i <= maxShift; i++)
            for (int j = -maxShift; j <= maxShift; j++) {
                if (i * i + j * j > maxShift2) continue;
                else if (MC(i, j) > bestCorr) { imax = i; jmax = j; bestCorr = MC(imax, jmax); }
            }
    }
    const double max = MC(imax, jmax);
    bool neighbourhood = true;
    int n_max = -1;
    while (neighbourhood) {
        n_max++;
        for (int i = -n_max; i <= n_max && neighbourhood; i++) {
            int i_actual = i + imax;
            if (i_actual < starty || i_actual > finy) { neighbourhood = false; break; }
            for (int j = -n_max; j <= n_max && neighbourhood; j++) {
                int j_actual = j + jmax;
                if (j_actual < startx || j_actual > finx) { neighbourhood = false; break; }
                else if (max / 1.414 > MC(i_actual, j_actual)) { neighbourhood = false; break; }
            }
        }
    }
    double xmax = 0, ymax = 0, sumcorr = 0;
    // (the reference compares jmax against STARTINGY/FINISHINGY here, FIL:1697-1700; kept)
    if (imax - n_max < starty) n_max = std::min(imax - starty, n_max);
    if (imax + n_max > finy) n_max = std::min(finy - imax, n_max);
    if (jmax - n_max < starty) n_max = std::min(jmax - startx, n_max);
    if (jmax + n_max > finy) n_max = std::min(finx - jmax, n_max);
    for (int i = -n_max; i <= n_max; i++) {
        int i_actual = i + imax;
        for (int j = -n_max; j <= n_max; j++) {
            int j_actual = j + jmax;
            double val = MC(i_actual, j_actual);
            ymax += i_actual * val;
            xmax += j_actual * val;
            sumcorr += val;
        }
    }
    if (sumcorr != 0) { shiftX = xmax / sumcorr; shiftY = ymax / sumcorr; }
#undef MC
    return max;
}

double correlation_index(const double *x, const double *y, size_t N)
{
    // xmippCore correlationIndex without mask; computeAvgStdev's N/(N-1) factor is an
    // integer division (=1 for N>2), hence population sigma -- pinned by
    // test_filters_main.cpp:94-103 (correlationIndex(x,x) == 1 exactly).
    double mx = 0, my = 0, sx = 0, sy = 0;
    for (size_t n = 0; n < N; ++n) { mx += x[n]; sx += x[n] * x[n]; my += y[n]; sy += y[n] * y[n]; }
    mx /= N; my /= N;
    if (N > 1) {
        sx = sx / N - mx * mx; sx *= (double)(N / (N - 1)); sx = std::sqrt(std::fabs(sx));
        sy = sy / N - my * my; sy *= (double)(N / (N - 1)); sy = std::sqrt(std::fabs(sy));
    } else sx = sy = 0;
    if (std::fabs(sx) < XO_EQUAL_ACCURACY || std::fabs(sy) < XO_EQUAL_ACCURACY) return 0;
    double r = 0;
    for (size_t n = 0; n < N; ++n) r += (x[n] - mx) * (y[n] - my);
    return r / ((sx * sy) * N);
}

}  // namespace xo

extern "C" {
void xo_bspline3_prefilter2d(const double *in, int ydim, int xdim, double *coef)
{
    xo::prefilter2d(in, ydim, xdim, coef);
}
double xo_bspline3_interp2d(const double *coef, int ydim, int xdim, int starty, int startx, double x,
                            double y)
{
    return xo::interp2d(coef, ydim, xdim, starty, startx, x, y);
}
void xo_apply_geometry2d(int degree, const double *in, int ydim, int xdim, const double *A,
                         int is_inv, int wrap, double *out)
{
    xo::apply_geometry2d(degree, in, ydim, xdim, A, is_inv != 0, wrap != 0, out);
}
void xo_rotate2d(int degree, const double *in, int ydim, int xdim, double ang, int wrap, double *out)
{
    // xmippCore rotate(): rotation2DMatrix(ang) + applyGeometry(..., IS_NOT_INV, wrap)
    double A[9];
    xo::rotation2DMatrix(ang, A);
    xo::apply_geometry2d(degree, in, ydim, xdim, A, false, wrap != 0, out);
}
void xo_translate2d(int degree, const double *in, int ydim, int xdim, double sx, double sy, int wrap,
                    double *out)
{
    // xmippCore translate(): translation2DMatrix(v) + applyGeometry(..., IS_NOT_INV, wrap)
    double A[9] = {1, 0, sx, 0, 1, sy, 0, 0, 1};
    xo::apply_geometry2d(degree, in, ydim, xdim, A, false, wrap != 0, out);
}
void xo_correlation_matrix(const double *m1, const double *m2, int ydim, int xdim, double *R)
{
    xo::correlation_matrix(m1, m2, ydim, xdim, R);
}
double xo_best_shift_mcorr(double *Mcorr, int ydim, int xdim, int maxShift, double *shiftX,
                           double *shiftY)
{
    return xo::best_shift_mcorr(Mcorr, ydim, xdim, maxShift, *shiftX, *shiftY);
}
double xo_best_shift(const double *I1, const double *I2, int ydim, int xdim, int maxShift,
                     double *shiftX, double *shiftY)
{
    std::vector<double> R((size_t)ydim * xdim);
    xo::correlation_matrix(I1, I2, ydim, xdim, R.data());
    return xo::best_shift_mcorr(R.data(), ydim, xdim, maxShift, *shiftX, *shiftY);
}
double xo_correlation_index(const double *x, const double *y, size_t n)
{
    return xo::correlation_index(x, y, n);
}
}
