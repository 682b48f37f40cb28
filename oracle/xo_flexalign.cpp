// oracle/xo_flexalign.cpp -- CPU restatement of FlexAlign's GLOBAL alignment. TEST INFRASTRUCTURE ONLY.
//
//   ProgMovieAlignmentCorrelation<double>::computeGlobalAlignment / loadData / computeShifts
//       reconstruction/movie_alignment_correlation.cpp:45-157
//   AProgMovieAlignmentCorrelation<T>::loadFrame, createLPF, scaleLPF, getC, getTsPrime, getScaleFactor, computeTotalShift,
//   findReferenceImage, computeAlignment      reconstruction/movie_alignment_correlation_base.cpp:152-320,399-418
//   EquationSystemSolver::solve                reconstruction/eq_system_solver.cpp:35-106
//   bestShift on the correlation matrix        data/filters.cpp:1593-1719 (xo::best_shift_mcorr)
//
// Out of the tree (I2PC/xmippCore @ v4, restated from its published source): scaleToSizeFourier (xmipp_fftw.cpp: forward r2c,
// copy of the rows 0 .. ihalf-1 and of the ihalf-2 last rows, columns 0 .. xsize-1, inverse c2r at the new size),
// correlation_matrix on two spectra (xmipp_filters: FFT1 conj(FFT2) dSize, inverse, CenterFFT), interpolatedElement1D
// (linear, 0 outside), weightedLeastSquares (rows scaled by sqrt(w), normal equations), Matrix1D::computeMeanAndStddev
// (sample standard deviation). PARITY UNPINNED: the reference holds no known answer for this path that runs without CUDA
// (test_cuda_flexalign_correlate.cpp, test_movie_filter_dose.cpp need a device); the restatement follows the source text and is
// checked on physics (synthetic movies with known drifts, tests/test_oracle_pins.py).
// The patch (local) alignment has no CPU form at all in the reference: movie_alignment_correlation.cpp:63-76 throw.
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "xo.h"
#include "xo_internal.h"

namespace {
typedef std::complex<double> cd;

// xmippCore scaleToSizeFourier(1, Ydim, Xdim, in, out)
void scale_to_size_fourier(const double *in, int Y, int X, int nY, int nX, double *out)
{
    const int xh = X / 2 + 1, nxh = nX / 2 + 1;
    std::vector<double> F((size_t)Y * xh * 2);
    xo_fft2d_r2c(in, Y, X, F.data());
    const cd *B = reinterpret_cast<const cd *>(F.data());
    std::vector<double> G((size_t)nY * nxh * 2, 0.0);
    cd *C = reinterpret_cast<cd *>(G.data());
    const int ihalf = std::min(nY / 2 + 1, Y / 2 + 1), xsize = std::min(xh, nxh);
    for (int i = 0; i < ihalf; ++i)
        for (int j = 0; j < xsize; ++j) C[(size_t)i * nxh + j] = B[(size_t)i * xh + j];
    for (int i = nY - 1, n = 1; n < ihalf - 1; --i, ++n) {
        const long ip = Y - n;
        for (int j = 0; j < xsize; ++j) C[(size_t)i * nxh + j] = B[(size_t)ip * xh + j];
    }
    xo_fft2d_c2r(G.data(), nY, nX, out);
}

// createLPF + scaleLPF (movie_alignment_correlation_base.cpp:184-227), T = double, getC() float
void create_lpf(double Ts, float maxRes, int nX, int nY, std::vector<double> &filter)
{
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));
    std::vector<double> lpf(nX);
    const double iX = 1 / (double)nX;
    const double sigma = (Ts * c) / maxRes;
    for (int x = 0; x < nX; ++x) {
        const double w = x * iX;
        lpf[x] = std::exp(-0.5 * (w * w) / (sigma * sigma));
    }
    const int nxh = nX / 2 + 1;
    filter.assign((size_t)nY * nxh, 0.0);
    for (int i = 0; i < nY; ++i)
        for (int j = 0; j < nxh; ++j) {
            const double wy = xo_fft_idx2digfreq(i, nY), wx = xo_fft_idx2digfreq(j, nX);
            const double x = std::sqrt(wx * wx + wy * wy) * nX;
            // interpolatedElement1D
            const int x0 = (int)std::floor(x), x1 = x0 + 1;
            const double fx = x - x0;
            const double d0 = (x0 < 0 || x0 >= nX) ? 0.0 : lpf[x0], d1 = (x1 < 0 || x1 >= nX) ? 0.0 : lpf[x1];
            filter[(size_t)i * nxh + j] = (1 - fx) * d0 + fx * d1;
        }
}

// sample standard deviation (Matrix1D::computeMeanAndStddev)
void mean_stddev(const std::vector<double> &v, double &mean, double &sd)
{
    const size_t n = v.size();
    double s = 0, s2 = 0;
    for (double x : v) { s += x; s2 += x * x; }
    mean = s / n;
    sd = 0;
    if (n > 1) {
        double var = s2 / n - mean * mean;
        var *= (double)n / (n - 1);
        sd = std::sqrt(std::fabs(var));
    }
}

// weightedLeastSquares for two right-hand sides: rows scaled by sqrt(w), normal equations, Gaussian elimination with pivoting
void weighted_least_squares(std::vector<double> &A, int rows, int cols, const std::vector<double> &w, const std::vector<double> &bx,
                            const std::vector<double> &by, std::vector<double> &sx, std::vector<double> &sy)
{
    std::vector<double> wbx(rows), wby(rows);
    for (int i = 0; i < rows; ++i) {
        const double wii = std::sqrt(w[i]);
        wbx[i] = bx[i] * wii; wby[i] = by[i] * wii;
        for (int j = 0; j < cols; ++j) A[(size_t)i * cols + j] *= wii;       // (the caller's A is updated, eq_system_solver.cpp:73)
    }
    std::vector<double> M((size_t)cols * (cols + 2), 0.0);
    for (int p = 0; p < cols; ++p) {
        for (int q = 0; q < cols; ++q) {
            double s = 0;
            for (int i = 0; i < rows; ++i) s += A[(size_t)i * cols + p] * A[(size_t)i * cols + q];
            M[(size_t)p * (cols + 2) + q] = s;
        }
        double tx = 0, ty = 0;
        for (int i = 0; i < rows; ++i) { tx += A[(size_t)i * cols + p] * wbx[i]; ty += A[(size_t)i * cols + p] * wby[i]; }
        M[(size_t)p * (cols + 2) + cols] = tx; M[(size_t)p * (cols + 2) + cols + 1] = ty;
    }
    const int W = cols + 2;
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
}  // namespace

extern "C" {

// EquationSystemSolver::solve + computeAlignment (movie_alignment_correlation_base.cpp:399-418): the N-1 shifts between successive
// frames from the N(N-1)/2 pair shifts, one round of 3-sigma outlier rejection (solverIterations = 2), the reference frame as
// the minimax of the X shifts (findReferenceImage looks at X only, L258-261), the total shift of every frame from it.
void xo_fa_solve(const double *bX, const double *bY, int N, int iterations, double *shiftX, double *shiftY, int *refFrame)
{
    const int rows = N * (N - 1) / 2, cols = N - 1;
    std::vector<double> A0((size_t)rows * cols, 0.0), bx(bX, bX + rows), by(bY, bY + rows), w(rows, 1.0), sx, sy;
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j) {
            for (int ij = i; ij < j; ++ij) A0[(size_t)idx * cols + ij] = 1;
            ++idx;
        }
    int it = 0;
    do {
        std::vector<double> A = A0;
        weighted_least_squares(A, rows, cols, w, bx, by, sx, sy);
        std::vector<double> ex(rows), ey(rows);
        for (int r = 0; r < rows; ++r) {
            double px = 0, py = 0;
            for (int c = 0; c < cols; ++c) { px += A[(size_t)r * cols + c] * sx[c]; py += A[(size_t)r * cols + c] * sy[c]; }
            ex[r] = bx[r] - px; ey[r] = by[r] - py;
        }
        double mean, sdx, sdy;
        mean_stddev(ex, mean, sdx);
        mean_stddev(ey, mean, sdy);
        double oldSum = 0, newSum = 0;
        for (double v : w) oldSum += v;
        for (int r = 0; r < rows; ++r)
            if (std::fabs(ex[r]) > 3 * sdx || std::fabs(ey[r]) > 3 * sdy) w[r] = 0.0;
        for (double v : w) newSum += v;
        (void)oldSum; (void)newSum;          // (the early exit on "no outlier" only happens at verbosity > 1, L88-91)
        ++it;
    } while (it < iterations);
    // computeTotalShift (L229-244)
    auto total = [&](int iref, int j, double &tx, double &ty) {
        tx = ty = 0;
        if (iref < j) for (int jj = j - 1; jj >= iref; --jj) { tx -= sx[jj]; ty -= sy[jj]; }
        else if (iref > j) for (int jj = j; jj <= iref - 1; ++jj) { tx += sx[jj]; ty += sy[jj]; }
    };
    int best = -1;
    double worstEver = 1.79769313486231570815e+308;
    for (int iref = 0; iref < N; ++iref) {
        double worst = -1;
        for (int j = 0; j < N; ++j) {
            double tx, ty;
            total(iref, j, tx, ty);
            if (std::fabs(tx) > worst) worst = std::fabs(tx);
        }
        if (worst < worstEver) { worstEver = worst; best = iref; }
    }
    *refFrame = best;
    for (int i = 0; i < N; ++i) total(best, i, shiftX[i], shiftY[i]);
}

// Returns 0, or 1 when the correlation scale factor is >= 1 (checkSettings, L74-79). frames: [N][Y][X]; dark / igain: [Y][X] or
// null; maxShift in pixels of the movie (the program divides --maxShift by the sampling rate, L43).
// Outputs: pair shifts bX, bY [N(N-1)/2] in movie pixels (null to skip), frame shifts [N] from the reference frame, newDims[2]
// = (newYdim, newXdim) of the reduced frames.
int xo_fa_global_alignment(const double *frames, int N, int Y, int X, const double *dark, const double *igain, float Ts,
                           float maxShift, float maxRes, double *bX, double *bY, double *shiftX, double *shiftY, int *refFrame,
                           int *newDims)
{
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));       // getC
    const float tsPrime = maxRes / (8.f * c);                         // getTsPrime
    const float scale = Ts / tsPrime;                                 // getScaleFactor
    if (scale >= 1) return 1;
    const double sizeFactor = scale;
    const int nX = (int)(X * sizeFactor), nY = (int)(Y * sizeFactor);
    if (newDims) { newDims[0] = nY; newDims[1] = nX; }
    const int nxh = nX / 2 + 1;
    std::vector<double> filter;
    create_lpf((double)(Ts / (float)sizeFactor), maxRes, nX, nY, filter);         // getPixelResolution returns float
    // loadData
    std::vector<std::vector<double>> FF(N);
    std::vector<double> frame((size_t)Y * X), reduced((size_t)nY * nX);
    for (int n = 0; n < N; ++n) {
        const double *f = frames + (size_t)n * Y * X;
        for (size_t k = 0; k < (size_t)Y * X; ++k) {
            double v = f[k];
            if (dark) v -= dark[k];
            if (igain) v *= igain[k];
            frame[k] = v;
        }
        scale_to_size_fourier(frame.data(), Y, X, nY, nX, reduced.data());
        FF[n].resize((size_t)nY * nxh * 2);
        xo_fft2d_r2c(reduced.data(), nY, nX, FF[n].data());
        cd *F = reinterpret_cast<cd *>(FF[n].data());
        for (size_t k = 0; k < (size_t)nY * nxh; ++k) F[k] *= filter[k];
    }
    // computeShifts
    const int rows = N * (N - 1) / 2;
    std::vector<double> bx(rows), by(rows), prod((size_t)nY * nxh * 2), r((size_t)nY * nX), Mcorr((size_t)nY * nX);
    const double dSize = (double)nX * nY;
    const int ms = (int)(maxShift * sizeFactor);
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j) {
            const cd *F1 = reinterpret_cast<const cd *>(FF[i].data()), *F2 = reinterpret_cast<const cd *>(FF[j].data());
            cd *P = reinterpret_cast<cd *>(prod.data());
            for (size_t k = 0; k < (size_t)nY * nxh; ++k) P[k] = F1[k] * std::conj(F2[k]) * dSize;
            xo_fft2d_c2r(prod.data(), nY, nX, r.data());
            const int sy = nY / 2, sx = nX / 2;       // CenterFFT(R, true)
            for (int a = 0; a < nY; ++a)
                for (int b = 0; b < nX; ++b) Mcorr[(size_t)((a + sy) % nY) * nX + (b + sx) % nX] = r[(size_t)a * nX + b];
            double x = 0, y = 0;
            xo::best_shift_mcorr(Mcorr.data(), nY, nX, ms, x, y);
            bx[idx] = x / sizeFactor; by[idx] = y / sizeFactor;
            ++idx;
        }
    if (bX) std::memcpy(bX, bx.data(), sizeof(double) * rows);
    if (bY) std::memcpy(bY, by.data(), sizeof(double) * rows);
    xo_fa_solve(bx.data(), by.data(), N, 2, shiftX, shiftY, refFrame);
    return 0;
}

}  // extern "C"
