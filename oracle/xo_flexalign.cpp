// oracle/xo_flexalign.cpp -- CPU restatement of FlexAlign: the global alignment of the CPU program, the
// local (patch) alignment, the B-spline fit and the B-spline warp of the CUDA program. TEST INFRASTRUCTURE ONLY.
//
//   ProgMovieAlignmentCorrelation<double>::computeGlobalAlignment / loadData / computeShifts
//       reconstruction/movie_alignment_correlation.cpp:45-157
//   AProgMovieAlignmentCorrelation<T>::loadFrame, createLPF, scaleLPF, getC, getTsPrime, getScaleFactor, computeTotalShift,
//   findReferenceImage, computeAlignment      reconstruction/movie_alignment_correlation_base.cpp:152-320,399-418
//   EquationSystemSolver::solve                reconstruction/eq_system_solver.cpp:35-106
//   bestShift on the correlation matrix        data/filters.cpp:1593-1719 (xo::best_shift_mcorr)
//   ProgMovieAlignmentCorrelationGPU<T>::computeLocalAlignment, getPatchesLocation, getPatchData, getMovieBorders,
//   getCorrelationHint, localFromGlobal        reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp:124-222,288-456
//   performFFTAndScale / scaleFFT2DKernel, computeCorrelations, sFindMax2DAroundCenter, refineLocation
//       reconstruction_cuda/cuda_flexalign_scale.cpp:58-77, cuda_flexalign_correlate.cpp:95-140
//   BSplineHelper::computeBSplineCoeffs / getShift   reconstruction/bspline_helper.cpp:34-148
//   GeoTransformer::applyBSplineTransform, applyLocalShiftGeometryKernelMorePixels, interpolatedElementBSpline2D_Degree3*
//       reconstruction_cuda/cuda_gpu_geo_transformer.cpp:186-239, .cu:140-254, cuda_gpu_multidim_array.cu:236-334
//
// Out of the tree (I2PC/xmippCore @ v4, restated from its published source): scaleToSizeFourier (xmipp_fftw.cpp: forward r2c,
// copy of the rows 0 .. ihalf-1 and of the ihalf-2 last rows, columns 0 .. xsize-1, inverse c2r at the new size),
// correlation_matrix on two spectra (xmipp_filters: FFT1 conj(FFT2) dSize, inverse, CenterFFT), interpolatedElement1D
// (linear, 0 outside), weightedLeastSquares (rows scaled by sqrt(w), normal equations), Matrix1D::computeMeanAndStddev
// (sample standard deviation).
// PINNED: the pair-correlation stage of the local alignment (xo_fa_correlate) on the known answers of the reference's
// FlexAlignCorrelateTest (test_cuda_flexalign_correlate.cpp: sizes, maximal distance and 1e-4 tolerance of that test), the warp on
// the identities GeoTransformerApplyBSplineTransformTest asserts (zero coefficients = identity, zero image stays zero);
// tests/test_oracle_pins.py. PARITY UNPINNED for the rest: the reference holds no known answer for the global alignment of the
// CPU program, the patch layout, the Fourier-space reduction or the spline fit (its program tests need a CUDA device and
// compare against files that are not in the tree); those follow the source text and are checked on physics (synthetic movies
// with known drift fields).
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

namespace {
// EquationSystemSolver::solve (eq_system_solver.cpp:35-106) for any observation matrix A (rows x cols, row-major): weighted least
// squares for both right-hand sides, residuals against the row-weighted A, 3-sigma outliers get weight 0, `iterations` rounds
void solve_system(const std::vector<double> &A0, int rows, int cols, const std::vector<double> &bx, const std::vector<double> &by,
                  int iterations, std::vector<double> &sx, std::vector<double> &sy)
{
    std::vector<double> w(rows, 1.0);
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
        for (int r = 0; r < rows; ++r)
            if (std::fabs(ex[r]) > 3 * sdx || std::fabs(ey[r]) > 3 * sdy) w[r] = 0.0;
        ++it;          // (the early exit on "no outlier" only happens at verbosity > 1, L88-91)
    } while (it < iterations);
}

// computeAlignment (movie_alignment_correlation_base.cpp:399-418) from the pair shifts; refFrame < 0: findReferenceImage
void alignment_from_pairs(const std::vector<double> &bx, const std::vector<double> &by, int N, int iterations, int refIn, double *shiftX,
                          double *shiftY, int *refFrame)
{
    const int rows = N * (N - 1) / 2, cols = N - 1;
    std::vector<double> A0((size_t)rows * cols, 0.0), sx, sy;
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j, ++idx)
            for (int ij = i; ij < j; ++ij) A0[(size_t)idx * cols + ij] = 1;
    solve_system(A0, rows, cols, bx, by, iterations, sx, sy);
    // computeTotalShift (L229-244)
    auto total = [&](int iref, int j, double &tx, double &ty) {
        tx = ty = 0;
        if (iref < j) for (int jj = j - 1; jj >= iref; --jj) { tx -= sx[jj]; ty -= sy[jj]; }
        else if (iref > j) for (int jj = j; jj <= iref - 1; ++jj) { tx += sx[jj]; ty += sy[jj]; }
    };
    int best = refIn;
    if (best < 0) {
        // findReferenceImage (L246-266): minimax of the X shifts only
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
    }
    *refFrame = best;
    for (int i = 0; i < N; ++i) total(best, i, shiftX[i], shiftY[i]);
}

// One frame pair of computeCorrelations (cuda_flexalign_correlate.cpp:95-140; kernels correlate, cuda_gpu_movie_alignment_correlation_kernels.cu;
// sFindMax2DAroundCenter and refineLocation<3>, cuda_find_extrema.cu / find_extrema.h): F1 conj(F2) (-1)^(x+y) on the half spectra
// [CY][CX/2+1] (even sizes: the factor centres the correlation), c2r, first maximum in raster order within maxDist of
// (CX/2, CY/2), centre of mass of the 3 x 3 window with values relative to the maximum. Position in pixels of the map.
void correlate_pair(const cd *F1, const cd *F2, int CY, int CX, int maxDist, std::vector<double> &prod, std::vector<double> &corr, double &posX, double &posY)
{
    const int cxh = CX / 2 + 1;
    cd *P = reinterpret_cast<cd *>(prod.data());
    for (int iy = 0; iy < CY; ++iy)
        for (int ix = 0; ix < cxh; ++ix) {
            const double a = 1 - 2 * ((ix + iy) & 1);
            P[(size_t)iy * cxh + ix] = F1[(size_t)iy * cxh + ix] * std::conj(F2[(size_t)iy * cxh + ix]) * a;
        }
    xo_fft2d_c2r(prod.data(), CY, CX, corr.data());
    const int xHalf = CX / 2, yHalf = CY / 2;
    double best = -1.79769313486231570815e+308;
    int pos = -1;
    for (int y = std::max(0, yHalf - maxDist); y <= std::min(CY - 1, yHalf + maxDist); ++y)
        for (int x = std::max(0, xHalf - maxDist); x <= std::min(CX - 1, xHalf + maxDist); ++x) {
            const int ly = y - yHalf, lx = x - xHalf;
            if (ly * ly + lx * lx > maxDist * maxDist) continue;
            if (corr[(size_t)y * CX + x] > best) { best = corr[(size_t)y * CX + x]; pos = y * CX + x; }
        }
    posX = posY = 0;
    if (pos >= 0) {
        const int refY = pos / CX, refX = pos % CX;
        double refVal = corr[pos];
        refVal = (refVal == 0) ? 0 : 1.0 / refVal;
        double sw = 0, slx = 0, sly = 0;
        for (int y = std::max(0, refY - 1); y <= std::min(CY - 1, refY + 1); ++y)
            for (int x = std::max(0, refX - 1); x <= std::min(CX - 1, refX + 1); ++x) {
                const double rel = corr[(size_t)y * CX + x] * refVal;
                sw += rel; slx += x * rel; sly += y * rel;
            }
        sw = (sw == 0) ? 0 : 1.0 / sw;
        posX = slx * sw; posY = sly * sw;
    }
}

// Bspline03 (xmippCore numerical tools): the cubic B-spline
double bspline03(double x)
{
    x = std::fabs(x);
    if (x < 1) return (x * x * (x - 2) * 3 + 4) * (1.0 / 6.0);
    if (x < 2) { x -= 2; return x * x * x * (-1.0 / 6.0); }
    return 0;
}
}  // namespace

extern "C" {

// EquationSystemSolver::solve + computeAlignment (movie_alignment_correlation_base.cpp:399-418): the N-1 shifts between successive
// frames from the N(N-1)/2 pair shifts, one round of 3-sigma outlier rejection (solverIterations = 2), the reference frame as
// the minimax of the X shifts (findReferenceImage looks at X only, L258-261), the total shift of every frame from it.
void xo_fa_solve(const double *bX, const double *bY, int N, int iterations, double *shiftX, double *shiftY, int *refFrame)
{
    const int rows = N * (N - 1) / 2;
    std::vector<double> bx(bX, bX + rows), by(bY, bY + rows);
    alignment_from_pairs(bx, by, N, iterations, -1, shiftX, shiftY, refFrame);
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


// ---------------------------------------------------------------------------------------------------------------------------
// Local (patch) alignment. The reference has it on CUDA only (reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp:
// 140-222,288-430; kernels reconstruction_cuda/cuda_scaleFFT_kernels.cu:44-81, cuda_gpu_movie_alignment_correlation_kernels.cu:
// 134-183, cuda_single_extrema_finder.cu:131-187,255-311; BSplineHelper, reconstruction/bspline_helper.cpp:34-148; warp
// reconstruction_cuda/cuda_gpu_geo_transformer.cu:96-131 + cuda_gpu_multidim_array.cu:160-235): restated here from those
// sources in double. What is NOT taken from the reference: the patch and correlation sizes, which it picks by benchmarking
// cuFFT (findGoodPatchSize / findGoodCorrelationSize) -- here the requested patch size rounded down to even and the correlation
// hint (getCorrelationHint) as they are. PARITY UNPINNED (no CPU form, CUDA-only tests): checked on synthetic movies.
//
// frames: corrected (dark / gain applied) [N][Y][X]; global shifts and reference frame from the global alignment.
// Outputs: patchShifts [py][px][N][2] = round(global) + local (x, y); centers [py][px][2]; B-spline coefficients
// coeffsX / coeffsY [lT][lY][lX] (computeBSplineCoeffs: they describe the OPPOSITE transformation, used to compensate).
// dims[4] = (patch size x, y, correlation size x, y). Returns 0, or 2 when the movie is too small for the patches.
}  // extern "C"

template <typename FT>
static int local_alignment_impl(const FT *frames, int N, int Y, int X, const double *gShiftX, const double *gShiftY, int refFrame,
                          float Ts, float maxShift, float maxRes, int patchesX, int patchesY, int patchSizeX, int patchSizeY,
                          int patchesAvg, int lX, int lY, int lT, double *patchShifts, double *centers, double *coeffsX,
                          double *coeffsY, int *dims, const uint8_t *patchMask)
{
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));
    const float reqScale = Ts / (maxRes / (8.f * c));                       // getScaleFactor
    const int PX = patchSizeX & ~1, PY = patchSizeY & ~1;
    if (X < PX || Y < PY || PX < 8 || PY < 8) return 2;
    auto nearestEven = [](int v, float minScale) { int size = 2; while ((size / (float)v) < minScale) size += 2; return size; };   // getCorrelationHint
    const int CX = nearestEven(PX, reqScale), CY = nearestEven(PY, reqScale);
    if (dims) { dims[0] = PX; dims[1] = PY; dims[2] = CX; dims[3] = CY; }
    const float actualScale = (float)CX / (float)PX;
    // getMovieBorders (:204-222)
    double minX = 1e300, maxX = -1e300, minY = 1e300, maxY = -1e300;
    for (int i = 0; i < N; ++i) {
        minX = std::min(std::floor(gShiftX[i]), minX); maxX = std::max(std::ceil(gShiftX[i]), maxX);
        minY = std::min(std::floor(gShiftY[i]), minY); maxY = std::max(std::ceil(gShiftY[i]), maxY);
    }
    const double bordX = std::fabs(maxX - minX), bordY = std::fabs(maxY - minY);
    // getPatchesLocation (:139-164)
    const double windowX = X - 2 * bordX, windowY = Y - 2 * bordY;
    const double corrX = std::ceil(((patchesX * PX) - windowX) / (double)(patchesX - 1)), corrY = std::ceil(((patchesY * PY) - windowY) / (double)(patchesY - 1));
    const double stepX = PX - corrX, stepY = PY - corrY;
    // low-pass of the correlation size (setFilter, :265-271)
    std::vector<double> filter;
    create_lpf((double)(Ts / actualScale), maxRes, CX, CY, filter);
    const int pxh = PX / 2 + 1, cxh = CX / 2 + 1;
    const double norm = 1.0 / ((double)PX * PY);
    const int maxDist = (int)(maxShift * actualScale);            // context.maxShift -> size_t in sFindMax2DAroundCenter
    const int rows = N * (N - 1) / 2;
    std::vector<double> patch((size_t)PY * PX), F((size_t)PY * pxh * 2), prod((size_t)CY * cxh * 2), corr((size_t)CY * CX);
    std::vector<std::vector<double>> S(N, std::vector<double>((size_t)CY * cxh * 2));
    for (int py = 0; py < patchesY; ++py)
        for (int px = 0; px < patchesX; ++px) {
            const double tlx = bordX + px * stepX, tly = bordY + py * stepY;
            const double brx = tlx + PX - 1, bry = tly + PY - 1;
            // Rectangle::getCenter: (tl + br) / 2 ... of the point type T (float): the centre may be fractional
            const double cx = (tlx + brx) / 2, cy = (tly + bry) / 2;
            centers[((size_t)py * patchesX + px) * 2] = cx; centers[((size_t)py * patchesX + px) * 2 + 1] = cy;
            if (patchMask && !patchMask[(size_t)py * patchesX + px]) continue;          // a subset of the patches (full-size tests): their shifts only
            for (int t = 0; t < N; ++t) {
                // getPatchData (:166-202): the frames t - (avg-1)/2 .. t + avg/2 at their rounded global shift, summed
                std::fill(patch.begin(), patch.end(), 0.0);
                for (int f = std::max(0, t - ((patchesAvg - 1) / 2)); f <= std::min(N - 1, t + (patchesAvg / 2)); ++f) {
                    const int xs = (int)std::round(gShiftX[f]), ys = (int)std::round(gShiftY[f]);
                    const FT *fr = frames + (size_t)f * Y * X;
                    for (int y = 0; y < PY; ++y) {
                        const int srcY = (int)tly + y + ys, srcX = (int)tlx + xs;
                        for (int x = 0; x < PX; ++x) patch[(size_t)y * PX + x] += fr[(size_t)srcY * X + srcX + x];
                    }
                }
                // runFFTScale (cuda_flexalign_scale.cpp:58-77): un-normalised forward transform, rows 0 .. C/2 from the top and the
                // C/2 - 1 last ones from the bottom, filter, 1 / (P P)
                xo_fft2d_r2c(patch.data(), PY, PX, F.data());          // (normalised by norm already)
                const cd *B = reinterpret_cast<const cd *>(F.data());
                cd *O = reinterpret_cast<cd *>(S[t].data());
                const int yhalf = CY / 2;
                for (int iy = 0; iy < CY; ++iy) {
                    const int origY = (iy <= yhalf) ? iy : (PY - (CY - iy));
                    for (int ix = 0; ix < cxh; ++ix) O[(size_t)iy * cxh + ix] = B[(size_t)origY * pxh + ix] * filter[(size_t)iy * cxh + ix];
                }
            }
            (void)norm;
            std::vector<double> bx(rows), by(rows);
            int idx = 0;
            for (int i = 0; i < N - 1; ++i)
                for (int j = i + 1; j < N; ++j, ++idx) {
                    double posX, posY;
                    correlate_pair(reinterpret_cast<const cd *>(S[i].data()), reinterpret_cast<const cd *>(S[j].data()), CY, CX, maxDist, prod, corr, posX, posY);
                    // computeAlignment (:776-797): deduct the centre, scale to the patch's pixels
                    bx[idx] = (posX - CX / 2.0) * ((double)PX / CX);
                    by[idx] = (posY - CY / 2.0) * ((double)PY / CY);
                }
            std::vector<double> lsx(N), lsy(N);
            int ref = refFrame;
            alignment_from_pairs(bx, by, N, 2, refFrame, lsx.data(), lsy.data(), &ref);
            for (int t = 0; t < N; ++t) {
                double *o = patchShifts + (((size_t)py * patchesX + px) * N + t) * 2;
                o[0] = std::round(gShiftX[t]) + lsx[t];
                o[1] = std::round(gShiftY[t]) + lsy[t];
            }
        }
    if (patchMask) return 0;               // no fit over a subset
    // BSplineHelper::computeBSplineCoeffs (bspline_helper.cpp:34-87)
    const int nP = patchesX * patchesY, R = nP * N, Cc = lX * lY * lT;
    std::vector<double> A((size_t)R * Cc, 0.0), bX(R), bY(R), cX, cY;
    const double hX = (lX == 3) ? X : (X / (double)(lX - 3)), hY = (lY == 3) ? Y : (Y / (double)(lY - 3)), hT = (lT == 3) ? N : (N / (double)(lT - 3));
    for (int py = 0; py < patchesY; ++py)
        for (int px = 0; px < patchesX; ++px)
            for (int t = 0; t < N; ++t) {
                const int i = py * patchesX + px, row = t * nP + i;
                const int tcx = (int)centers[(size_t)i * 2], tcy = (int)centers[(size_t)i * 2 + 1];        // int tileCenterX = meta.rec.getCenter().x
                for (int ct = -1; ct < lT - 1; ++ct) {
                    const double tT = bspline03((t / hT) - ct);
                    if (tT == 0) continue;
                    for (int cy = -1; cy < lY - 1; ++cy) {
                        const double tY = bspline03((tcy / hY) - cy);
                        if (tY == 0) continue;
                        for (int cx = -1; cx < lX - 1; ++cx) {
                            const double tX = bspline03((tcx / hX) - cx);
                            A[(size_t)row * Cc + ((ct + 1) * lX * lY) + ((cy + 1) * lX) + (cx + 1)] = tT * tY * tX;
                        }
                    }
                }
                const double *sh = patchShifts + ((size_t)i * N + t) * 2;
                bX[row] = -sh[0]; bY[row] = -sh[1];
            }
    solve_system(A, R, Cc, bX, bY, 2, cX, cY);
    for (int k = 0; k < Cc; ++k) { coeffsX[k] = cX[k]; coeffsY[k] = cY[k]; }
    return 0;
}

extern "C" {

int xo_fa_local_alignment(const double *frames, int N, int Y, int X, const double *gShiftX, const double *gShiftY, int refFrame,
                          float Ts, float maxShift, float maxRes, int patchesX, int patchesY, int patchSizeX, int patchSizeY,
                          int patchesAvg, int lX, int lY, int lT, double *patchShifts, double *centers, double *coeffsX,
                          double *coeffsY, int *dims)
{
    return local_alignment_impl<double>(frames, N, Y, X, gShiftX, gShiftY, refFrame, Ts, maxShift, maxRes, patchesX, patchesY, patchSizeX, patchSizeY,
                                        patchesAvg, lX, lY, lT, patchShifts, centers, coeffsX, coeffsY, dims, nullptr);
}

// The same on float frames (a K3 movie is 3.8 GB as floats) and for the patches patchMask [py][px] marks: centres of every patch,
// shifts of the marked ones, no spline fit.
int xo_fa_local_patch_shifts_f32(const float *frames, int N, int Y, int X, const double *gShiftX, const double *gShiftY, int refFrame,
                                 float Ts, float maxShift, float maxRes, int patchesX, int patchesY, int patchSizeX, int patchSizeY,
                                 int patchesAvg, const uint8_t *patchMask, double *patchShifts, double *centers, int *dims)
{
    return local_alignment_impl<float>(frames, N, Y, X, gShiftX, gShiftY, refFrame, Ts, maxShift, maxRes, patchesX, patchesY, patchSizeX, patchSizeY,
                                       patchesAvg, 3, 3, 3, patchShifts, centers, nullptr, nullptr, dims, patchMask);
}

// BSplineHelper::getShift (bspline_helper.cpp:104-148) at pixel (x, y) of frame n
void xo_fa_bspline_shift(const double *coeffsX, const double *coeffsY, int lX, int lY, int lT, int X, int Y, int N, int x, int y, int n,
                         double *shiftX, double *shiftY)
{
    const double delta = 0.0001;
    const double hX = (lX == 3) ? X : (X / (double)(lX - 3)), hY = (lY == 3) ? Y : (Y / (double)(lY - 3)), hT = (lT == 3) ? N : (N / (double)(lT - 3));
    const double xPos = x / hX, yPos = y / hY, tPos = n / hT;
    double sx = 0, sy = 0;
    for (int it = std::max(-1, (int)tPos - 1); it <= std::min((int)tPos + 2, lT - 2); ++it) {
        const double tT = bspline03(tPos - it);
        for (int iy = std::max(-1, (int)yPos - 1); iy <= std::min((int)yPos + 2, lY - 2); ++iy) {
            const double tY = bspline03(yPos - iy);
            for (int ix = std::max(-1, (int)xPos - 1); ix <= std::min((int)xPos + 2, lX - 2); ++ix) {
                const double tmp = bspline03(xPos - ix) * tY * tT;
                if (std::fabs((float)tmp) > delta) {
                    const size_t o = (size_t)(it + 1) * (lX * lY) + (size_t)(iy + 1) * lX + (ix + 1);
                    sx += coeffsX[o] * tmp; sy += coeffsY[o] * tmp;
                }
            }
        }
    }
    *shiftX = sx; *shiftY = sy;
}

// applyBSplineTransform(3, out, frame, coeffs, n) (cuda_gpu_geo_transformer.cpp:155-184, kernel cuda_gpu_geo_transformer.cu:96-131):
// cubic B-spline coefficients of the frame, every output pixel sampled at (x - shiftX, y - shiftY), mirror boundaries
void xo_fa_apply_bspline(const double *frame, int Y, int X, const double *coeffsX, const double *coeffsY, int lX, int lY, int lT, int N, int n,
                         double *out)
{
    std::vector<double> coef((size_t)Y * X);
    xo::prefilter2d(frame, Y, X, coef.data());
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            double sx, sy;
            xo_fa_bspline_shift(coeffsX, coeffsY, lX, lY, lT, X, Y, N, x, y, n, &sx, &sy);
            out[(size_t)y * X + x] = xo::interp2d(coef.data(), Y, X, 0, 0, x - sx, y - sy);
        }
}

// CUDAFlexAlignCorrelate<T>::run (cuda_flexalign_correlate.cpp:95-140) on N real frames [N][Y][X] (even sizes): the position of
// the correlation maximum of every pair (i, j), i < j, in pixels of the map -- pos [N(N-1)/2][2] = (x, y). Known answers:
// FlexAlignCorrelateTest (applications/tests/function_tests/test_cuda_flexalign_correlate.cpp:23-70).
void xo_fa_correlate(const double *frames, int N, int Y, int X, double maxDist, double *pos)
{
    const int xh = X / 2 + 1;
    std::vector<std::vector<double>> F(N, std::vector<double>((size_t)Y * xh * 2));
    for (int n = 0; n < N; ++n) xo_fft2d_r2c(frames + (size_t)n * Y * X, Y, X, F[n].data());
    std::vector<double> prod((size_t)Y * xh * 2), corr((size_t)Y * X);
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j, ++idx)
            correlate_pair(reinterpret_cast<const cd *>(F[i].data()), reinterpret_cast<const cd *>(F[j].data()), Y, X, (int)maxDist, prod, corr, pos[2 * idx], pos[2 * idx + 1]);
}

// ProgMovieFilterDose (reconstruction/movie_filter_dose.cpp:85-170; the critical-dose curve of summovie): scalar pieces and one
// frame. PINNED on MovieFilterDoseTest (applications/tests/function_tests/test_movie_filter_dose.cpp:15-90).
double xo_dose_voltage_scaling(double accelerationVoltage)
{
    if (accelerationVoltage < 301 && accelerationVoltage > 299.) return 1.0;
    if (accelerationVoltage < 201.0 && accelerationVoltage > 199.0) return 0.8;
    return -1.0;                                   // "Bad acceleration voltage (must be 200 or 300 kV"
}
double xo_dose_filter(double dose_at_end_of_frame, double critical_dose) { return std::exp((-0.5 * dose_at_end_of_frame) / critical_dose); }
double xo_dose_critical(double spatial_frequency, double voltage_scaling_factor)
{
    return ((0.24499 * std::pow(spatial_frequency, -1.6649)) + 2.8141) * voltage_scaling_factor;
}
double xo_dose_optimal(double critical_dose) { return 2.51284 * critical_dose; }

// applyDoseFilterToImage (:115-170) between FourierTransform and inverseFourierTransform of one frame [Y][X]
void xo_dose_filter_frame(double *frame, int Y, int X, double pixel_size, double voltage_scaling_factor, double dose_start, double dose_finish)
{
    const int xh = X / 2 + 1;
    std::vector<double> F((size_t)Y * xh * 2);
    xo_fft2d_r2c(frame, Y, X, F.data());
    cd *P = reinterpret_cast<cd *>(F.data());
    const double dc = 1.79769313486231570815e+308 * 0.001;
    for (int i = 0; i < Y; ++i) {
        const double y = xo_fft_idx2digfreq(i, Y), yy = y * y;
        for (int j = 0; j < xh; ++j) {
            const double x = xo_fft_idx2digfreq(j, X);
            const double crit = (i == 0 && j == 0) ? dc : xo_dose_critical(std::sqrt(x * x + yy) / pixel_size, voltage_scaling_factor);
            const double opt = xo_dose_optimal(crit);
            if (std::fabs(dose_finish - opt) < std::fabs(dose_start - opt)) P[(size_t)i * xh + j] *= xo_dose_filter(dose_finish, crit);
            else P[(size_t)i * xh + j] = cd(0, 0);
        }
    }
    xo_fft2d_c2r(F.data(), Y, X, frame);               // (xo_fft2d_r2c divides by Y X like FourierTransform, the inverse does not)
}

}  // extern "C"

extern "C" {
// --bin of the CUDA program (CUDAFlexAlignScale::runScaleIFT, reconstruction_cuda/cuda_flexalign_scale.cpp:101-121; scaleFFT2DKernel,
// cuda_scaleFFT_kernels.cu:44-79): forward transform of the raw frame, its half spectrum cropped to the binned size (columns
// 0 .. Xb/2; rows 0 .. Yb/2 from the top, the others from the bottom) times 1 / (X Y), inverse real transform.
void xo_fa_bin_frame(const double *frame, int Y, int X, int Yb, int Xb, double *out)
{
    const int xh = X / 2 + 1, xbh = Xb / 2 + 1, yhalf = Yb / 2;
    std::vector<double> F((size_t)Y * xh * 2), H((size_t)Yb * xbh * 2);
    xo_fft2d_r2c(frame, Y, X, F.data());                    // FourierTransformer convention: already divided by X Y
    for (int idy = 0; idy < Yb; ++idy) {
        const int origY = (idy <= yhalf) ? idy : (Y - (Yb - idy));
        for (int idx = 0; idx < xbh; ++idx) {
            H[((size_t)idy * xbh + idx) * 2] = F[((size_t)origY * xh + idx) * 2];
            H[((size_t)idy * xbh + idx) * 2 + 1] = F[((size_t)origY * xh + idx) * 2 + 1];
        }
    }
    xo_fft2d_c2r(H.data(), Yb, Xb, out);
}
}
