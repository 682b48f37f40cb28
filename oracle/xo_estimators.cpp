// oracle/xo_estimators.cpp -- the batched estimator chain behind IterativeAlignmentEstimator.  TEST INFRASTRUCTURE ONLY.
//
// Follows (paths relative to /root/reference/src/xmipp/libraries/reconstruction):
//   shift_corr_estimator.cpp:160-283      ShiftCorrEstimator<T>::sComputeCorrelations2DOneToN / computeShift2DOneToN / computeShifts2DOneToN
//   single_extrema_finder.cpp:246-311     sFindUniversal2DAroundCenter (std::greater: the first maximum inside the circle)
//   polar_rotation_estimator.cpp:49-99    (xo_es_polar_rotation, in xo_polar.cpp next to the Polar<> restatement)
//   bspline_geo_transformer.cpp:103-137   BSplineGeoTransformer<T>::interpolate = applyGeometry(LINEAR, out, in, M, IS_INV, DONT_WRAP)
//   correlation_computer.cpp:30-56        CorrelationComputer<T>::computeOneToN<true> = correlationIndex(ref, other)
//   iterative_alignment_estimator.cpp:96-176  compute(iters, est, rotationFirst) and compute(others, iters)
// and the population of the reference's typed tests (applications/tests/function_tests/aiterative_alignment_tests.h:107-205,
// alignment_test_utils.h:36-96) so the -m gpu test can be run on the images the reference's own test draws.
//
// The reference instantiates the chain for T = float: images, the transformer's output and the pose matrices (Matrix2D<float>) are
// floats; the polar transform and correlationIndex work on doubles.  The oracle keeps images as floats where the reference stores
// floats and computes in double elsewhere (its FFT is double; the shift estimator's fftwf arithmetic is the one place where the
// oracle is more precise than the reference -- only the position of a maximum leaves that step).
#include "xo.h"
#include "xo_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

extern "C" void xo_es_polar_rotation(const double *ref, const double *others, int n, int D, int first_ring, int last_ring,
                                     double *rotations, double *corr_out);

namespace {

// computeShifts2DOneToN (shift_corr_estimator.cpp:248-283): correlation in the Fourier domain, ref * conj(other), multiplied by
// (-1)^(x + y) over the half spectrum so that the inverse transform comes out centred (:178-196), then the first maximum within
// maxShift of the centre (single_extrema_finder.cpp:265-311) as (x - X / 2, y - Y / 2)
// ties: null, or {pick, count}: the positions whose float value lies within two float ulps of the maximum are counted (in scan order) into
// ties[1] and the one with index ties[0] (clamped) is returned instead of the first maximum -- see xo_es_iterative_pass_ties
void shift_one(const std::vector<double> &Fref, const float *other, int Y, int X, int maxShift, float *sx, float *sy, double *map_out, int *ties = nullptr)
{
    const int xh = X / 2 + 1;
    std::vector<double> img((size_t)Y * X), F((size_t)Y * xh * 2), map((size_t)Y * X);
    for (size_t i = 0; i < img.size(); ++i) img[i] = other[i];
    xo_fft2d_r2c(img.data(), Y, X, F.data());
    for (int y = 0; y < Y; ++y) {
        int centerCoeff = (0 == y % 2) ? 1 : -1;
        for (int x = 0; x < xh; ++x) {
            const size_t i = (size_t)y * xh + x;
            const double a = Fref[2 * i], b = Fref[2 * i + 1], c = F[2 * i], d = -F[2 * i + 1];   // r * conj(o)
            F[2 * i] = (a * c - b * d) * centerCoeff;
            F[2 * i + 1] = (a * d + b * c) * centerCoeff;
            centerCoeff *= -1;
        }
    }
    xo_fft2d_c2r(F.data(), Y, X, map.data());
    const size_t xHalf = X / 2, yHalf = Y / 2;
    const size_t minX = xHalf - std::min(xHalf, (size_t)maxShift), minY = yHalf - std::min(yHalf, (size_t)maxShift);
    const size_t maxX = std::min((size_t)X - 1, xHalf + maxShift), maxY = std::min((size_t)Y - 1, yHalf + maxShift);
    const size_t maxDistSq = (size_t)maxShift * maxShift;
    // the reference's map is a float (T = float): the comparison is made on floats
    float extrema = std::numeric_limits<float>::lowest();
    float pos = -1;
    for (size_t y = minY; y <= maxY; ++y) {
        const int logicY = (int)y - (int)yHalf;
        const size_t ySq = (size_t)(logicY * logicY);
        for (size_t x = minX; x <= maxX; ++x) {
            const int logicX = (int)x - (int)xHalf;
            if ((ySq + (size_t)(logicX * logicX)) > maxDistSq) continue;
            const float tmp = (float)map[y * X + x];
            if (tmp > extrema) { extrema = tmp; pos = (float)(y * X + x); }
        }
    }
    if (ties) {
        const float floor2 = std::nextafterf(std::nextafterf(extrema, -INFINITY), -INFINITY);
        int cnt = 0;
        float chosen = pos;
        for (size_t y = minY; y <= maxY; ++y) {
            const int logicY = (int)y - (int)yHalf;
            const size_t ySq = (size_t)(logicY * logicY);
            for (size_t x = minX; x <= maxX; ++x) {
                const int logicX = (int)x - (int)xHalf;
                if ((ySq + (size_t)(logicX * logicX)) > maxDistSq) continue;
                if ((float)map[y * X + x] >= floor2) { if (cnt == ties[0]) chosen = (float)(y * X + x); ++cnt; }
            }
        }
        ties[1] = cnt;
        pos = chosen;
    }
    *sx = (float)(((int)pos % X) - (int)xHalf);
    *sy = (float)(((int)pos / X) - (int)yHalf);
    if (map_out) std::memcpy(map_out, map.data(), sizeof(double) * map.size());
}

void shifts(const float *ref, const float *others, int n, int Y, int X, int maxShift, float *out)
{
    const int xh = X / 2 + 1;
    std::vector<double> img((size_t)Y * X), Fref((size_t)Y * xh * 2);
    for (size_t i = 0; i < img.size(); ++i) img[i] = ref[i];
    xo_fft2d_r2c(img.data(), Y, X, Fref.data());
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; ++i) shift_one(Fref, others + (size_t)i * Y * X, Y, X, maxShift, out + 2 * i, out + 2 * i + 1, nullptr);
}

void rotations(const float *ref, const float *imgs, int n, int D, int first, int last, float *out)
{
    const size_t per = (size_t)D * D;
    std::vector<double> r(per), o(per * n), rot(n);
    for (size_t i = 0; i < per; ++i) r[i] = ref[i];
    for (size_t i = 0; i < per * n; ++i) o[i] = imgs[i];
    xo_es_polar_rotation(r.data(), o.data(), n, D, first, last, rot.data(), nullptr);
    for (int i = 0; i < n; ++i) out[i] = (float)rot[i];                       // getRotations2D() is a std::vector<float>
}

// M3x3_INV (xmippCore matrix2d.h) on Matrix2D<float>: cofactors in float, the reciprocal of the determinant in double
void m3x3_inv(const float *m, float *o)
{
    o[0] = m[8] * m[4] - m[7] * m[5];
    o[1] = -(m[8] * m[1] - m[7] * m[2]);
    o[2] = m[5] * m[1] - m[4] * m[2];
    o[3] = -(m[8] * m[3] - m[6] * m[5]);
    o[4] = m[8] * m[0] - m[6] * m[2];
    o[5] = -(m[5] * m[0] - m[3] * m[2]);
    o[6] = m[7] * m[3] - m[6] * m[4];
    o[7] = -(m[7] * m[0] - m[6] * m[1]);
    o[8] = m[4] * m[0] - m[3] * m[1];
    const double t = 1.0 / (double)(m[0] * o[0] + m[3] * o[1] + m[6] * o[2]);          // SPEED_UP_temps0: double spduptmp0
    for (int q = 0; q < 9; ++q) o[q] = (float)(o[q] * t);                                // M3x3_BY_CT into a Matrix2D<float>
}

// applyTransform (:46-58): the ORIGINAL images through the inverse of every pose, LINEAR, IS_INV, DONT_WRAP
void transform(const float *src, const std::vector<float> &poses, int n, int D, float *dest)
{
    const size_t per = (size_t)D * D;
#pragma omp parallel for schedule(dynamic)
    for (int j = 0; j < n; ++j) {
        float inv[9];
        m3x3_inv(&poses[9 * (size_t)j], inv);
        double A[9];
        for (int q = 0; q < 9; ++q) A[q] = inv[q];
        std::vector<double> in(per), out(per);
        for (size_t i = 0; i < per; ++i) in[i] = src[(size_t)j * per + i];
        xo::apply_geometry2d(1, in.data(), D, D, A, true, false, out.data());
        for (size_t i = 0; i < per; ++i) dest[(size_t)j * per + i] = (float)out[i];
    }
}

// picks / counts (n == 1 only, else null): picks[k] chooses among the tied maxima of the k-th shift step, counts[k] receives their number
void pass(const float *ref, const float *others, int n, int D, int maxShift, int first, int last, int iters, bool rotationFirst,
          std::vector<float> &poses, std::vector<float> &merit, const int *picks = nullptr, int *counts = nullptr)
{
    int shiftStep = 0;
    const size_t per = (size_t)D * D;
    std::vector<float> dest(others, others + per * n);          // copySrcToDest
    poses.assign(9 * (size_t)n, 0.f);
    for (int j = 0; j < n; ++j) poses[9 * (size_t)j] = poses[9 * (size_t)j + 4] = poses[9 * (size_t)j + 8] = 1.f;
    std::vector<float> rot(n), sh(2 * (size_t)n);
    auto stepRotation = [&] {
        rotations(ref, dest.data(), n, D, first, last, rot.data());
        for (int j = 0; j < n; ++j) {
            // rotation2DMatrix(angle, r) on a Matrix2D<float>; lhs = r * lhs
            const double a = (double)rot[j] * M_PI / 180.0;
            const float c = (float)std::cos(a), s = (float)std::sin(a);
            const float r[9] = {c, s, 0.f, -s, c, 0.f, 0.f, 0.f, 1.f};
            float *m = &poses[9 * (size_t)j], o[9];
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) {
                    float acc = 0.f;
                    for (int k = 0; k < 3; ++k) acc += r[3 * p + k] * m[3 * k + q];
                    o[3 * p + q] = acc;
                }
            std::memcpy(m, o, sizeof(o));
        }
        transform(others, poses, n, D, dest.data());
    };
    auto stepShift = [&] {
        if (picks && n == 1) {
            const int xh = D / 2 + 1;
            std::vector<double> img(per), Fref((size_t)D * xh * 2);
            for (size_t i = 0; i < per; ++i) img[i] = ref[i];
            xo_fft2d_r2c(img.data(), D, D, Fref.data());
            int t[2] = {picks[shiftStep], 0};
            shift_one(Fref, dest.data(), D, D, maxShift, &sh[0], &sh[1], nullptr, t);
            counts[shiftStep++] = t[1];
        } else
        shifts(ref, dest.data(), n, D, D, maxShift, sh.data());
        for (int j = 0; j < n; ++j) { poses[9 * (size_t)j + 2] += sh[2 * j]; poses[9 * (size_t)j + 5] += sh[2 * j + 1]; }
        transform(others, poses, n, D, dest.data());
    };
    for (int i = 0; i < iters; ++i) {
        if (rotationFirst) { stepRotation(); stepShift(); }
        else { stepShift(); stepRotation(); }
    }
    merit.resize(n);
    std::vector<double> r(per);
    for (size_t i = 0; i < per; ++i) r[i] = ref[i];
#pragma omp parallel for
    for (int j = 0; j < n; ++j) {
        std::vector<double> o(per);
        for (size_t i = 0; i < per; ++i) o[i] = dest[(size_t)j * per + i];
        merit[j] = (float)xo::correlation_index(r.data(), o.data(), per);
    }
}

}  // namespace

extern "C" {

void xo_es_shifts(const float *ref, const float *others, int n, int Y, int X, int maxShift, float *out) { shifts(ref, others, n, Y, X, maxShift, out); }

void xo_es_rotations(const float *ref, const float *others, int n, int D, int first_ring, int last_ring, float *out)
{
    rotations(ref, others, n, D, first_ring, last_ring, out);
}

// one half of compute(others, iters): the poses and merits after `iters` rounds in the given order
void xo_es_iterative_pass(const float *ref, const float *others, int n, int D, int maxShift, int first_ring, int last_ring, int iters,
                          int rotationFirst, float *poses, float *merit)
{
    std::vector<float> p, m;
    pass(ref, others, n, D, maxShift, first_ring, last_ring, iters, rotationFirst != 0, p, m);
    std::memcpy(poses, p.data(), sizeof(float) * p.size());
    std::memcpy(merit, m.data(), sizeof(float) * m.size());
}

// The same for ONE image with the arg-max ties of the shift steps resolved by the caller: picks[k] (k-th shift step) selects among the
// positions whose correlation value, as a float, lies within two ulps of the maximum; counts[k] returns how many there were.  The
// reference's map is a float computed by a float FFT: which of such positions wins there is decided by that transform's rounding,
// which neither this restatement (double transforms) nor the device can reproduce -- the tests enumerate the choices instead.
void xo_es_iterative_pass_ties(const float *ref, const float *other, int D, int maxShift, int first_ring, int last_ring, int iters,
                               int rotationFirst, const int *picks, int *counts, float *pose, float *merit)
{
    std::vector<float> p, m;
    pass(ref, other, 1, D, maxShift, first_ring, last_ring, iters, rotationFirst != 0, p, m, picks, counts);
    std::memcpy(pose, p.data(), sizeof(float) * 9);
    *merit = m[0];
}

// IterativeAlignmentEstimator<T>::compute(others, iters) (:148-169): both orders, the better merit per image
void xo_es_iterative_alignment(const float *ref, const float *others, int n, int D, int maxShift, int first_ring, int last_ring,
                               int iters, float *poses, float *merit)
{
    std::vector<float> pRS, mRS, pSR, mSR;
    pass(ref, others, n, D, maxShift, first_ring, last_ring, iters, true, pRS, mRS);
    pass(ref, others, n, D, maxShift, first_ring, last_ring, iters, false, pSR, mSR);
    for (int i = 0; i < n; ++i) {
        const bool sr = mRS[i] < mSR[i];
        merit[i] = sr ? mSR[i] : mRS[i];
        std::memcpy(poses + 9 * (size_t)i, (sr ? pSR : pRS).data() + 9 * (size_t)i, 9 * sizeof(float));
    }
}

// ---- the population of IterativeAlignmentEstimator_Test (aiterative_alignment_tests.h) --------------------------------------
// The class-static std::mt19937 mt(42) is drawn once per size (generateAndTestStatistics2D, :107-121: dist1(0, 368) then dist2(369,
// 768), halved and doubled), and COPIES of it in its state after that draw seed the shifts and the rotations of that size
// (alignment_test_utils.h:41-71: the generators take the engine by value).  draw k (0-based) of the sequence: even k = a "smaller"
// size, odd k = a "bigger" one.  Returns the size; shifts [n][2] and rotations [n] as the test generates them with libstdc++'s
// distributions (the same library the reference is built against on Linux).
int xo_es_test_population(int draw, int n, float *shiftsOut, float *rotationsOut)
{
    std::mt19937 mt(42);
    std::uniform_int_distribution<> dist1(0, 368);
    std::uniform_int_distribution<> dist2(369, 768);
    int size = 0;
    for (int k = 0; k <= draw; ++k) size = (k % 2 == 0) ? ((int)dist1(mt) / 2) * 2 : ((int)dist2(mt) / 2) * 2;
    const size_t half = (size_t)size / 2;
    const size_t maxShift = std::min((size_t)20, half - 1);          // std::min(20, getMaxShift(dims)), alignment_test_utils.h:36-39
    {   // generateShifts(dims, maxShift, mt) -- by value
        std::mt19937 g = mt;
        const size_t maxShiftSq = maxShift * maxShift;
        std::uniform_int_distribution<> dist(0, (int)maxShift);
        for (int i = 0; i < n; ++i) {
            int shiftX = dist(g);
            int shiftXSq = shiftX * shiftX;
            int maxShiftY = (int)std::floor(std::sqrt((double)(maxShiftSq - shiftXSq)));
            int shiftY = (0 == maxShiftY) ? 0 : dist(g) % maxShiftY;
            shiftsOut[2 * i] = (float)shiftX;
            shiftsOut[2 * i + 1] = (float)shiftY;
        }
    }
    {   // generateRotations(dims, maxRotation, mt) -- by value; maxRotation = 360.f - FLT_MIN = 360.f
        std::mt19937 g = mt;
        std::uniform_real_distribution<> distRot(0, 360.f - std::numeric_limits<float>::min());
        for (int i = 0; i < n; ++i) rotationsOut[i] = (float)distRot(g);
    }
    return size;
}

// addNoise (alignment_test_utils.h:66-71): std::normal_distribution<T>(0, .5) from a copy of mt_noise(23), T = float
void xo_es_test_add_noise(float *data, size_t count)
{
    std::mt19937 g(23);
    std::normal_distribution<float> dist(0.f, .5f);
    for (size_t i = 0; i < count; ++i) data[i] += dist(g);
}

// IterativeAlignmentEstimatorHelper::applyTransform (aiterative_alignment_tests.h:12-25) + sApplyTransform (:60-86): every image is the
// ONE reference moved by pose = rotation2DMatrix(rot) * (I + shift), applyGeometry(LINEAR, out, in, pose, IS_NOT_INV, DONT_WRAP)
void xo_es_test_make_others(const float *ref, int D, int n, const float *shiftsIn, const float *rotationsIn, float *others)
{
    const size_t per = (size_t)D * D;
    std::vector<double> in(per);
    for (size_t i = 0; i < per; ++i) in[i] = ref[i];
#pragma omp parallel for schedule(dynamic)
    for (int j = 0; j < n; ++j) {
        float m[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        m[2] += shiftsIn[2 * j];
        m[5] += shiftsIn[2 * j + 1];
        const double a = (double)rotationsIn[j] * M_PI / 180.0;
        const float c = (float)std::cos(a), s = (float)std::sin(a);
        const float r[9] = {c, s, 0.f, -s, c, 0.f, 0.f, 0.f, 1.f};
        double A[9];
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) {
                float acc = 0.f;
                for (int k = 0; k < 3; ++k) acc += r[3 * p + k] * m[3 * k + q];
                A[3 * p + q] = acc;
            }
        std::vector<double> out(per);
        xo::apply_geometry2d(1, in.data(), D, D, A, false, false, out.data());
        for (size_t i = 0; i < per; ++i) others[(size_t)j * per + i] = (float)out[i];
    }
}

}  // extern "C"
