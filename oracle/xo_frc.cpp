// oracle/xo_frc.cpp -- CPU restatement of the Fourier ring / shell correlation behind
// xmipp_resolution_fsc. TEST INFRASTRUCTURE ONLY: never linked, imported or executed by the product.
//
// The program (reconstruction/resolution_fsc.cpp:179-203) reads two volumes and calls
//     frc_dpr(refI(), img(), sam, freq, frc, frc_noise, dpr, error_l2, do_dpr, do_rfactor, min_samp,
//             sam/max_sam, &rFactor)
// which lives in xmippCore (xmipp_fftw.cpp, I2PC/xmippCore @ v4, absent from /root/reference). Restated from
// its published source: both inputs are transformed with FourierTransformer (r2c, forward normalised by
// 1/size; the normalisation cancels in every output except error_l2), every FFTW half-complex coefficient
// with digital frequency R <= 0.5 goes to the shell round(R * xsize) and
//     frc      = sum Re(conj(z1) z2) / sqrt(sum |z1|^2 * sum |z2|^2)
//     frc_noise= 2 / sqrt(count)            error_l2 = sum |z1 - z2| / count
//     dpr      = sqrt( sum (|z1|+|z2|) dphi^2 / sum (|z1|+|z2|) ),  dphi = wrap(deg(arg z1 - arg z2), -180, 180)
//     freq[i]  = i / (xsize * sampling_rate)
//     rFactor  = sum | |z1| - |z2| | / sum |z1|   over minFreq <= R <= maxFreq, BEFORE the R <= 0.5 cut
// PIN: the only known answer in the reference tree is the R-factor of two 3x3x3 volumes,
// 0.134661 +- 1e-5 (applications/tests/function_tests/test_resolution_frc.cpp:17-131); it fixes the
// half-complex domain and the position of the frequency window relative to the Nyquist cut (the other three
// combinations give 0.1093, 0.1355, 0.1852). The shell sums themselves have no known answer in the tree:
// "parity unpinned" for frc/dpr/error_l2 beyond their definition above and the numpy cross-check of
// tests/test_oracle_pins.py.
#include "xo.h"
#include "xo_internal.h"
#include <cmath>
#include <vector>

typedef std::complex<double> cd;

namespace {
// FFT_IDX2DIGFREQ_FAST (xmippCore xmipp_fft.h): idx -> digital frequency in [-0.5, 0.5]
inline double digfreq(int idx, int size) { return (double)(idx <= size / 2 ? idx : idx - size) / (double)size; }

// r2c 3-D transform normalised by 1/size: out[z][y][x/2+1]
void fft3d_r2c(const double *in, int Z, int Y, int X, std::vector<cd> &out)
{
    const int xh = X / 2 + 1;
    out.assign((size_t)Z * Y * xh, cd(0, 0));
    std::vector<cd> a, A;
    a.resize(X); A.resize(X);
    for (int k = 0; k < Z; ++k)
        for (int i = 0; i < Y; ++i) {
            for (int j = 0; j < X; ++j) a[j] = in[((size_t)k * Y + i) * X + j];
            xo::c2c(a.data(), X, -1, A.data());
            for (int j = 0; j < xh; ++j) out[((size_t)k * Y + i) * xh + j] = A[j];
        }
    a.resize(Y); A.resize(Y);
    for (int k = 0; k < Z; ++k)
        for (int j = 0; j < xh; ++j) {
            for (int i = 0; i < Y; ++i) a[i] = out[((size_t)k * Y + i) * xh + j];
            xo::c2c(a.data(), Y, -1, A.data());
            for (int i = 0; i < Y; ++i) out[((size_t)k * Y + i) * xh + j] = A[i];
        }
    a.resize(Z); A.resize(Z);
    for (int i = 0; i < Y; ++i)
        for (int j = 0; j < xh; ++j) {
            for (int k = 0; k < Z; ++k) a[k] = out[((size_t)k * Y + i) * xh + j];
            xo::c2c(a.data(), Z, -1, A.data());
            for (int k = 0; k < Z; ++k) out[((size_t)k * Y + i) * xh + j] = A[k];
        }
    const double inv = 1.0 / ((double)Z * Y * X);
    for (cd &v : out) v *= inv;
}
}  // namespace

extern "C" int xo_frc_dpr(const double *m1, const double *m2, int Z, int Y, int X, double sampling_rate, int dodpr,
                          int dorfactor, double minFreq, double maxFreq, double *freq, double *frc, double *frc_noise,
                          double *dpr, double *error_l2, double *rFactor)
{
    const int L = X / 2 + 1, xh = X / 2 + 1;
    std::vector<cd> F1, F2;
    fft3d_r2c(m1, Z, Y, X, F1);
    fft3d_r2c(m2, Z, Y, X, F2);
    std::vector<double> num(L, 0.), den1(L, 0.), den2(L, 0.), dprn(L, 0.), dprd(L, 0.), l2(L, 0.);
    std::vector<long> count(L, 0);
    double Rn = 0, Rd = 0;
    for (int k = 0; k < Z; ++k) {
        const double fz = digfreq(k, Z);
        for (int i = 0; i < Y; ++i) {
            const double fy = digfreq(i, Y);
            for (int j = 0; j < xh; ++j) {
                const double fx = digfreq(j, X);
                const double R2 = fz * fz + fy * fy + fx * fx;
                const cd z1 = F1[((size_t)k * Y + i) * xh + j], z2 = F2[((size_t)k * Y + i) * xh + j];
                const double absz1 = std::abs(z1), absz2 = std::abs(z2);
                if (dorfactor) {
                    const double R = std::sqrt(R2);
                    if (R >= minFreq && R <= maxFreq) { Rn += std::fabs(absz1 - absz2); Rd += absz1; }
                }
                if (R2 > 0.25) continue;
                const int idx = (int)std::round(std::sqrt(R2) * X);
                if (idx >= L) continue;
                num[idx] += std::real(std::conj(z1) * z2);
                den1[idx] += absz1 * absz1;
                den2[idx] += absz2 * absz2;
                l2[idx] += std::abs(z1 - z2);
                if (dodpr) {
                    const double dphi = xo::realWRAP((std::atan2(z1.imag(), z1.real()) - std::atan2(z2.imag(), z2.real())) * 180.0 / M_PI, -180, 180);
                    dprn[idx] += (absz1 + absz2) * dphi * dphi;
                    dprd[idx] += absz1 + absz2;
                }
                ++count[idx];
            }
        }
    }
    for (int i = 0; i < L; ++i) {
        freq[i] = (double)i / (X * sampling_rate);
        frc[i] = num[i] / std::sqrt(den1[i] * den2[i]);
        frc_noise[i] = 2 / std::sqrt((double)count[i]);
        error_l2[i] = l2[i] / (double)count[i];
        if (dodpr && dpr) dpr[i] = std::sqrt(dprn[i] / dprd[i]);
    }
    if (dorfactor && rFactor) *rFactor = Rn / Rd;
    return L;
}
