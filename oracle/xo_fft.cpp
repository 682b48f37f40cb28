// oracle/xo_fft.cpp -- self-contained double-precision FFT for the oracle.
// TEST INFRASTRUCTURE ONLY (see xo.h).
//
// Stands in for FFTW3 under xmippCore's FourierTransformer (neither is in the
// reference tree nor installed here).  Convention restated from the call sites
// and pinned by applications/tests/function_tests/test_fftw_main.cpp:35-51:
// forward transform divided by the number of samples, inverse un-normalised,
// half spectrum ydim x (xdim/2+1).
//
// Algorithm: recursive mixed-radix decimation in time for factors <= 13,
// Bluestein chirp-z for lengths with a larger prime factor (ring lengths such
// as 394 = 2*197 and 796 = 4*199 need it).
#include "xo.h"
#include <cmath>
#include <complex>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {
typedef std::complex<double> cd;

static cd unit_root(long long num, long long den, int sign)
{
    // exp(sign * 2*pi*i * num/den) with argument reduction in integers
    num %= den;
    if (num < 0) num += den;
    long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)num / (long double)den;
    return cd((double)cosl(a), (double)(sign * sinl(a)));
}

struct Plan {
    int n = 0;
    std::vector<int> factors;  // radix sequence, product = n
    std::vector<cd> tw;        // exp(-2 pi i k / n), k = 0..n-1
    bool bluestein = false;
    int m = 0;                       // Bluestein FFT length (power of two)
    std::vector<cd> chirp;           // exp(-i pi k^2 / n)
    std::vector<cd> chirpF;          // FFT_m of conj-chirp kernel
    std::shared_ptr<Plan> sub;       // plan of length m
};

static void factorize(int n, std::vector<int> &f, bool &needBluestein)
{
    needBluestein = false;
    while (n % 4 == 0) { f.push_back(4); n /= 4; }
    while (n % 2 == 0) { f.push_back(2); n /= 2; }
    for (int p = 3; p <= 13; p += 2)
        while (n % p == 0) { f.push_back(p); n /= p; }
    if (n > 1) needBluestein = true;
}

static std::shared_ptr<Plan> get_plan(int n);

// out[k*? ] : classic kissfft-like recursion. in is read with stride `istride`.
static void rec(const Plan &P, const cd *in, int istride, cd *out, int n, size_t fidx, int twstride)
{
    if (n == 1) { out[0] = in[0]; return; }
    const int p = P.factors[fidx];
    const int m = n / p;
    // sub-transforms
    for (int q = 0; q < p; ++q)
        rec(P, in + (size_t)q * istride, istride * p, out + (size_t)q * m, m, fidx + 1, twstride * p);
    // butterflies
    cd scratch[16];
    for (int k = 0; k < m; ++k) {
        for (int q = 0; q < p; ++q)
            scratch[q] = out[(size_t)q * m + k] * P.tw[(size_t)((long long)q * k * twstride) % P.n];
        if (p == 2) {
            out[k] = scratch[0] + scratch[1];
            out[k + m] = scratch[0] - scratch[1];
        } else if (p == 4) {
            cd a = scratch[0] + scratch[2], b = scratch[0] - scratch[2];
            cd c = scratch[1] + scratch[3], d = scratch[1] - scratch[3];
            cd jd(d.imag(), -d.real());  // -i*d  (forward sign)
            out[k] = a + c;
            out[k + m] = b + jd;
            out[k + 2 * m] = a - c;
            out[k + 3 * m] = b - jd;
        } else {
            // generic radix-p DFT
            const int step = P.n / p;
            for (int r = 0; r < p; ++r) {
                cd acc = scratch[0];
                for (int q = 1; q < p; ++q)
                    acc += scratch[q] * P.tw[(size_t)(((long long)q * r) % p) * step];
                out[(size_t)r * m + k] = acc;
            }
        }
    }
}

static void bluestein_exec(const Plan &P, const cd *in, cd *out)
{
    const int n = P.n, m = P.m;
    std::vector<cd> a(m), A(m);
    for (int k = 0; k < n; ++k) a[k] = in[k] * P.chirp[k];
    for (int k = n; k < m; ++k) a[k] = 0;
    rec(*P.sub, a.data(), 1, A.data(), m, 0, 1);
    for (int k = 0; k < m; ++k) A[k] = std::conj(A[k] * P.chirpF[k]);
    // inverse via conj(FFT(conj(x)))
    rec(*P.sub, A.data(), 1, a.data(), m, 0, 1);
    const double inv = 1.0 / m;
    for (int k = 0; k < n; ++k) out[k] = std::conj(a[k]) * inv * P.chirp[k];
}

static std::mutex g_mutex;
static std::map<int, std::shared_ptr<Plan>> g_plans;

static std::shared_ptr<Plan> build_plan(int n)
{
    auto P = std::make_shared<Plan>();
    P->n = n;
    bool blu = false;
    factorize(n, P->factors, blu);
    if (blu) {
        P->bluestein = true;
        P->factors.clear();
        int m = 1;
        while (m < 2 * n - 1) m <<= 1;
        P->m = m;
        P->sub = build_plan(m);
        P->chirp.resize(n);
        for (int k = 0; k < n; ++k) {
            long long k2 = ((long long)k * k) % (2LL * n);
            P->chirp[k] = unit_root(k2, 2LL * n, -1);  // exp(-i pi k^2/n)
        }
        std::vector<cd> b(m, cd(0, 0));
        b[0] = std::conj(P->chirp[0]);
        for (int k = 1; k < n; ++k) b[k] = b[m - k] = std::conj(P->chirp[k]);
        P->chirpF.resize(m);
        rec(*P->sub, b.data(), 1, P->chirpF.data(), m, 0, 1);
    } else {
        P->tw.resize(n);
        for (int k = 0; k < n; ++k) P->tw[k] = unit_root(k, n, -1);
    }
    return P;
}

static std::shared_ptr<Plan> get_plan(int n)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_plans.find(n);
    if (it != g_plans.end()) return it->second;
    auto P = build_plan(n);
    g_plans[n] = P;
    return P;
}

// forward (sign -1) un-normalised complex FFT, contiguous in/out (may not alias)
static void fft_forward(const Plan &P, const cd *in, cd *out)
{
    if (P.n == 1) { out[0] = in[0]; return; }
    if (P.bluestein) bluestein_exec(P, in, out);
    else rec(P, in, 1, out, P.n, 0, 1);
}
static void fft_inverse(const Plan &P, const cd *in, cd *out)
{
    std::vector<cd> t(P.n), u(P.n);
    for (int i = 0; i < P.n; ++i) t[i] = std::conj(in[i]);
    fft_forward(P, t.data(), u.data());
    for (int i = 0; i < P.n; ++i) out[i] = std::conj(u[i]);
}
}  // namespace

namespace xo {
void c2c(const std::complex<double> *in, int n, int sign, std::complex<double> *out)
{
    auto P = get_plan(n);
    if (sign < 0) fft_forward(*P, in, out);
    else fft_inverse(*P, in, out);
}
}  // namespace xo

extern "C" {

void xo_fft1d_c2c(const double *in, int n, int sign, double *out)
{
    xo::c2c(reinterpret_cast<const cd *>(in), n, sign, reinterpret_cast<cd *>(out));
}

double xo_fft_idx2digfreq(int idx, int size)
{
    // FFT_IDX2DIGFREQ (xmippCore xmipp_fft.h; in-tree copy
    // reconstruction_cuda/cuda_gpu_reconstruct_fourier.cpp:381-385)
    if (size <= 1) return 0;
    return ((double)(idx <= size / 2 ? idx : idx - size)) / (double)size;
}

void xo_fft1d_r2c(const double *in, int n, double *out)
{
    std::vector<cd> a(n), A(n);
    for (int i = 0; i < n; ++i) a[i] = in[i];
    xo::c2c(a.data(), n, -1, A.data());
    const double inv = 1.0 / n;
    for (int k = 0; k <= n / 2; ++k) {
        out[2 * k] = A[k].real() * inv;
        out[2 * k + 1] = A[k].imag() * inv;
    }
}

void xo_fft1d_c2r(const double *in, int n, double *out)
{
    // c2r semantics of FFTW: the imaginary parts of the DC (and Nyquist for
    // even n) coefficients are ignored; negative frequencies are the conjugates.
    std::vector<cd> A(n), a(n);
    const cd *F = reinterpret_cast<const cd *>(in);
    A[0] = cd(F[0].real(), 0);
    for (int k = 1; k <= n / 2; ++k) {
        if (2 * k == n) A[k] = cd(F[k].real(), 0);
        else { A[k] = F[k]; A[n - k] = std::conj(F[k]); }
    }
    xo::c2c(A.data(), n, +1, a.data());
    for (int i = 0; i < n; ++i) out[i] = a[i].real();
}

void xo_fft2d_r2c(const double *in, int ydim, int xdim, double *out)
{
    const int xh = xdim / 2 + 1;
    std::vector<cd> rows((size_t)ydim * xh);
    std::vector<cd> a(std::max(xdim, ydim)), A(std::max(xdim, ydim));
    for (int i = 0; i < ydim; ++i) {
        for (int j = 0; j < xdim; ++j) a[j] = in[(size_t)i * xdim + j];
        xo::c2c(a.data(), xdim, -1, A.data());
        for (int j = 0; j < xh; ++j) rows[(size_t)i * xh + j] = A[j];
    }
    const double inv = 1.0 / ((double)xdim * ydim);
    cd *O = reinterpret_cast<cd *>(out);
    for (int j = 0; j < xh; ++j) {
        for (int i = 0; i < ydim; ++i) a[i] = rows[(size_t)i * xh + j];
        xo::c2c(a.data(), ydim, -1, A.data());
        for (int i = 0; i < ydim; ++i) O[(size_t)i * xh + j] = A[i] * inv;
    }
}

void xo_fft2d_c2r(const double *in, int ydim, int xdim, double *out)
{
    const int xh = xdim / 2 + 1;
    const cd *F = reinterpret_cast<const cd *>(in);
    std::vector<cd> cols((size_t)ydim * xh);
    std::vector<cd> a(std::max(xdim, ydim)), A(std::max(xdim, ydim));
    for (int j = 0; j < xh; ++j) {
        for (int i = 0; i < ydim; ++i) A[i] = F[(size_t)i * xh + j];
        xo::c2c(A.data(), ydim, +1, a.data());
        for (int i = 0; i < ydim; ++i) cols[(size_t)i * xh + j] = a[i];
    }
    std::vector<double> row(xdim);
    for (int i = 0; i < ydim; ++i) {
        xo_fft1d_c2r(reinterpret_cast<const double *>(&cols[(size_t)i * xh]), xdim, row.data());
        for (int j = 0; j < xdim; ++j) out[(size_t)i * xdim + j] = row[j];
    }
}

void xo_fft3d_c2r(const double *in, int zdim, int ydim, int xdim, double *out)
{
    const int xh = xdim / 2 + 1;
    const cd *F = reinterpret_cast<const cd *>(in);
    std::vector<cd> w((size_t)zdim * ydim * xh);
    // along z
#pragma omp parallel
    {
        std::vector<cd> a(std::max(zdim, ydim)), A(std::max(zdim, ydim));
#pragma omp for collapse(2)
        for (int i = 0; i < ydim; ++i)
            for (int j = 0; j < xh; ++j) {
                for (int k = 0; k < zdim; ++k) A[k] = F[((size_t)k * ydim + i) * xh + j];
                xo::c2c(A.data(), zdim, +1, a.data());
                for (int k = 0; k < zdim; ++k) w[((size_t)k * ydim + i) * xh + j] = a[k];
            }
#pragma omp for collapse(2)
        for (int k = 0; k < zdim; ++k)
            for (int j = 0; j < xh; ++j) {
                for (int i = 0; i < ydim; ++i) A[i] = w[((size_t)k * ydim + i) * xh + j];
                xo::c2c(A.data(), ydim, +1, a.data());
                for (int i = 0; i < ydim; ++i) w[((size_t)k * ydim + i) * xh + j] = a[i];
            }
    }
#pragma omp parallel for collapse(2)
    for (int k = 0; k < zdim; ++k)
        for (int i = 0; i < ydim; ++i)
            xo_fft1d_c2r(reinterpret_cast<const double *>(&w[((size_t)k * ydim + i) * xh]), xdim,
                         out + ((size_t)k * ydim + i) * xdim);
}

}  // extern "C"
