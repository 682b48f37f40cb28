// oracle/xo_polar.cpp -- Polar<>, ring FFTs, rotationalCorrelation and the
// ProgAngularProjectionMatching search loops.  TEST INFRASTRUCTURE ONLY.
//
// Follows data/polar.h:488-534,625-738, data/polar.cpp:34-148 and
// reconstruction/angular_projection_matching.cpp:408-528 (getCurrentReference),
// :530-773 (threadRotationallyAlignOneImage), :776-868 (translationallyAlignOneImage),
// :991-1192 (processSomeImages).  Pins: test_polar_main.cpp:32-40.
#include "xo.h"
#include "xo_internal.h"
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
typedef std::complex<double> cd;
const double XO_TWOPI = 6.2831853071795864769;  // xmippCore TWOPI
const double XO_PI = 3.14159265358979323846;

inline int nsam_of(double twopi, double radius)
{
    // polar.h:723-726 getNoOfSamples with oversample = 1
    int result = 2 * (int)(0.5 * 1.0 * twopi * radius);
    return result > 1 ? result : 1;
}

struct Layout {
    int Ri, Ro, nrings;
    std::vector<int> nsam, soff, coff;  // samples per ring, sample offset, coef offset
    int nsamples, ncoefs;
    // float angle cache, polar.cpp:57-83
    std::vector<float> sinr, cosr;
    void init(int ri, int ro)
    {
        Ri = ri; Ro = ro; nrings = ro - ri + 1;
        nsam.resize(nrings); soff.resize(nrings); coff.resize(nrings);
        nsamples = ncoefs = 0;
        for (int r = 0; r < nrings; ++r) {
            float radius = (float)r + ri;
            nsam[r] = nsam_of(XO_TWOPI, radius);
            soff[r] = nsamples; coff[r] = ncoefs;
            nsamples += nsam[r];
            ncoefs += nsam[r] / 2 + 1;
        }
        sinr.resize(nsamples); cosr.resize(nsamples);
        for (int r = 0; r < nrings; ++r) {
            float radius = r + ri;
            int n = nsam[r];
            float dphi = XO_TWOPI / (float)n;
            for (int i = 0; i < n; ++i) {
                float phi = i * dphi;
                // polar.cpp:78-79: "sin(phi) * radius" with float phi/radius, stored as float.
                // <cmath> selects the float overload for a float argument.
                sinr[soff[r] + i] = std::sin(phi) * radius;
                cosr[soff[r] + i] = std::cos(phi) * radius;
            }
        }
    }
};

void polar_from_cartesian(const Layout &L, const double *coef, int ydim, int xdim, int starty,
                          int startx, double xoff, double yoff, double *rings)
{
    // polar.h:625-703
    const double minxp = xo::first_xmipp_index(xdim), minyp = xo::first_xmipp_index(ydim);
    const double maxxp = xo::last_xmipp_index(xdim), maxyp = xo::last_xmipp_index(ydim);
    const double minxp_e = minxp - XO_EQUAL_ACCURACY, minyp_e = minyp - XO_EQUAL_ACCURACY;
    const double maxxp_e = maxxp + XO_EQUAL_ACCURACY, maxyp_e = maxyp + XO_EQUAL_ACCURACY;
    for (int r = 0; r < L.nrings; ++r) {
        const int n = L.nsam[r];
        for (int s = 0; s < n; ++s) {
            double xp = L.sinr[L.soff[r] + s];
            double yp = L.cosr[L.soff[r] + s];
            xp += xoff;
            yp += yoff;
            if (xp < minxp_e || xp > maxxp_e) xp = xo::realWRAP(xp, minxp - 0.5, maxxp + 0.5);
            if (yp < minyp_e || yp > maxyp_e) yp = xo::realWRAP(yp, minyp - 0.5, maxyp + 0.5);
            rings[L.soff[r] + s] = xo::interp2d(coef, ydim, xdim, starty, startx, xp, yp);
        }
    }
}

void polar_avg_std(const Layout &L, const double *rings, double &avg, double &stddev)
{
    // polar.h:488-534, FULL_CIRCLES
    double sum = 0, sum2 = 0, N = 0;
    const double twopi = 2. * XO_PI;
    for (int i = 0; i < L.nrings; ++i) {
        const double radius = (float)i + L.Ri;
        const double w = (twopi * radius) / (double)L.nsam[i];
        for (int j = 0; j < L.nsam[i]; ++j) {
            double aux = rings[L.soff[i] + j];
            double waux = w * aux;
            sum += waux;
            sum2 += waux * aux;
            N += w;
        }
    }
    if (N > 0) {
        sum2 = sum2 / N;
        avg = sum / N;
        stddev = std::sqrt(std::fabs(sum2 - avg * avg));
    } else if (N != 0.) {
        avg = sum;
        stddev = std::sqrt(std::fabs(sum2 - avg * avg));
    } else stddev = avg = 0;
}

struct Scratch {
    std::vector<cd> a, A;
    void need(int n) { if ((int)a.size() < n) { a.resize(n); A.resize(n); } }
};

void fft_rings(const Layout &L, const double *rings, bool conjugated, cd *coefs, Scratch &S)
{
    // polar.cpp:34-54; FourierTransformer forward = FFTW r2c / nsam
    for (int r = 0; r < L.nrings; ++r) {
        const int n = L.nsam[r];
        S.need(n);
        for (int i = 0; i < n; ++i) S.a[i] = rings[L.soff[r] + i];
        xo::c2c(S.a.data(), n, -1, S.A.data());
        const double inv = 1.0 / n;
        for (int k = 0; k <= n / 2; ++k) {
            cd v = S.A[k] * inv;
            if (conjugated) v = cd(v.real(), v.imag() * -1);
            coefs[L.coff[r] + k] = v;
        }
    }
}

// rotationalCorrelation, polar.cpp:99-148. corr has N = nsam(outer ring) entries.
void rot_corr(const Layout &L, const cd *M1, const cd *M2, double *corr, Scratch &S)
{
    const int N = L.nsam[L.nrings - 1];
    const int nh = N / 2 + 1;
    S.need(N);
    std::vector<double> Fsum(2 * (size_t)nh, 0.0);
    for (int iring = 0; iring < L.nrings; ++iring) {
        const double radius = (float)iring + L.Ri;
        const double w = (2. * XO_PI * radius);
        const int imax = L.nsam[iring] / 2 + 1;
        const double *ptr1 = reinterpret_cast<const double *>(M1 + L.coff[iring]);
        const double *ptr2 = reinterpret_cast<const double *>(M2 + L.coff[iring]);
        double *ptrFsum = Fsum.data();
        for (int i = 0; i < imax; i++) {
            double a = *ptr1++;
            double b = *ptr1++;
            double c = *ptr2++;
            double d = *ptr2++;
            *(ptrFsum++) += w * (a * c - b * d);
            *(ptrFsum++) += w * (b * c + a * d);
        }
    }
    // inverse c2r (un-normalised); DC/Nyquist imaginary parts are ignored by c2r
    cd *A = S.A.data();
    cd *a = S.a.data();
    A[0] = cd(Fsum[0], 0);
    for (int k = 1; k < nh; ++k) {
        cd f(Fsum[2 * k], Fsum[2 * k + 1]);
        if (2 * k == N) A[k] = cd(f.real(), 0);
        else { A[k] = f; A[N - k] = std::conj(f); }
    }
    xo::c2c(A, N, +1, a);
    for (int i = 0; i < N; ++i) corr[i] = a[i].real();
}
}  // namespace

struct xo_pm {
    int D, Ri, Ro, nrefs;
    Layout L;
    std::vector<cd> fP_ref;          // nrefs * ncoefs (conjugated)
    std::vector<double> stddev_ref;  // nrefs
    std::vector<double> proj_ref;    // nrefs * D * D (after optional CTF)
};

extern "C" {

int xo_polar_nsam(int radius) { return nsam_of(XO_TWOPI, (float)radius); }

void xo_polar_layout(int Ri, int Ro, int *nsam, int *total_samples, int *total_coefs)
{
    Layout L;
    L.init(Ri, Ro);
    if (nsam) for (int r = 0; r < L.nrings; ++r) nsam[r] = L.nsam[r];
    if (total_samples) *total_samples = L.nsamples;
    if (total_coefs) *total_coefs = L.ncoefs;
}

void xo_polar_from_cartesian_bspline(const double *coef, int ydim, int xdim, int starty, int startx,
                                     int Ri, int Ro, double xoff, double yoff, double *rings)
{
    Layout L;
    L.init(Ri, Ro);
    polar_from_cartesian(L, coef, ydim, xdim, starty, startx, xoff, yoff, rings);
}

void xo_polar_avg_std(const double *rings, int Ri, int Ro, double *avg, double *stddev)
{
    Layout L;
    L.init(Ri, Ro);
    polar_avg_std(L, rings, *avg, *stddev);
}

void xo_polar_fft_rings(const double *rings, int Ri, int Ro, int conjugated, double *coefs)
{
    Layout L;
    L.init(Ri, Ro);
    Scratch S;
    fft_rings(L, rings, conjugated != 0, reinterpret_cast<cd *>(coefs), S);
}

void xo_rotational_correlation(const double *F1, const double *F2, int Ri, int Ro, double *corr)
{
    Layout L;
    L.init(Ri, Ro);
    Scratch S;
    rot_corr(L, reinterpret_cast<const cd *>(F1), reinterpret_cast<const cd *>(F2), corr, S);
}

xo_pm *xo_pm_create(int D, int Ri, int Ro, int nrefs, const double *refs, const double *Mctf,
                    int paddim)
{
    // APM:262-274 ring defaults; APM:408-528 getCurrentReference for every reference
    xo_pm *pm = new xo_pm;
    pm->D = D;
    if (Ri < 1) Ri = 1;
    if (Ro < 0) Ro = (D / 2) - 1;
    pm->Ri = Ri; pm->Ro = Ro; pm->nrefs = nrefs;
    pm->L.init(Ri, Ro);
    const Layout &L = pm->L;
    pm->fP_ref.resize((size_t)nrefs * L.ncoefs);
    pm->stddev_ref.resize(nrefs);
    pm->proj_ref.assign(refs, refs + (size_t)nrefs * D * D);
    const int start = xo::first_xmipp_index(D);
#pragma omp parallel
    {
        Scratch S;
        std::vector<double> Maux((size_t)D * D), P(L.nsamples);
#pragma omp for schedule(dynamic)
        for (int r = 0; r < nrefs; ++r) {
            double *img = &pm->proj_ref[(size_t)r * D * D];
            if (Mctf) {
                // APM:457-481: window to paddim, FFT, multiply by Mctf (index-wise on the
                // half spectrum), IFFT, window back
                const int P0 = paddim;
                std::vector<double> pad((size_t)P0 * P0, 0.0), F((size_t)P0 * (P0 / 2 + 1) * 2);
                const int x0 = xo::first_xmipp_index(P0);
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        pad[(size_t)(i + start - x0) * P0 + (j + start - x0)] = img[(size_t)i * D + j];
                xo_fft2d_r2c(pad.data(), P0, P0, F.data());
                const int xh = P0 / 2 + 1;
                for (int i = 0; i < P0; ++i)
                    for (int j = 0; j < xh; ++j) {
                        F[2 * ((size_t)i * xh + j)] *= Mctf[(size_t)i * P0 + j];
                        F[2 * ((size_t)i * xh + j) + 1] *= Mctf[(size_t)i * P0 + j];
                    }
                xo_fft2d_c2r(F.data(), P0, P0, pad.data());
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        img[(size_t)i * D + j] = pad[(size_t)(i + start - x0) * P0 + (j + start - x0)];
            }
            xo::prefilter2d(img, D, D, Maux.data());
            polar_from_cartesian(L, Maux.data(), D, D, start, start, 0., 0., P.data());
            double mean, stddev;
            polar_avg_std(L, P.data(), mean, stddev);
            for (int i = 0; i < L.nsamples; ++i) P[i] -= mean;
            fft_rings(L, P.data(), true, &pm->fP_ref[(size_t)r * L.ncoefs], S);
            pm->stddev_ref[r] = stddev;
        }
    }
    return pm;
}

void xo_pm_destroy(xo_pm *pm) { delete pm; }
int xo_pm_nsam_outer(const xo_pm *pm) { return pm->L.nsam[pm->L.nrings - 1]; }
int xo_pm_ncoef(const xo_pm *pm) { return pm->L.ncoefs; }
const double *xo_pm_ref_coefs(const xo_pm *pm, int ref)
{
    return reinterpret_cast<const double *>(&pm->fP_ref[(size_t)ref * pm->L.ncoefs]);
}
double xo_pm_ref_sigma(const xo_pm *pm, int ref) { return pm->stddev_ref[ref]; }

static void prepare_particle(const xo_pm *pm, const double *img, double xoff, double yoff, cd *fP,
                             cd *fPm, double *sigma, Scratch &S, std::vector<double> &Maux,
                             std::vector<double> &P, bool havePrefilter)
{
    // APM:569-597
    const Layout &L = pm->L;
    const int D = pm->D, start = xo::first_xmipp_index(D);
    if (!havePrefilter) xo::prefilter2d(img, D, D, Maux.data());
    polar_from_cartesian(L, Maux.data(), D, D, start, start, xoff, yoff, P.data());
    double mean, stddev;
    polar_avg_std(L, P.data(), mean, stddev);
    for (int i = 0; i < L.nsamples; ++i) P[i] -= mean;
    fft_rings(L, P.data(), false, fP, S);
    fft_rings(L, P.data(), true, fPm, S);
    *sigma = stddev;
}

void xo_pm_prepare_particle(const xo_pm *pm, const double *img, double xoff, double yoff, double *fP,
                            double *fPm, double *sigma)
{
    Scratch S;
    std::vector<double> Maux((size_t)pm->D * pm->D), P(pm->L.nsamples);
    prepare_particle(pm, img, xoff, yoff, reinterpret_cast<cd *>(fP), reinterpret_cast<cd *>(fPm),
                     sigma, S, Maux, P, false);
}

void xo_pm_corr_rows(const xo_pm *pm, const double *img, int ref, double *corr2N)
{
    const Layout &L = pm->L;
    const int N = L.nsam[L.nrings - 1];
    Scratch S;
    std::vector<double> Maux((size_t)pm->D * pm->D), P(L.nsamples);
    std::vector<cd> fP(L.ncoefs), fPm(L.ncoefs);
    double sigma;
    prepare_particle(pm, img, 0, 0, fP.data(), fPm.data(), &sigma, S, Maux, P, false);
    const cd *fr = &pm->fP_ref[(size_t)ref * L.ncoefs];
    rot_corr(L, fP.data(), fr, corr2N, S);
    rot_corr(L, fPm.data(), fr, corr2N + N, S);
    const double den = pm->stddev_ref[ref] * sigma;
    for (int i = 0; i < 2 * N; ++i) corr2N[i] /= den;
}

void xo_pm_match(const xo_pm *pm, const double *particles, int n, const int32_t *nbr_off,
                 const int32_t *nbr_ids, int first_image_parity, int n_orient,
                 const int32_t *xoff5d, const int32_t *yoff5d, int ntrans, int nthreads,
                 int32_t *refno, int32_t *psi_idx, uint8_t *flip, double *cc)
{
    xo_pm_match_thr(pm, particles, n, nbr_off, nbr_ids, first_image_parity, n_orient, xoff5d, yoff5d, ntrans, nthreads, 1, refno, psi_idx, flip, cc);
}

// ref_threads = the program's --thr (APM:64,537,631): worker c of ref_threads takes the list positions i with i % ref_threads == c, in
// the image's visiting order, and keeps its own running top-N (APM:714-735); the lists are merged afterwards (APM:1063-1108): rank n of
// the result is the head of the list whose head is strictly greatest, the lowest worker among equals.  ref_threads == 1 is the plain loop.
void xo_pm_match_thr(const xo_pm *pm, const double *particles, int n, const int32_t *nbr_off,
                     const int32_t *nbr_ids, int first_image_parity, int n_orient,
                     const int32_t *xoff5d, const int32_t *yoff5d, int ntrans, int nthreads, int ref_threads,
                     int32_t *refno, int32_t *psi_idx, uint8_t *flip, double *cc)
{
    if (ref_threads < 1) ref_threads = 1;
    const Layout &L = pm->L;
    const int N = L.nsam[L.nrings - 1];
    const int D = pm->D;
    const int32_t zero = 0;
    if (ntrans <= 0 || !xoff5d) { xoff5d = &zero; yoff5d = &zero; ntrans = 1; }
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        Scratch S;
        std::vector<double> Maux((size_t)D * D), P(L.nsamples), corr(N), allCorr(2 * (size_t)N);
        std::vector<cd> fP((size_t)ntrans * L.ncoefs), fPm((size_t)ntrans * L.ncoefs);
        std::vector<double> stddev_img(ntrans);
        std::vector<double> maxcorr(n_orient);
        // --thr > 1: worker c's running list (APM:1026-1056)
        const int T = ref_threads;
        std::vector<double> tcorr((size_t)T * n_orient);
        std::vector<int32_t> tref((size_t)T * n_orient), tpsi((size_t)T * n_orient);
        std::vector<uint8_t> tflip((size_t)T * n_orient);
#pragma omp for schedule(dynamic, 1)
        for (int imgno = 0; imgno < n; ++imgno) {
            const double *img = particles + (size_t)imgno * D * D;
            // APM:1049-1055
            for (int i = 0; i < n_orient; ++i) {
                maxcorr[i] = -99.e99;
                refno[(size_t)imgno * n_orient + i] = -1;
                psi_idx[(size_t)imgno * n_orient + i] = 0;
                flip[(size_t)imgno * n_orient + i] = 0;
            }
            xo::prefilter2d(img, D, D, Maux.data());
            for (int it = 0; it < ntrans; ++it)
                prepare_particle(pm, img, (double)xoff5d[it], (double)yoff5d[it],
                                 &fP[(size_t)it * L.ncoefs], &fPm[(size_t)it * L.ncoefs],
                                 &stddev_img[it], S, Maux, P, true);
            // neighbour list & visiting order (APM:609-626; flag flips per image APM:1112)
            int nn;
            const int32_t *ids = nullptr;
            if (nbr_off) { nn = nbr_off[imgno + 1] - nbr_off[imgno]; ids = nbr_ids + nbr_off[imgno]; }
            else nn = pm->nrefs;
            const bool forward = (((imgno + first_image_parity) & 1) == 0);
            if (T > 1) {
                for (int c = 0; c < T; ++c)
                    for (int no = 0; no < n_orient; ++no) {
                        tcorr[(size_t)c * n_orient + no] = -99.e99; tref[(size_t)c * n_orient + no] = -1;
                        tpsi[(size_t)c * n_orient + no] = 0; tflip[(size_t)c * n_orient + no] = 0;
                    }
                for (int c = 0; c < T; ++c) {
                    double *mc = &tcorr[(size_t)c * n_orient];
                    for (int t = 0; t < nn; ++t) {
                        const int i = forward ? t : nn - 1 - t;
                        if (i % T != c) continue;                                  // APM:631
                        const int ref = ids ? ids[i] : i;
                        const cd *fr = &pm->fP_ref[(size_t)ref * L.ncoefs];
                        for (int it = 0; it < ntrans; ++it) {
                            const double den = pm->stddev_ref[ref] * stddev_img[it];
                            rot_corr(L, &fP[(size_t)it * L.ncoefs], fr, corr.data(), S);
                            for (int k = 0; k < N; ++k) allCorr[k] = corr[k] / den;
                            rot_corr(L, &fPm[(size_t)it * L.ncoefs], fr, corr.data(), S);
                            for (int k = 0; k < N; ++k) allCorr[N + k] = corr[k] / den;
                            const int nIter = n_orient < N ? n_orient : N;
                            double bestLastCorr = 99e99;
                            for (int no = 0; no < nIter; no++) {
                                for (int k = 0; k < 2 * N; k++)
                                    if ((allCorr[k] > mc[no]) && (allCorr[k] < bestLastCorr)) {
                                        mc[no] = allCorr[k];
                                        tpsi[(size_t)c * n_orient + no] = k % N;
                                        tref[(size_t)c * n_orient + no] = ref;
                                        tflip[(size_t)c * n_orient + no] = (k >= N);
                                    }
                                bestLastCorr = mc[no];
                            }
                        }
                    }
                }
                // APM:1063-1108
                std::vector<int> indexThreads(T, 0);
                for (int no = 0; no < n_orient; ++no) {
                    double tempCorr = -99.e99;
                    bool validCorr = false;
                    int best = 0;
                    for (int c = 0; c < T; ++c) {
                        // (a worker that has handed out all its ranks has nothing left: the reference reads one past its list here)
                        if (indexThreads[c] >= n_orient) continue;
                        if (tcorr[(size_t)c * n_orient + indexThreads[c]] > tempCorr) {
                            validCorr = true;
                            best = c;
                            tempCorr = tcorr[(size_t)c * n_orient + indexThreads[c]];
                        }
                    }
                    if (!validCorr) break;
                    const size_t src = (size_t)best * n_orient + indexThreads[best];
                    refno[(size_t)imgno * n_orient + no] = tref[src];
                    psi_idx[(size_t)imgno * n_orient + no] = tpsi[src];
                    flip[(size_t)imgno * n_orient + no] = tflip[src];
                    maxcorr[no] = tcorr[src];
                    indexThreads[best]++;
                }
                for (int i = 0; i < n_orient; ++i) cc[(size_t)imgno * n_orient + i] = maxcorr[i];
                continue;
            }
            for (int t = 0; t < nn; ++t) {
                const int i = forward ? t : nn - 1 - t;
                const int ref = ids ? ids[i] : i;
                const cd *fr = &pm->fP_ref[(size_t)ref * L.ncoefs];
                for (int it = 0; it < ntrans; ++it) {
                    const double den = pm->stddev_ref[ref] * stddev_img[it];
                    rot_corr(L, &fP[(size_t)it * L.ncoefs], fr, corr.data(), S);
                    for (int k = 0; k < N; ++k) allCorr[k] = corr[k] / den;
                    rot_corr(L, &fPm[(size_t)it * L.ncoefs], fr, corr.data(), S);
                    for (int k = 0; k < N; ++k) allCorr[N + k] = corr[k] / den;
                    // APM:714-735
                    const int nIter = n_orient < N ? n_orient : N;
                    double bestLastCorr = 99e99;
                    for (int no = 0; no < nIter; no++) {
                        for (int k = 0; k < 2 * N; k++) {
                            if ((allCorr[k] > maxcorr[no]) && (allCorr[k] < bestLastCorr)) {
                                maxcorr[no] = allCorr[k];
                                psi_idx[(size_t)imgno * n_orient + no] = k % N;
                                refno[(size_t)imgno * n_orient + no] = ref;
                                flip[(size_t)imgno * n_orient + no] = (k >= N);
                            }
                        }
                        bestLastCorr = maxcorr[no];
                    }
                }
            }
            for (int i = 0; i < n_orient; ++i) cc[(size_t)imgno * n_orient + i] = maxcorr[i];
        }
    }
}

void xo_pm_translate(const xo_pm *pm, const double *particles, int n, const int32_t *refno,
                     const int32_t *psi_idx, const uint8_t *flipv, double max_shift, int nthreads,
                     double *shiftX, double *shiftY, double *maxCC)
{
    // APM:776-868
    const int D = pm->D;
    const int N = pm->L.nsam[pm->L.nrings - 1];
    if (max_shift < 0) max_shift = D / 2;  // APM:262-263
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<double> Mref((size_t)D * D), Mimg((size_t)D * D), Mtrans((size_t)D * D),
            R((size_t)D * D);
#pragma omp for schedule(dynamic, 1)
        for (int p = 0; p < n; ++p) {
            if (refno[p] < 0) { shiftX[p] = shiftY[p] = 0; maxCC[p] = 0; continue; }
            const double *img = particles + (size_t)p * D * D;
            const double *ref = &pm->proj_ref[(size_t)refno[p] * D * D];
            const double opt_psi = (double)psi_idx[p] * (360. / N);  // polar.cpp:145-147
            xo_rotate2d(3, ref, D, D, opt_psi, 0, Mref.data());
            if (flipv[p]) {
                double A[9] = {-1, 0, 0, 0, 1, 0, 0, 0, 1};
                xo::apply_geometry2d(1, img, D, D, A, true, false, Mimg.data());
            } else std::memcpy(Mimg.data(), img, sizeof(double) * (size_t)D * D);
            double ox = 0, oy = 0;
            if (max_shift > 0) {
                xo::correlation_matrix(Mref.data(), Mimg.data(), D, D, R.data());
                xo::best_shift_mcorr(R.data(), D, D, -1, ox, oy);
            }
            if (ox * ox + oy * oy > max_shift * max_shift) ox = oy = 0.;
            xo_translate2d(1, Mimg.data(), D, D, ox, oy, 1, Mtrans.data());
            maxCC[p] = xo::correlation_index(Mref.data(), Mtrans.data(), (size_t)D * D);
            if (flipv[p]) ox *= -1.;
            shiftX[p] = ox;
            shiftY[p] = oy;
        }
    }
}

int xo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}

/* ---- PolarRotationEstimator (reconstruction/polar_rotation_estimator.cpp:49-99) -------------
 * load2DReferenceOneToN: polarFourierTransform<false>(ref, ..., conjugated = false, firstRing, lastRing, plans, BsplineOrder = 1)
 * computeRotation2DOneToN: the same with conjugated = true per image, then best_rotation (polar.cpp:212-233) over a correlation
 * array of 2 * getSampleNoOuterRing() - 1 entries (:58).  BsplineOrder 1 = interpolatedElement2DOutsideZero (xmippCore
 * multidim_array.h: bilinear, samples outside the image are 0; LIN_INTERP(a, l, h) = l + (h - l) * a), no normalisation. */
static double es_interp_outside_zero(const double *img, int ydim, int xdim, double x, double y)
{
    const int x0 = (int)std::floor(x);
    const double fx = x - x0;
    const int x1 = x0 + 1;
    const int y0 = (int)std::floor(y);
    const double fy = y - y0;
    const int y1 = y0 + 1;
    const int i0 = xo::first_xmipp_index(ydim), j0 = xo::first_xmipp_index(xdim);
    const int iF = i0 + ydim - 1, jF = j0 + xdim - 1;
    auto at = [&](int i, int j) -> double {
        if (j < j0 || j > jF || i < i0 || i > iF) return 0.0;
        return img[(size_t)(i - i0) * xdim + (j - j0)];
    };
    const double d00 = at(y0, x0), d01 = at(y0, x1), d10 = at(y1, x0), d11 = at(y1, x1);
    const double d0 = d00 + (d01 - d00) * fx;
    const double d1 = d10 + (d11 - d10) * fx;
    return d0 + (d1 - d0) * fy;
}

static void es_polar_linear(const Layout &L, const double *img, int D, double *rings)
{
    // polar.h:625-703 with BsplineOrder == 1
    const double minp = xo::first_xmipp_index(D), maxp = xo::last_xmipp_index(D);
    const double min_e = minp - XO_EQUAL_ACCURACY, max_e = maxp + XO_EQUAL_ACCURACY;
    for (int r = 0; r < L.nrings; ++r)
        for (int s = 0; s < L.nsam[r]; ++s) {
            double xp = L.sinr[L.soff[r] + s], yp = L.cosr[L.soff[r] + s];
            if (xp < min_e || xp > max_e) xp = xo::realWRAP(xp, minp - 0.5, maxp + 0.5);
            if (yp < min_e || yp > max_e) yp = xo::realWRAP(yp, minp - 0.5, maxp + 0.5);
            rings[L.soff[r] + s] = es_interp_outside_zero(img, D, D, xp, yp);
        }
}

// rotationalCorrelation (polar.cpp:99-148) into a correlation array of `len` entries (Fsum has len / 2 + 1 of them, every ring
// adds its nsam / 2 + 1 coefficients), inverse transform un-normalised
static void es_rot_corr(const Layout &L, const cd *M1, const cd *M2, int len, double *corr)
{
    const int nh = len / 2 + 1;
    std::vector<double> Fsum(2 * (size_t)nh, 0.0);
    for (int iring = 0; iring < L.nrings; ++iring) {
        const double radius = (float)iring + L.Ri;
        const double w = (2. * XO_PI * radius);
        const int imax = L.nsam[iring] / 2 + 1;
        const double *ptr1 = reinterpret_cast<const double *>(M1 + L.coff[iring]);
        const double *ptr2 = reinterpret_cast<const double *>(M2 + L.coff[iring]);
        double *ptrFsum = Fsum.data();
        for (int i = 0; i < imax && i < nh; i++) {
            double a = *ptr1++;
            double b = *ptr1++;
            double c = *ptr2++;
            double d = *ptr2++;
            *(ptrFsum++) += w * (a * c - b * d);
            *(ptrFsum++) += w * (b * c + a * d);
        }
    }
    std::vector<cd> A(len), a(len);
    A[0] = cd(Fsum[0], 0);
    for (int k = 1; k < nh; ++k) {
        cd f(Fsum[2 * k], Fsum[2 * k + 1]);
        if (2 * k == len) A[k] = cd(f.real(), 0);
        else { A[k] = f; A[len - k] = std::conj(f); }
    }
    xo::c2c(A.data(), len, +1, a.data());
    for (int i = 0; i < len; ++i) corr[i] = a[i].real();
}

extern "C" {

int xo_es_rotation_corr_len(int first_ring, int last_ring)
{
    (void)first_ring;
    return 2 * nsam_of(XO_TWOPI, (float)last_ring) - 1;          // polar_rotation_estimator.cpp:58
}

// ref [D][D], others [n][D][D] (the float images of the estimator, handed over as doubles: convert(), :61-68);
// rotations [n] in degrees = angles[imax] = imax * 360 / len (polar.cpp:143-146,218-232: the first maximum);
// corr_out (optional) [n][len]
void xo_es_polar_rotation(const double *ref, const double *others, int n, int D, int first_ring, int last_ring, double *rotations,
                          double *corr_out)
{
    Layout L;
    L.init(first_ring, last_ring);
    const int len = 2 * L.nsam[L.nrings - 1] - 1;
    std::vector<double> rings(L.nsamples);
    std::vector<cd> Fref(L.ncoefs);
    Scratch S;
    es_polar_linear(L, ref, D, rings.data());
    fft_rings(L, rings.data(), false, Fref.data(), S);
#pragma omp parallel
    {
        std::vector<double> rg(L.nsamples), corr(len);
        std::vector<cd> F(L.ncoefs);
        Scratch S2;
#pragma omp for schedule(dynamic)
        for (int i = 0; i < n; ++i) {
            es_polar_linear(L, others + (size_t)i * D * D, D, rg.data());
            fft_rings(L, rg.data(), true, F.data(), S2);
            es_rot_corr(L, Fref.data(), F.data(), len, corr.data());
            int imax = 0;
            double maxval = corr[0];
            for (int k = 0; k < len; ++k)
                if (corr[k] > maxval) { maxval = corr[k]; imax = k; }
            rotations[i] = (double)imax * (360. / len);
            if (corr_out) std::memcpy(corr_out + (size_t)i * len, corr.data(), sizeof(double) * len);
        }
    }
}

}  // extern "C"
