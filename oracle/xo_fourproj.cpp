// oracle/xo_fourproj.cpp -- FourierProjector: central-slice projection of a volume (the producer of the
// reference gallery, xmipp_angular_project_library --method fourier).  TEST INFRASTRUCTURE ONLY.
//
// Follows libraries/data/fourier_projection.cpp: produceSideInfo (L247-303), produceSideInfoProjection
// (L305-330), project (L91-245).  xmippCore pieces restated from their published behaviour (source not
// in the tree): FourierTransformer::completeFourierTransform (forward, divided by the number of voxels),
// ShiftFFT (multiply coefficient (k,i,j) by exp(-2 pi i (s_z k/Z + s_y i/Y + s_x j/X))), CenterFFT(.,true)
// (DC to index n/2), produceSplineCoefficients (xo::prefilter3d), inverse transform un-normalised.
// Parity of the whole operator is unpinned by the reference's tests; tests/test_oracle_pins.py pins it
// against the analytic projection of a Gaussian phantom instead (conventions: axis order, centring, sign).
#include "xo.h"
#include "xo_internal.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

typedef std::complex<double> cd;

struct xo_fp {
    int D, P;            // volumeSize, volumePaddedSize
    double maxFreq;
    int degree;          // 0 nearest, 1 linear, 3 cubic B-spline
    int cdim, cstart;    // cropped coefficient cube: size and STARTING index
    std::vector<double> re, im;        // [cdim][cdim][cdim]
    std::vector<double> phA, phB;      // [D][D/2+1]
};

static inline double bspline03(double x)
{
    double a = std::fabs(x);
    if (a < 1.0) return a * a * (a - 2.0) * 0.5 + 2.0 / 3.0;
    if (a < 2.0) { a -= 2.0; return a * a * a * (-1.0 / 6.0); }
    return 0.0;
}

extern "C" {

xo_fp *xo_fp_create(const double *vol, int D, double padding, double maxFreq, int degree)
{
    xo_fp *fp = new xo_fp;
    fp->D = D;
    fp->maxFreq = maxFreq;
    fp->degree = degree;
    const int P = (int)(padding * D);   // L251
    fp->P = P;
    const size_t P3 = (size_t)P * P * P;
    // volume->window(Vpadded, FIRST_XMIPP_INDEX(P) ...): zero padding about the Xmipp origin
    std::vector<cd> F(P3, cd(0, 0));
    const int o = xo::first_xmipp_index(D) - xo::first_xmipp_index(P);
    for (int k = 0; k < D; ++k)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j)
                F[((size_t)(k + o) * P + (i + o)) * P + (j + o)] = vol[((size_t)k * D + i) * D + j];
    // completeFourierTransform: forward DFT / P^3
    {
#pragma omp parallel
        {
            std::vector<cd> a(P), A(P);
#pragma omp for collapse(2)
            for (int k = 0; k < P; ++k)
                for (int i = 0; i < P; ++i) {
                    cd *line = &F[((size_t)k * P + i) * P];
                    xo::c2c(line, P, -1, A.data());
                    for (int j = 0; j < P; ++j) line[j] = A[j];
                }
#pragma omp for collapse(2)
            for (int k = 0; k < P; ++k)
                for (int j = 0; j < P; ++j) {
                    for (int i = 0; i < P; ++i) a[i] = F[((size_t)k * P + i) * P + j];
                    xo::c2c(a.data(), P, -1, A.data());
                    for (int i = 0; i < P; ++i) F[((size_t)k * P + i) * P + j] = A[i];
                }
#pragma omp for collapse(2)
            for (int i = 0; i < P; ++i)
                for (int j = 0; j < P; ++j) {
                    for (int k = 0; k < P; ++k) a[k] = F[((size_t)k * P + i) * P + j];
                    xo::c2c(a.data(), P, -1, A.data());
                    for (int k = 0; k < P; ++k) F[((size_t)k * P + i) * P + j] = A[k];
                }
        }
        const double inv = 1.0 / (double)P3;
        for (size_t n = 0; n < P3; ++n) F[n] *= inv;
    }
    // ShiftFFT(Vfourier, FIRST_XMIPP_INDEX(P) x3): the volume's origin goes from the centre to the corner
    {
        const double shift = xo::first_xmipp_index(P);
        const double xx = -2 * M_PI * shift / (double)P;
#pragma omp parallel for
        for (int k = 0; k < P; ++k)
            for (int i = 0; i < P; ++i)
                for (int j = 0; j < P; ++j) {
                    const double dotp = (double)j * xx + (double)i * xx + (double)k * xx;
                    F[((size_t)k * P + i) * P + j] *= cd(std::cos(dotp), std::sin(dotp));
                }
    }
    // CenterFFT(Vfourier, true) + setXmippOrigin; K = P^3 / D^2 (L263-265)
    const double K = (double)P * P * P / ((double)D * D);
    std::vector<double> re(P3), im(P3);
    const int h = P / 2;
#pragma omp parallel for
    for (int k = 0; k < P; ++k)
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) {
                const cd v = F[((size_t)k * P + i) * P + j] * K;
                const size_t d = ((size_t)((k + h) % P) * P + (i + h) % P) * P + (j + h) % P;
                re[d] = v.real();
                im[d] = v.imag();
            }
    F.clear();
    F.shrink_to_fit();
    int idxMax, idxMin;
    if (degree == 3) {
        xo::prefilter3d(re.data(), P, P, P);
        xo::prefilter3d(im.data(), P, P, P);
        idxMax = (int)(maxFreq * P + 10);                      // L281 (+10: safety guard)
        idxMax = std::min(xo::last_xmipp_index(P), idxMax);
        idxMin = std::max(-idxMax, xo::first_xmipp_index(P));
    } else {
        idxMin = xo::first_xmipp_index(P);
        idxMax = xo::last_xmipp_index(P);
    }
    const int c = idxMax - idxMin + 1;
    fp->cdim = c;
    fp->cstart = idxMin;
    fp->re.resize((size_t)c * c * c);
    fp->im.resize((size_t)c * c * c);
    const int po = idxMin - xo::first_xmipp_index(P);
    for (int k = 0; k < c; ++k)
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < c; ++j) {
                const size_t s = ((size_t)(k + po) * P + (i + po)) * P + (j + po);
                fp->re[((size_t)k * c + i) * c + j] = re[s];
                fp->im[((size_t)k * c + i) * c + j] = im[s];
            }
    // produceSideInfoProjection: phase shift that moves the image origin to the corner (L313-329)
    const int xh = D / 2 + 1;
    fp->phA.resize((size_t)D * xh);
    fp->phB.resize((size_t)D * xh);
    const double shift = -xo::first_xmipp_index(D);
    const double xxshift = -2 * M_PI * shift / D;
    for (int i = 0; i < D; ++i) {
        const double phasey = (double)i * xxshift;
        for (int j = 0; j < xh; ++j) {
            const double dotp = (double)j * xxshift + phasey;
            fp->phB[(size_t)i * xh + j] = std::sin(dotp);
            fp->phA[(size_t)i * xh + j] = std::cos(dotp);
        }
    }
    return fp;
}

void xo_fp_destroy(xo_fp *fp) { delete fp; }
int xo_fp_padded_size(const xo_fp *fp) { return fp->P; }
int xo_fp_coef_dim(const xo_fp *fp) { return fp->cdim; }
int xo_fp_coef_start(const xo_fp *fp) { return fp->cstart; }
const double *xo_fp_coefs(const xo_fp *fp, int imag) { return imag ? fp->im.data() : fp->re.data(); }

// project (L91-245); ctf: optional [D][D/2+1] multiplier; out: D x D, Xmipp origin
void xo_fp_project(const xo_fp *fp, double rot, double tilt, double psi, const double *ctf, double *out)
{
    const int D = fp->D, xh = D / 2 + 1, P = fp->P, c = fp->cdim, st = fp->cstart;
    double E[9];
    xo_euler_matrix(rot, tilt, psi, E);
    std::vector<cd> pf((size_t)D * xh, cd(0, 0));
    const double maxFreq2 = fp->maxFreq * fp->maxFreq;
    const double *R = fp->re.data(), *I = fp->im.data();
    for (int i = 0; i < D; ++i) {
        const double freqy = xo_fft_idx2digfreq(i, D);
        const double freqy2 = freqy * freqy;
        const double fyX = E[3] * freqy, fyY = E[4] * freqy, fyZ = E[5] * freqy;
        for (int j = 0; j < xh; ++j) {
            const double freqx = xo_fft_idx2digfreq(j, D);
            if ((freqy2 + freqx * freqx) > maxFreq2) continue;
            const double fX = fyX + E[0] * freqx, fY = fyY + E[1] * freqx, fZ = fyZ + E[2] * freqx;
            double cc = 0, dd = 0;
            if (fp->degree == 0) {
                const int kV = (int)std::round(fZ * P), iV = (int)std::round(fY * P), jV = (int)std::round(fX * P);
                const size_t s = ((size_t)(kV - st) * c + (iV - st)) * c + (jV - st);
                cc = R[s];
                dd = I[s];
            } else if (fp->degree == 3) {
                double z = fZ * P - st, y = fY * P - st, x = fX * P - st;   // logical to physical
                const int l1 = (int)std::ceil(x - 2), m1 = (int)std::ceil(y - 2), n1 = (int)std::ceil(z - 2);
                for (int nn = n1; nn <= n1 + 3; nn++) {
                    int en = nn;
                    if (nn < 0) en = -nn - 1;
                    else if (nn >= c) en = 2 * c - nn - 1;
                    double yxRe = 0, yxIm = 0;
                    for (int m = m1; m <= m1 + 3; m++) {
                        int em = m;
                        if (m < 0) em = -m - 1;
                        else if (m >= c) em = 2 * c - m - 1;
                        double xRe = 0, xIm = 0;
                        for (int l = l1; l <= l1 + 3; l++) {
                            int el = l;
                            if (l < 0) el = -l - 1;
                            else if (l >= c) el = 2 * c - l - 1;
                            const size_t s = ((size_t)en * c + em) * c + el;
                            const double aux = bspline03(x - (double)l);
                            xRe += R[s] * aux;
                            xIm += I[s] * aux;
                        }
                        const double aux = bspline03(y - (double)m);
                        yxRe += xRe * aux;
                        yxIm += xIm * aux;
                    }
                    const double aux = bspline03(z - (double)nn);
                    cc += yxRe * aux;
                    dd += yxIm * aux;
                }
            } else {
                // interpolatedElement3D (trilinear; outside the array counts as 0)
                const double z = fZ * P, y = fY * P, x = fX * P;
                const int x0 = (int)std::floor(x), y0 = (int)std::floor(y), z0 = (int)std::floor(z);
                const double fx = x - x0, fy = y - y0, fz = z - z0;
                for (int dz = 0; dz < 2; ++dz)
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int kk = z0 + dz - st, ii = y0 + dy - st, jj = x0 + dx - st;
                            if (kk < 0 || kk >= c || ii < 0 || ii >= c || jj < 0 || jj >= c) continue;
                            const double w = (dz ? fz : 1 - fz) * (dy ? fy : 1 - fy) * (dx ? fx : 1 - fx);
                            const size_t s = ((size_t)kk * c + ii) * c + jj;
                            cc += w * R[s];
                            dd += w * I[s];
                        }
            }
            double a = fp->phA[(size_t)i * xh + j], b = fp->phB[(size_t)i * xh + j];
            if (ctf) { a *= ctf[(size_t)i * xh + j]; b *= ctf[(size_t)i * xh + j]; }
            const double ac = a * cc, bd = b * dd, ab_cd = (a + b) * (cc + dd);
            pf[(size_t)i * xh + j] = cd(ac - bd, ab_cd - ac - bd);
        }
    }
    xo_fft2d_c2r(reinterpret_cast<const double *>(pf.data()), D, D, out);
}

}  // extern "C"
