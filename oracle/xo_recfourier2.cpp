// oracle/xo_recfourier2.cpp -- CPU restatement of ProgRecFourier, the double-precision scatter variant behind the name
// xmipp_reconstruct_fourier (reconstruction/reconstruct_fourier.cpp, "RF"). TEST INFRASTRUCTURE ONLY: nothing under
// xmipp3_amd/ links, imports or calls it. It exists to MEASURE how far the accel arithmetic (RFA, what the device
// implements under both program names) is from RF's on the same inputs, and what `--iter` (NiterWeight) does.
//
// Follows RF line by line:
//   tables                RF:222-269   (the same Kaiser-Bessel tables as RFA, kept in double)
//   image preparation     RF:386-404   zero-pad about the Xmipp origin, CenterFFT(true), r2c normalised by 1/N
//   scatter               RF:571-793   for every Fourier pixel within maxResolution: rotate the digital frequency by A_SL,
//                                      index = freq * volPadSize, box of +-radius, w = table[(int)(d2 iDelta + 0.5)] weight mod,
//                                      FFTW layout with wrap; x beyond the half -> the point-mirrored, conjugated slot
//   correctWeight         RF:1056-1101 NiterWeight iterations of w <- w / (scatter of w), forceWeightSymmetry RF:1186-1221
//   finishComputations    RF:1103-1178 enforceHermitianSymmetry (xmippCore, absent: restated from its published source,
//                                      unpinned), PROCESS_WEIGHTS RF:451-480, inverse FFT, CenterFFT, window, blob/sinc
// Parity status: unpinned above the xmippCore boundary (no unit test of ProgRecFourier holds numbers), like RFA.
#include "xo.h"
#include "xo_internal.h"
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {
const double PI2 = 3.14159265358979323846;
const int TABLE = 10000;

inline int int_wrap(int x, int x0, int xF)
{
    // intWRAP of xmippCore: wrap x into [x0, xF]
    const int len = xF - x0 + 1;
    int r = (x - x0) % len;
    if (r < 0) r += len;
    return r + x0;
}
inline double idx2digfreq(int idx, int size)
{
    // FFT_IDX2DIGFREQ (test_fftw_main.cpp:80-109)
    return (idx <= (size >> 1)) ? (double)idx / size : (double)(idx - size) / size;
}
}  // namespace

struct xo_rf2 {
    int D, P, V;                     // image size, padded image size, padded volume size
    double pad_proj, pad_vol, maxRes2, radius, iDeltaSqrt, iDeltaFourier;
    int niter;
    std::vector<double> table, ftable;
    std::vector<std::complex<double>> F, Fsave;   // [V][V][V/2+1]
    std::vector<double> W;
    int xh;
    inline size_t at(int k, int i, int j) const { return ((size_t)k * V + i) * xh + j; }
};

extern "C" {

xo_rf2 *xo_rf2_create(int D, double pad_proj, double pad_vol, double max_resolution, double blob_radius, int blob_order,
                      double blob_alpha, int niter_weight)
{
    xo_rf2 *R = new xo_rf2;
    R->D = D; R->pad_proj = pad_proj; R->pad_vol = pad_vol;
    R->P = (int)(D * pad_proj);
    R->V = (int)(D * pad_vol);
    R->xh = R->V / 2 + 1;
    R->maxRes2 = max_resolution * max_resolution;
    R->radius = blob_radius;
    R->niter = niter_weight;
    R->table.resize(TABLE); R->ftable.resize(TABLE);
    const double rF = blob_radius / (pad_vol * D), rN = blob_radius / (pad_proj / pad_vol);
    const double deltaSqrt = blob_radius * blob_radius / (TABLE - 1), deltaFourier = (std::sqrt(3.) * D / 2.) / (TABLE - 1);
    const double iw0 = 1.0 / xo_kaiser_fourier_value(0.0, rN, blob_alpha, blob_order);
    double pad3 = pad_vol * D; pad3 = pad3 * pad3 * pad3;
    const double tsz = blob_radius * std::sqrt(1. / (TABLE - 1));
    for (int i = 0; i < TABLE; ++i) {
        R->table[i] = xo_kaiser_value(tsz * std::sqrt((double)i), blob_radius, blob_alpha, blob_order) * iw0;
        R->ftable[i] = xo_kaiser_fourier_value(deltaFourier * i, rF, blob_alpha, blob_order) * pad3 * iw0;
    }
    R->iDeltaSqrt = 1 / deltaSqrt; R->iDeltaFourier = 1 / deltaFourier;
    const size_t n = (size_t)R->V * R->V * R->xh;
    R->F.assign(n, std::complex<double>(0, 0));
    R->W.assign(n, 0.0);
    return R;
}
void xo_rf2_destroy(xo_rf2 *R) { delete R; }
double *xo_rf2_weights(xo_rf2 *R) { return R->W.data(); }
double *xo_rf2_fourier(xo_rf2 *R) { return reinterpret_cast<double *>(R->F.data()); }
int xo_rf2_vol_pad(const xo_rf2 *R) { return R->V; }

// One image x one symmetry matrix. img: D*D doubles (shifts already applied; ignored when reprocess != 0),
// A_SL = R * localAInv (RF:560-566 forms it per symmetry), 9 doubles row-major. ctf may be NULL.
void xo_rf2_insert(xo_rf2 *R, const double *img, const double *localAInv, const double *Rsym, double weight,
                   const xo_ctf_params *ctf, double iTs, double minCTF, int phaseFlipped, int reprocess)
{
    if (weight == 0.0) return;
    const int D = R->D, P = R->P, V = R->V, pxh = P / 2 + 1;
    std::vector<std::complex<double>> pf((size_t)P * pxh, std::complex<double>(0, 0));
    if (!reprocess) {
        // RF:386-404
        std::vector<double> padded((size_t)P * P, 0.0), shifted((size_t)P * P);
        const int s0 = xo::first_xmipp_index(D), p0 = xo::first_xmipp_index(P);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) padded[(size_t)(i + s0 - p0) * P + (j + s0 - p0)] = img[(size_t)i * D + j];
        const int sh = P / 2;      // CenterFFT(forward): circular shift by +n/2 (Appendix B of SURVEY.md)
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) shifted[(size_t)((i + sh) % P) * P + ((j + sh) % P)] = padded[(size_t)i * P + j];
        xo_fft2d_r2c(shifted.data(), P, P, reinterpret_cast<double *>(pf.data()));     // normalised by 1/(P*P)
    }
    double A[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += Rsym[r * 3 + k] * localAInv[k * 3 + c];
            A[r * 3 + c] = acc;
        }
    const double r2 = R->radius * R->radius;
    const int xsize_1 = R->xh - 1, zsize_1 = V - 1;
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < pxh; ++j) {
            double fx = idx2digfreq(j, P), fy = idx2digfreq(i, P);
            if (fx * fx + fy * fy > R->maxRes2) continue;
            double wCTF = 1, wMod = 1;
            if (ctf && !reprocess) {
                wCTF = xo_ctf_value_pure_nok(ctf, fx * iTs, fy * iTs);
                if (std::isnan(wCTF)) { if (i == 0 && j == 0) wMod = wCTF = 1.0; else wMod = wCTF = 0.0; }
                if (std::fabs(wCTF) < minCTF) { wMod = std::fabs(wCTF); wCTF = (wCTF >= 0) ? 1.0 : -1.0; }
                else wCTF = 1.0 / wCTF;
                if (phaseFlipped) wCTF = std::fabs(wCTF);
            }
            // M3x3_BY_V3x1(freq, A_SL, freq) with freq.z = 0, then DIGFREQ2FFT_IDX_DOUBLE: index = freq * size
            const double rx = (A[0] * fx + A[1] * fy) * V, ry = (A[3] * fx + A[4] * fy) * V, rz = (A[6] * fx + A[7] * fy) * V;
            const int x1 = (int)std::ceil(rx - R->radius), x2 = (int)std::floor(rx + R->radius);
            const int y1 = (int)std::ceil(ry - R->radius), y2 = (int)std::floor(ry + R->radius);
            const int z1 = (int)std::ceil(rz - R->radius), z2 = (int)std::floor(rz + R->radius);
            const std::complex<double> in = pf[(size_t)i * pxh + j];
            for (int iz = z1; iz <= z2; ++iz) {
                const double dz = iz - rz, z2v = dz * dz;
                const int kz = int_wrap(iz, 0, zsize_1), kzn = int_wrap(-kz, 0, zsize_1);
                for (int iy = y1; iy <= y2; ++iy) {
                    const double dy = iy - ry, y2z2 = dy * dy + z2v;
                    if (y2z2 > r2) continue;
                    const int ky = int_wrap(iy, 0, zsize_1), kyn = int_wrap(-ky, 0, zsize_1);
                    for (int ix = x1; ix <= x2; ++ix) {
                        const double dx = ix - rx, d2 = dx * dx + y2z2;
                        if (d2 > r2) continue;
                        const int aux = (int)(d2 * R->iDeltaSqrt + 0.5);
                        const double w = R->table[aux] * weight * wMod;
                        const int kx = int_wrap(ix, 0, zsize_1);
                        bool conj = false;
                        int pz = kz, py = ky, px = kx;
                        if (kx > xsize_1) { pz = kzn; py = kyn; px = int_wrap(-kx, 0, zsize_1); conj = true; }
                        const size_t o = R->at(pz, py, px);
                        if (reprocess) R->W[o] += w * R->F[o].real();       // RF:770-775: F holds the current 1/w estimate
                        else {
                            const double we = w * wCTF;
                            R->F[o] += std::complex<double>(we * in.real(), conj ? -we * in.imag() : we * in.imag());
                            R->W[o] += w;
                        }
                    }
                }
            }
        }
}

static void force_weight_symmetry(xo_rf2 *R)
{
    // RF:1186-1221
    const int V = R->V;
    int yHalf = V / 2; if (V % 2 == 0) yHalf--;
    int zHalf = V / 2; if (V % 2 == 0) zHalf--;
    for (int k = 0; k < V; ++k) {
        const int ks = int_wrap(-k, 0, V - 1);
        for (int i = 1; i <= yHalf; ++i) {
            const int is = int_wrap(-i, 0, V - 1);
            const double m = 0.5 * (R->W[R->at(k, i, 0)] + R->W[R->at(ks, is, 0)]);
            R->W[R->at(k, i, 0)] = R->W[R->at(ks, is, 0)] = m;
        }
    }
    for (int k = 1; k <= zHalf; ++k) {
        const int ks = int_wrap(-k, 0, V - 1);
        const double m = 0.5 * (R->W[R->at(k, 0, 0)] + R->W[R->at(ks, 0, 0)]);
        R->W[R->at(k, 0, 0)] = R->W[R->at(ks, 0, 0)] = m;
    }
}

// correctWeight RF:1056-1101, split so that the caller can replay the images (reprocess = 1) between the steps:
//   begin;  niter - 1 times { iter_begin; insert every image with reprocess = 1; iter_end };  end
void xo_rf2_weights_begin(xo_rf2 *R)
{
    force_weight_symmetry(R);
    if (R->niter == 0) { for (double &w : R->W) w = 1; return; }
    R->Fsave = R->F;
    force_weight_symmetry(R);
    for (size_t n = 0; n < R->W.size(); ++n)
        if (std::fabs(R->W[n]) > 1e-3) R->F[n] = std::complex<double>(1.0 / R->W[n], R->F[n].imag());
}
void xo_rf2_weights_iter_begin(xo_rf2 *R) { std::fill(R->W.begin(), R->W.end(), 0.0); }
void xo_rf2_weights_iter_end(xo_rf2 *R)
{
    force_weight_symmetry(R);
    for (size_t n = 0; n < R->W.size(); ++n)
        if (std::fabs(R->W[n]) > 1e-3) R->F[n] = std::complex<double>(R->F[n].real() / R->W[n], R->F[n].imag());
}
void xo_rf2_weights_end(xo_rf2 *R)
{
    if (R->niter == 0) return;
    for (size_t n = 0; n < R->W.size(); ++n) R->W[n] = R->F[n].real();    // where w <= 1e-3 this is Re(V): kept (RF:1073-1098)
    R->F = R->Fsave;
    R->Fsave.clear(); R->Fsave.shrink_to_fit();
}

// finishComputations RF:1103-1178 -> D^3 doubles
void xo_rf2_finish(xo_rf2 *R, double *vol)
{
    const int V = R->V, D = R->D;
    // FourierTransformer::enforceHermitianSymmetry (xmippCore xmipp_fftw.cpp, 3-D case): the x = 0 plane is averaged
    // with its conjugated point mirror
    {
        int yHalf = V / 2; if (V % 2 == 0) yHalf--;
        int zHalf = V / 2; if (V % 2 == 0) zHalf--;
        for (int k = 0; k < V; ++k) {
            const int ks = int_wrap(-k, 0, V - 1);
            for (int i = 1; i <= yHalf; ++i) {
                const int is = int_wrap(-i, 0, V - 1);
                const std::complex<double> m = 0.5 * (R->F[R->at(k, i, 0)] + std::conj(R->F[R->at(ks, is, 0)]));
                R->F[R->at(k, i, 0)] = m; R->F[R->at(ks, is, 0)] = std::conj(m);
            }
        }
        for (int k = 1; k <= zHalf; ++k) {
            const int ks = int_wrap(-k, 0, V - 1);
            const std::complex<double> m = 0.5 * (R->F[R->at(k, 0, 0)] + std::conj(R->F[R->at(ks, 0, 0)]));
            R->F[R->at(k, 0, 0)] = m; R->F[R->at(ks, 0, 0)] = std::conj(m);
        }
    }
    // PROCESS_WEIGHTS RF:451-480
    const double corr2D_3D = std::pow(R->pad_proj, 2.) / (D * std::pow(R->pad_vol, 3.));
    for (size_t n = 0; n < R->F.size(); ++n) {
        if (R->niter == 0) R->F[n] *= corr2D_3D;
        else {
            const double w = R->W[n];
            if (1.0 / w > XO_EQUAL_ACCURACY) R->F[n] *= corr2D_3D * w;
            else R->F[n] = 0;
        }
    }
    std::vector<double> Vout((size_t)V * V * V);
    xo_fft3d_c2r(reinterpret_cast<const double *>(R->F.data()), V, V, V, Vout.data());
    const int s0 = xo::first_xmipp_index(D);
    double pr = R->pad_proj / R->pad_vol; pr = pr * pr * pr;
    const double ipr = 1.0 / pr;
    double mean2 = 0;
    const int pc = V / 2;
    for (int k = 0; k < D; ++k)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                const int lk = k + s0, li = i + s0, lj = j + s0;
                const int rk = (lk + pc + V / 2) % V, ri = (li + pc + V / 2) % V, rj = (lj + pc + V / 2) % V;   // CenterFFT(false) + window
                double val = Vout[((size_t)rk * V + ri) * V + rj];
                const double radius = std::sqrt((double)(lk * lk + li * li + lj * lj));
                const double factor = R->ftable[(int)std::floor(radius * R->iDeltaFourier + 0.5)];
                const double xs = radius / (2 * D);
                const double sinc = (xs == 0) ? 1.0 : std::sin(PI2 * xs) / (PI2 * xs);
                const double factor2 = sinc * sinc;
                if (R->niter != 0) { val /= (ipr * factor2 * factor); mean2 += factor2; }
                else val /= (ipr * factor);
                vol[((size_t)k * D + i) * D + j] = val;
            }
    if (R->niter != 0) {
        mean2 /= (double)D * D * D;
        for (size_t n = 0; n < (size_t)D * D * D; ++n) vol[n] *= mean2;
    }
}

}  // extern "C"
