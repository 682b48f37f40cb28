// oracle/xo_ctfops.cpp -- CPU restatement of the CTF pre-steps. TEST INFRASTRUCTURE ONLY.
//
//   xo_ctf_phase_flip   actualPhaseFlip, reconstruction/ctf_phase_flip.cpp:88-117 (xmipp_ctf_phase_flip)
//   xo_ctf_wiener2d     Wiener2D::wienerFilter / applyWienerFilter, data/wiener2d.cpp:29-141 (xmipp_ctf_correct_wiener2d)
//
// CTF arithmetic: data/ctf.h:452-500,541-570 (getValuePureAt, getValuePureWithoutDampingAt), :1002-1029 (precomputeValues),
// :1216-1268 (generateCTF, generateCTFWithoutDamping), data/ctf.cpp:645-679,1392-1402 (produceSideInfo).
// Pinned on the only known answer the reference holds for it: the phase-flipped delta of test_ctf_main.cpp:126-149
// (tests/test_oracle_pins.py).
#include <cmath>
#include <complex>
#include <vector>

#include "xo.h"
#include "xo_internal.h"

namespace {
const double PI = 3.14159265358979323846;
typedef std::complex<double> cd;

double bessj0(double x)
{
    // xmippCore numerical_recipes bessj0 (in-tree copy: cuda_gpu_reconstruct_fourier.cpp is silent on it; the same
    // polynomial the gridding oracle uses, xo_recfourier.cpp)
    double ax, z, xx, y, ans, ans1, ans2;
    if ((ax = std::fabs(x)) < 8.0) {
        y = x * x;
        ans1 = 57568490574.0 + y * (-13362590354.0 + y * (651619640.7 + y * (-11214424.18 + y * (77392.33017 + y * (-184.9052456)))));
        ans2 = 57568490411.0 + y * (1029532985.0 + y * (9494680.718 + y * (59272.64853 + y * (267.8532712 + y * 1.0))));
        ans = ans1 / ans2;
    } else {
        z = 8.0 / ax;
        y = z * z;
        xx = ax - 0.785398164;
        ans1 = 1.0 + y * (-0.1098628627e-2 + y * (0.2734510407e-4 + y * (-0.2073370639e-5 + y * 0.2093887211e-6)));
        ans2 = -0.1562499995e-1 + y * (0.1430488765e-3 + y * (-0.6911147651e-5 + y * (0.7621095161e-6 - y * 0.934935152e-7)));
        ans = std::sqrt(0.636619772 / ax) * (std::cos(xx) * ans1 - z * std::sin(xx) * ans2);
    }
    return ans;
}

// side information of a CTF description (produceSideInfo) and its evaluation at a continuous frequency
struct Side {
    double K1, K2, K3, K5, K6, K7, Ksin, Kcos, rad_azimuth, defocus_average, defocus_deviation;
    double DeltaR, K, envR0, envR1, envR2, phase_shift, VPP_radius;
};
Side side_info(const xo_ctf_params &p)
{
    Side s;
    const double local_Cs = p.Cs * 1e7, local_Ca = p.Ca * 1e7, local_ispr = p.ispr * 1e6;
    const double lambda = xo_ctf_lambda(&p);
    s.K1 = PI * lambda;
    s.K2 = PI / 2 * local_Cs * lambda * lambda * lambda;
    s.K3 = std::pow(0.25 * PI * local_Ca * lambda * (p.espr / p.kV + 2 * local_ispr), 2) / std::log(2.0);
    s.K5 = PI * p.DeltaF * lambda;
    s.K6 = PI * PI * p.alpha * p.alpha;
    s.K7 = local_Cs * lambda * lambda;
    s.Ksin = std::sqrt(1 - p.Q0 * p.Q0);
    s.Kcos = p.Q0;
    s.rad_azimuth = p.azimuthal_angle * PI / 180.;
    s.defocus_average = -(p.DeltafU + p.DeltafV) * 0.5;
    s.defocus_deviation = -(p.DeltafU - p.DeltafV) * 0.5;
    s.DeltaR = p.DeltaR; s.K = p.K; s.envR0 = p.envR0; s.envR1 = p.envR1; s.envR2 = p.envR2;
    s.phase_shift = p.phase_shift; s.VPP_radius = p.VPP_radius;
    return s;
}
// damping: getValuePureAt (with K and the envelope E), else getValuePureWithoutDampingAt
double ctf_at(const Side &s, double X, double Y, bool damping)
{
    const double ang = std::atan2(Y, X);
    const double u2 = X * X + Y * Y;
    const double u = std::sqrt(u2);
    const double u4 = u2 * u2;
    double deltaf;
    if (std::fabs(X) < XO_EQUAL_ACCURACY && std::fabs(Y) < XO_EQUAL_ACCURACY) deltaf = 0;
    else deltaf = s.defocus_average + s.defocus_deviation * std::cos(2 * (ang - s.rad_azimuth));
    double VPP = 0;
    if (std::round(s.VPP_radius * 1000) != 0) VPP = -s.phase_shift * (1 - std::exp(-u2 / (2 * std::pow(s.VPP_radius, 2.0))));
    const double argument = VPP + s.K1 * deltaf * u2 + s.K2 * u4;
    const double sine_part = std::sin(argument), cosine_part = std::cos(argument);
    if (!damping) return -(s.Ksin * sine_part - s.Kcos * cosine_part);
    const double Eespr = std::exp(-s.K3 * u4);
    const double EdeltaF = bessj0(s.K5 * u2);
    const double xs = u * s.DeltaR;
    const double EdeltaR = (xs == 0) ? 1.0 : std::sin(PI * xs) / (PI * xs);
    const double aux = s.K7 * u2 * u + deltaf * u;
    const double Ealpha = std::exp(-s.K6 * aux * aux);
    double E = Eespr * EdeltaF * EdeltaR * Ealpha + s.envR0 + s.envR1 * u + s.envR2 * u2;
    if (E < 0) E = 0;
    return -s.K * (s.Ksin * sine_part - s.Kcos * cosine_part) * E;
}
}  // namespace

extern "C" {

// ctf->Tm: sampling rate of the image (the program sets it from --sampling or Tm * downsampling, ctf_phase_flip.cpp:75-79);
// ctf->phase_shift in degrees (converted like L99). with_damping = 0: actualPhaseFlip; 1: CTFDescription::correctPhase
// (ctf.cpp:1553-1582), the variant test_ctf_main.cpp pins.
void xo_ctf_phase_flip(double *img, int ydim, int xdim, const xo_ctf_params *ctf, int with_damping)
{
    const int xh = xdim / 2 + 1;
    std::vector<double> F((size_t)ydim * xh * 2);
    xo_fft2d_r2c(img, ydim, xdim, F.data());
    cd *C = reinterpret_cast<cd *>(F.data());
    Side s = side_info(*ctf);
    s.phase_shift = (ctf->phase_shift * PI) / 180;
    const double iTm = 1.0 / ctf->Tm;
    for (int i = 0; i < ydim; ++i) {
        const double fy = xo_fft_idx2digfreq(i, ydim) * iTm;
        for (int j = 0; j < xh; ++j) {
            const double fx = xo_fft_idx2digfreq(j, xdim) * iTm;
            if (ctf_at(s, fx, fy, with_damping != 0) < 0) C[(size_t)i * xh + j] *= -1.0;
        }
    }
    xo_fft2d_c2r(F.data(), ydim, xdim, img);
}

// One image through Wiener2D::applyWienerFilter. The reference calls produceSideInfo BEFORE it overwrites Tm and, for
// --isIsotropic, DeltafU/V (wiener2d.cpp:36-50): the defocus average / deviation the CTF is evaluated with are those of the
// unmodified description, so --isIsotropic changes nothing. Kept (is_isotropic is accepted and has no effect).
void xo_ctf_wiener2d(double *img, int ydim, int xdim, const xo_ctf_params *ctf, double sampling_rate, double pad,
                     int phase_flipped, int is_isotropic, double wiener_constant, int correct_envelope)
{
    (void)is_isotropic;
    if (pad < 1.) pad = 1.;
    const int pY = (int)(ydim * pad), pX = (int)(xdim * pad);
    Side s = side_info(*ctf);
    s.phase_shift = (ctf->phase_shift * PI) / 180;     // wiener2d.cpp:149
    const double iTs = 1.0 / sampling_rate;
    std::vector<double> ctfIm((size_t)pY * pX), Mwien((size_t)pY * pX);
    for (int i = 0; i < pY; ++i) {
        const double fy = xo_fft_idx2digfreq(i, pY) * iTs;
        for (int j = 0; j < pX; ++j) {
            const double fx = xo_fft_idx2digfreq(j, pX) * iTs;
            double v = ctf_at(s, fx, fy, correct_envelope != 0);
            if (phase_flipped) v = std::fabs(v);
            ctfIm[(size_t)i * pX + j] = v;
            Mwien[(size_t)i * pX + j] = v * v;
        }
    }
    double wc = wiener_constant;
    if (wc < 0.) {
        double sum = 0;
        for (double v : Mwien) sum += v;
        wc = 0.1 * (sum / (double)Mwien.size());
    }
    for (size_t k = 0; k < Mwien.size(); ++k) Mwien[k] = ctfIm[k] / (Mwien[k] + wc);
    // pad about the Xmipp origin (selfWindow), transform, filter the half spectrum with the leading columns of Mwien, back
    std::vector<double> P((size_t)pY * pX, 0.0);
    const int oy = xo::first_xmipp_index(ydim) - xo::first_xmipp_index(pY), ox = xo::first_xmipp_index(xdim) - xo::first_xmipp_index(pX);
    for (int i = 0; i < ydim; ++i)
        for (int j = 0; j < xdim; ++j) P[(size_t)(i + oy) * pX + (j + ox)] = img[(size_t)i * xdim + j];
    const int xh = pX / 2 + 1;
    std::vector<double> F((size_t)pY * xh * 2);
    xo_fft2d_r2c(P.data(), pY, pX, F.data());
    cd *C = reinterpret_cast<cd *>(F.data());
    for (int i = 0; i < pY; ++i)
        for (int j = 0; j < xh; ++j) C[(size_t)i * xh + j] *= Mwien[(size_t)i * pX + j];      // dAij(Faux,i,j) *= dAij(Mwien,i,j)
    xo_fft2d_c2r(F.data(), pY, pX, P.data());
    for (int i = 0; i < ydim; ++i)
        for (int j = 0; j < xdim; ++j) img[(size_t)i * xdim + j] = P[(size_t)(i + oy) * pX + (j + ox)];
}

}  // extern "C"
