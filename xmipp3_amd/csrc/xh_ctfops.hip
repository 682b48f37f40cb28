// xh_ctfops.hip -- the CTF pre-steps that sit in front of the matching / reconstruction loop:
//
//   xh_ctfop_phase_flip   actualPhaseFlip, reconstruction/ctf_phase_flip.cpp:88-117 (xmipp_ctf_phase_flip): transform,
//                         negate every coefficient where the undamped CTF is negative, transform back
//   xh_ctfop_wiener2d     Wiener2D::wienerFilter + applyWienerFilter, data/wiener2d.cpp:29-141 (xmipp_ctf_correct_wiener2d):
//                         pad about the Xmipp origin, transform, multiply by CTF / (CTF^2 + wc), transform back, crop
//
// The reference works on the half spectrum of a real transform (FFTW r2c / c2r). Here the image rides as the real part of a
// complex transform (xh_fft2d: whole micrographs fit, any size its line plans take) and the real filter is applied to the
// full spectrum so that the result is what the half-spectrum form gives: coefficient (i, j) with j beyond nx/2 is the
// conjugate of (-i, -j), which the reference multiplies by the filter value at ((ny - i) % ny, nx - j) -- evaluated at those
// indices here, because FFT_IDX2DIGFREQ maps the Nyquist index to +0.5 on either side. The CTF itself is evaluated in double
// precision with the reference's formulas (data/ctf.h:452-500,541-570,1002-1029); the transforms run in fp32 (reference:
// double), which is the tolerance the tests state (1e-5 of the image's largest value).
#include <cmath>
#include <cstring>

#include "xh_common.h"

namespace {
constexpr double kPI = 3.14159265358979323846;
typedef float2 xc_cf;

struct CtfSide {
    double K1, K2, K3, K5, K6, K7, Ksin, Kcos, rad_azimuth, defocus_average, defocus_deviation;
    double DeltaR, K, envR0, envR1, envR2, phase_shift, VPP_radius;
};

// produceSideInfo, data/ctf.cpp:645-679,1392-1402; phase_shift arrives in degrees (ctf_phase_flip.cpp:99, wiener2d.cpp:149)
CtfSide side_info(const xh_ctf_params &c)
{
    CtfSide d;
    const double local_Cs = c.Cs * 1e7, local_Ca = c.Ca * 1e7, local_kV = c.kV * 1e3, local_ispr = c.ispr * 1e6;
    const double lambda = 12.2643247 / std::sqrt(local_kV * (1. + 0.978466e-6 * local_kV));
    d.K1 = kPI * lambda;
    d.K2 = kPI / 2 * local_Cs * lambda * lambda * lambda;
    d.K3 = std::pow(0.25 * kPI * local_Ca * lambda * (c.espr / c.kV + 2 * local_ispr), 2) / std::log(2.0);
    d.K5 = kPI * c.DeltaF * lambda;
    d.K6 = kPI * kPI * c.alpha * c.alpha;
    d.K7 = local_Cs * lambda * lambda;
    d.Ksin = std::sqrt(1 - c.Q0 * c.Q0);
    d.Kcos = c.Q0;
    d.rad_azimuth = c.azimuthal_angle * kPI / 180.;
    d.defocus_average = -(c.DeltafU + c.DeltafV) * 0.5;
    d.defocus_deviation = -(c.DeltafU - c.DeltafV) * 0.5;
    d.DeltaR = c.DeltaR; d.K = c.K; d.envR0 = c.envR0; d.envR1 = c.envR1; d.envR2 = c.envR2;
    d.phase_shift = (c.phase_shift * kPI) / 180;
    d.VPP_radius = c.VPP_radius;
    return d;
}

__device__ __forceinline__ double d_bessj0(double x)
{
    const double ax = fabs(x);
    if (ax < 8.0) {
        const double y = x * x;
        const double a1 = 57568490574.0 + y * (-13362590354.0 + y * (651619640.7 + y * (-11214424.18 + y * (77392.33017 + y * (-184.9052456)))));
        const double a2 = 57568490411.0 + y * (1029532985.0 + y * (9494680.718 + y * (59272.64853 + y * (267.8532712 + y * 1.0))));
        return a1 / a2;
    }
    const double z = 8.0 / ax, y = z * z, xx = ax - 0.785398164;
    const double a1 = 1.0 + y * (-0.1098628627e-2 + y * (0.2734510407e-4 + y * (-0.2073370639e-5 + y * 0.2093887211e-6)));
    const double a2 = -0.1562499995e-1 + y * (0.1430488765e-3 + y * (-0.6911147651e-5 + y * (0.7621095161e-6 - y * 0.934935152e-7)));
    return sqrt(0.636619772 / ax) * (cos(xx) * a1 - z * sin(xx) * a2);
}

// getValuePureAt (damping) / getValuePureWithoutDampingAt after precomputeValues(X, Y)
__device__ __forceinline__ double d_ctf_at(const CtfSide &s, double X, double Y, bool damping)
{
    const double u2 = X * X + Y * Y, u = sqrt(u2), u4 = u2 * u2;
    double deltaf;
    if (fabs(X) < 1e-6 && fabs(Y) < 1e-6) deltaf = 0;
    else deltaf = s.defocus_average + s.defocus_deviation * cos(2 * (atan2(Y, X) - s.rad_azimuth));
    double VPP = 0;
    if (round(s.VPP_radius * 1000) != 0) VPP = -s.phase_shift * (1 - exp(-u2 / (2 * s.VPP_radius * s.VPP_radius)));
    const double argument = VPP + s.K1 * deltaf * u2 + s.K2 * u4;
    double sine_part, cosine_part;
    sincos(argument, &sine_part, &cosine_part);
    if (!damping) return -(s.Ksin * sine_part - s.Kcos * cosine_part);
    const double Eespr = exp(-s.K3 * u4);
    const double EdeltaF = d_bessj0(s.K5 * u2);
    const double xs = u * s.DeltaR;
    const double EdeltaR = (xs == 0) ? 1.0 : sin(kPI * xs) / (kPI * xs);
    const double aux = s.K7 * u2 * u + deltaf * u;
    const double Ealpha = exp(-s.K6 * aux * aux);
    double E = Eespr * EdeltaF * EdeltaR * Ealpha + s.envR0 + s.envR1 * u + s.envR2 * u2;
    if (E < 0) E = 0;
    return -s.K * (s.Ksin * sine_part - s.Kcos * cosine_part) * E;
}

// FFT_IDX2DIGFREQ (xmippCore xmipp_fft.h; in-tree copy cuda_gpu_reconstruct_fourier.cpp:381-385)
__device__ __forceinline__ double d_digfreq(int idx, int size) { return size <= 1 ? 0.0 : (double)(idx <= size / 2 ? idx : idx - size) / (double)size; }

// the index pair whose filter value the half-spectrum form applies to coefficient (i, j)
__device__ __forceinline__ void d_half_index(int &i, int &j, int ny, int nx)
{
    if (j > nx / 2) { j = nx - j; i = (ny - i) % ny; }
}

__global__ void __launch_bounds__(256) k_ctf_embed(const float *__restrict__ img, int ydim, int xdim, xc_cf *__restrict__ out, int pY, int pX, int oy, int ox)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)pY * pX) return;
    const int i = (int)(t / pX), j = (int)(t - (size_t)i * pX);
    const int si = i - oy, sj = j - ox;
    const bool in = si >= 0 && si < ydim && sj >= 0 && sj < xdim;
    out[t] = xc_cf{in ? img[(size_t)si * xdim + sj] : 0.f, 0.f};
}

__global__ void __launch_bounds__(256) k_ctf_extract(const xc_cf *__restrict__ in, int pY, int pX, int oy, int ox, float *__restrict__ img, int ydim, int xdim)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)ydim * xdim) return;
    const int i = (int)(t / xdim), j = (int)(t - (size_t)i * xdim);
    img[t] = in[(size_t)(i + oy) * pX + (j + ox)].x;
}

__global__ void __launch_bounds__(256) k_ctf_flip(xc_cf *__restrict__ F, int ny, int nx, CtfSide s, double iTm)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)ny * nx) return;
    int i = (int)(t / nx), j = (int)(t - (size_t)i * nx);
    d_half_index(i, j, ny, nx);
    if (d_ctf_at(s, d_digfreq(j, nx) * iTm, d_digfreq(i, ny) * iTm, false) < 0) { xc_cf v = F[t]; F[t] = xc_cf{-v.x, -v.y}; }
}

// ctfIm (wiener2d.cpp:55-69) over the whole padded array, and the sum of its squares for Grigorieff's default constant
__global__ void __launch_bounds__(256) k_ctf_wiener_ctf(double *__restrict__ ctfIm, int pY, int pX, CtfSide s, double iTs, int damping, int phaseFlipped,
                                                        double *__restrict__ sumSq)
{
    __shared__ double red[256];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    double sq = 0;
    if (t < (size_t)pY * pX) {
        const int i = (int)(t / pX), j = (int)(t - (size_t)i * pX);
        double v = d_ctf_at(s, d_digfreq(j, pX) * iTs, d_digfreq(i, pY) * iTs, damping != 0);
        if (phaseFlipped) v = fabs(v);
        ctfIm[t] = v;
        sq = v * v;
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(sumSq, red[0]);
}

// F(i, j) *= ctf / (ctf^2 + wc) at the half-spectrum index of (i, j); wc < 0: 0.1 * mean(ctf^2) (wiener2d.cpp:78-98)
__global__ void __launch_bounds__(256) k_ctf_wiener_apply(xc_cf *__restrict__ F, const double *__restrict__ ctfIm, int pY, int pX, double wcIn,
                                                          const double *__restrict__ sumSq)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)pY * pX) return;
    int i = (int)(t / pX), j = (int)(t - (size_t)i * pX);
    d_half_index(i, j, pY, pX);
    const double wc = wcIn < 0. ? 0.1 * (*sumSq / ((double)pY * (double)pX)) : wcIn;
    const double c = ctfIm[(size_t)i * pX + j];
    const double m = c / (c * c + wc);
    const xc_cf v = F[t];
    F[t] = xc_cf{(float)(v.x * m), (float)(v.y * m)};
}
}  // namespace

struct xh_ctfop {
    xh_ctx *ctx;
    int ydim, xdim, pY, pX;
    xh_fft2d *fft;
    XhBuf work, ctfIm, sum;
};

extern "C" {

int xh_ctfop_create(xh_ctx *ctx, int32_t ydim, int32_t xdim, double pad, xh_ctfop **out)
{
    XH_CHECK(ctx && out && ydim >= 2 && xdim >= 2, XH_ERR_ARG, "xh_ctfop_create: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    if (!(pad >= 1.)) pad = 1.;                 // XMIPP_MAX(1., pad), ctf_correct_wiener2d.cpp:34
    xh_ctfop *h = new xh_ctfop;
    h->ctx = ctx; h->ydim = ydim; h->xdim = xdim;
    h->pY = (int)(ydim * pad); h->pX = (int)(xdim * pad);        // wiener2d.cpp:31-33
    h->fft = nullptr;
    int rc = xh_fft2d_create(ctx, h->pY, h->pX, &h->fft);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->work, sizeof(xc_cf) * (size_t)h->pY * h->pX);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->sum, sizeof(double));
    if (rc != XH_OK) { xh_ctfop_destroy(h); return rc; }
    *out = h;
    return XH_OK;
}

int xh_ctfop_destroy(xh_ctfop *h)
{
    if (!h) return XH_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->fft) xh_fft2d_destroy(h->fft);
    xh_buf_free(h->work); xh_buf_free(h->ctfIm); xh_buf_free(h->sum);
    delete h;
    return XH_OK;
}

int xh_ctfop_phase_flip(xh_ctfop *h, float *d_img, const xh_ctf_params *ctf, double sampling_rate)
{
    XH_CHECK(h && d_img && ctf && sampling_rate > 0, XH_ERR_ARG, "xh_ctfop_phase_flip: bad argument");
    XH_CHECK(h->pY == h->ydim && h->pX == h->xdim, XH_ERR_STATE, "xh_ctfop_phase_flip: the handle was created with padding; phase flipping works on the image as it is");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)h->ydim * h->xdim;
    const unsigned nb = (unsigned)((tot + 255) / 256);
    xc_cf *w = (xc_cf *)h->work.p;
    hipLaunchKernelGGL(k_ctf_embed, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_img, h->ydim, h->xdim, w, h->ydim, h->xdim, 0, 0);
    XH_LAUNCH_CHECK();
    XH_TRY(xh_fft2d_exec(h->fft, (float *)w, 0));
    hipLaunchKernelGGL(k_ctf_flip, dim3(nb), dim3(256), 0, ctx->stream, w, h->ydim, h->xdim, side_info(*ctf), 1.0 / sampling_rate);
    XH_LAUNCH_CHECK();
    XH_TRY(xh_fft2d_exec(h->fft, (float *)w, 1));
    hipLaunchKernelGGL(k_ctf_extract, dim3(nb), dim3(256), 0, ctx->stream, (const xc_cf *)w, h->ydim, h->xdim, 0, 0, d_img, h->ydim, h->xdim);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

int xh_ctfop_wiener2d(xh_ctfop *h, float *d_imgs, int32_t n, const xh_ctf_params *ctfs, double sampling_rate, int32_t phase_flipped,
                      int32_t is_isotropic, double wiener_constant, int32_t correct_envelope)
{
    XH_CHECK(h && d_imgs && ctfs && n >= 0 && sampling_rate > 0, XH_ERR_ARG, "xh_ctfop_wiener2d: bad argument");
    (void)is_isotropic;      // the reference averages DeltafU/V after produceSideInfo has consumed them (wiener2d.cpp:36-50): no effect
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t ptot = (size_t)h->pY * h->pX, itot = (size_t)h->ydim * h->xdim;
    XH_TRY(xh_buf_reserve(ctx, h->ctfIm, sizeof(double) * ptot));
    const unsigned nbp = (unsigned)((ptot + 255) / 256), nbi = (unsigned)((itot + 255) / 256);
    // selfWindow about the Xmipp origin (wiener2d.cpp:111-119): the image keeps its logical coordinates
    const int oy = h->pY / 2 - h->ydim / 2, ox = h->pX / 2 - h->xdim / 2;
    xc_cf *w = (xc_cf *)h->work.p;
    for (int k = 0; k < n; ++k) {
        float *img = d_imgs + (size_t)k * itot;
        const CtfSide s = side_info(ctfs[k]);
        XH_HIP(hipMemsetAsync(h->sum.p, 0, sizeof(double), ctx->stream));
        hipLaunchKernelGGL(k_ctf_wiener_ctf, dim3(nbp), dim3(256), 0, ctx->stream, (double *)h->ctfIm.p, h->pY, h->pX, s, 1.0 / sampling_rate,
                           correct_envelope, phase_flipped, (double *)h->sum.p);
        hipLaunchKernelGGL(k_ctf_embed, dim3(nbp), dim3(256), 0, ctx->stream, (const float *)img, h->ydim, h->xdim, w, h->pY, h->pX, oy, ox);
        XH_LAUNCH_CHECK();
        XH_TRY(xh_fft2d_exec(h->fft, (float *)w, 0));
        hipLaunchKernelGGL(k_ctf_wiener_apply, dim3(nbp), dim3(256), 0, ctx->stream, w, (const double *)h->ctfIm.p, h->pY, h->pX, wiener_constant,
                           (const double *)h->sum.p);
        XH_LAUNCH_CHECK();
        XH_TRY(xh_fft2d_exec(h->fft, (float *)w, 1));
        hipLaunchKernelGGL(k_ctf_extract, dim3(nbi), dim3(256), 0, ctx->stream, (const xc_cf *)w, h->pY, h->pX, oy, ox, img, h->ydim, h->xdim);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

}  // extern "C"
