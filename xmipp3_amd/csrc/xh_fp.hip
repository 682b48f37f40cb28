// xh_fp.hip -- FourierProjector on the device (gfx950): central-slice projections of a volume, the
// producer of the reference gallery that projection matching consumes.
// Reference: libraries/data/fourier_projection.cpp -- produceSideInfo (L247-303), produceSideInfoProjection
// (L305-330), project (L91-245); cubic B-spline interpolation only ("bspline", the program's default).
//
//   create : pad the volume about the Xmipp origin, 3-D forward FFT (fp64, xh_plan.h line transforms),
//            ShiftFFT + CenterFFT + the P^3/D^2 scale fused into the split into Re / Im volumes, 3-D
//            B-spline prefilter of both, crop to the frequencies a projection can reach
//   project: thread per Fourier pixel of the D x (D/2+1) half spectrum: rotate the frequency, 4x4x4
//            B-spline taps of Re and Im (mirror at the crop boundary), phase shift (and optional CTF),
//            then the c2r inverse 2-D FFT (columns, Hermitian-extended rows) -> float image
// All arithmetic in fp64 like the reference; the images leave as float (what the gallery stack stores).
#include "xh_common.h"
#include "xh_fft.h"
#include "xh_plan.h"
#include "xh_bspline.h"
#include <cmath>
#include <vector>

namespace {
const double kPI = 3.14159265358979323846;

__global__ void k_fp_pad(const float *__restrict__ vol, xh_cd *__restrict__ F, int D, int P, int o)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)P * P * P;
    if (idx >= total) return;
    const int j = idx % P, i = (idx / P) % P, k = idx / ((size_t)P * P);
    const int jj = j - o, ii = i - o, kk = k - o;
    double v = 0;
    if (jj >= 0 && jj < D && ii >= 0 && ii < D && kk >= 0 && kk < D) v = (double)vol[((size_t)kk * D + ii) * D + jj];
    F[idx] = xh_cd{v, 0.};
}

// F (raw FFT order, un-normalised) -> Re / Im volumes: completeFourierTransform's 1/P^3, ShiftFFT by
// FIRST_XMIPP_INDEX(P) per axis, CenterFFT(.,true), K = P^3/D^2 (L257-266)
__global__ void k_fp_center_split(const xh_cd *__restrict__ F, double *__restrict__ re, double *__restrict__ im, int P, int D)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)P * P * P;
    if (idx >= total) return;
    const int j = idx % P, i = (idx / P) % P, k = idx / ((size_t)P * P);
    // exp(-2 pi i s (j+i+k)/P), s = -(P/2): reduce the integer phase exactly
    const long long m = ((long long)(P / 2) * (j + i + k)) % P;
    double sn, cs;
    sincos(2.0 * 3.14159265358979323846 * (double)m / (double)P, &sn, &cs);
    if (2 * m == P) { cs = -1.0; sn = 0.0; }
    if (m == 0) { cs = 1.0; sn = 0.0; }
    const xh_cd v = F[idx];
    const double scale = 1.0 / ((double)D * (double)D);        // (1/P^3) * (P^3/D^2)
    const double vr = (v.x * cs - v.y * sn) * scale, vi = (v.x * sn + v.y * cs) * scale;
    const int h = P / 2;
    const size_t d = ((size_t)((k + h) % P) * P + (i + h) % P) * P + (j + h) % P;
    re[d] = vr;
    im[d] = vi;
}

__global__ void k_fp_prefilter_z(double *__restrict__ data, int P)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P * P) return;
    d_prefilter_line<double>(data + gid, P, P * P);
}

__global__ void k_fp_crop(const double *__restrict__ in, double *__restrict__ out, int P, int c, int po)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)c * c * c;
    if (idx >= total) return;
    const int j = idx % c, i = (idx / c) % c, k = idx / ((size_t)c * c);
    out[idx] = in[((size_t)(k + po) * P + (i + po)) * P + (j + po)];
}

// project (L91-245), cubic B-spline branch; one thread per (projection, i, j) of the half spectrum
__global__ void __launch_bounds__(256)
k_fp_slice(const double *__restrict__ R, const double *__restrict__ I, const double *__restrict__ eul, const double *__restrict__ ctf,
           xh_cd *__restrict__ pf, int n, int D, int P, int c, int st, double maxFreq2)
{
    const int xh = D / 2 + 1;
    const size_t per = (size_t)D * xh;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * n) return;
    const int p = idx / per;
    const int rem = idx - (size_t)p * per;
    const int i = rem / xh, j = rem - i * xh;
    const double *E = eul + 9 * p;
    const double freqy = (double)(i <= D / 2 ? i : i - D) / (double)D;       // FFT_IDX2DIGFREQ
    const double freqx = (double)j / (double)D;                               // j <= D/2
    xh_cd out = xh_cd{0., 0.};
    if (!((freqy * freqy + freqx * freqx) > maxFreq2)) {
        const double fX = E[3] * freqy + E[0] * freqx, fY = E[4] * freqy + E[1] * freqx, fZ = E[5] * freqy + E[2] * freqx;
        const double z = fZ * P - st, y = fY * P - st, x = fX * P - st;      // logical to physical
        const int l1 = (int)ceil(x - 2), m1 = (int)ceil(y - 2), n1 = (int)ceil(z - 2);
        double wx[4], wy[4], wz[4];
        int ex[4], ey[4], ez[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int l = l1 + t, m = m1 + t, nn = n1 + t;
            wx[t] = d_bspline03<double>(x - (double)l);
            wy[t] = d_bspline03<double>(y - (double)m);
            wz[t] = d_bspline03<double>(z - (double)nn);
            ex[t] = l < 0 ? -l - 1 : (l >= c ? 2 * c - l - 1 : l);
            ey[t] = m < 0 ? -m - 1 : (m >= c ? 2 * c - m - 1 : m);
            ez[t] = nn < 0 ? -nn - 1 : (nn >= c ? 2 * c - nn - 1 : nn);
        }
        double cc = 0, dd = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double yxRe = 0, yxIm = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const size_t row = ((size_t)ez[a] * c + ey[b]) * c;
                double xRe = 0, xIm = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    xRe += R[row + ex[t]] * wx[t];
                    xIm += I[row + ex[t]] * wx[t];
                }
                yxRe += xRe * wy[b];
                yxIm += xIm * wy[b];
            }
            cc += yxRe * wz[a];
            dd += yxIm * wz[a];
        }
        // phase shift that moves the image origin to the corner (L313-329), optional CTF (L231-236)
        const double xxshift = -2.0 * 3.14159265358979323846 * (double)(D / 2) / (double)D;
        const double dotp = (double)j * xxshift + (double)i * xxshift;
        double a = cos(dotp), b = sin(dotp);
        if (ctf) { const double cv = ctf[rem]; a *= cv; b *= cv; }
        const double ac = a * cc, bd = b * dd, ab_cd = (a + b) * (cc + dd);
        out = xh_cd{ac - bd, ab_cd - ac - bd};
    }
    pf[idx] = out;
}

// last pass of the c2r inverse: Hermitian-extend a row of the half spectrum, inverse FFT, real part -> float
__global__ void __launch_bounds__(256)
k_fp_c2r_rows(const xh_cd *__restrict__ pf, XhPlan<double> plan, float *__restrict__ out, int D, size_t nrows, int lpb)
{
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cd *s = reinterpret_cast<xh_cd *>(smem);
    const int M = 1 << plan.logM, xh = D / 2 + 1;
    const int tid = threadIdx.x, nth = blockDim.x;
    const size_t row0 = (size_t)blockIdx.x * lpb;
    const int nl = (int)min((size_t)lpb, nrows - row0);
    for (int i = tid; i < lpb * D; i += nth) {
        const int l = i / D, e = i - l * D;
        xh_cd v = xh_cd{0., 0.};
        if (l < nl) {
            const xh_cd *row = pf + (row0 + l) * xh;
            if (e < xh) { v = row[e]; if (e == 0 || 2 * e == D) v.y = 0; }
            else { v = row[D - e]; v.y = -v.y; }
        }
        s[l * M + xh_plan_pos(plan, e)] = v;
    }
    __syncthreads();
    xh_plan_exec<double, true>(s, plan, lpb, tid, nth);
    for (int i = tid; i < nl * D; i += nth) {
        const int l = i / D, e = i - l * D;
        out[(row0 + l) * D + e] = (float)s[l * M + e].x;
    }
}

void h_euler(double rot, double tilt, double psi, double *A)
{
    // Euler_angles2matrix, closed form (function_tests/test_geometry_main.cpp:46-65)
    const double a = rot * kPI / 180., b = tilt * kPI / 180., g = psi * kPI / 180.;
    const double ca = cos(a), cb = cos(b), cg = cos(g), sa = sin(a), sb = sin(b), sg = sin(g);
    const double cc = cb * ca, cs = cb * sa, sc = sb * ca, ss = sb * sa;
    A[0] = cg * cc - sg * sa; A[1] = cg * cs + sg * ca; A[2] = -cg * sb;
    A[3] = -sg * cc - cg * sa; A[4] = -sg * cs + cg * ca; A[5] = sg * sb;
    A[6] = sc; A[7] = ss; A[8] = cb;
}
}  // namespace

struct xh_fp {
    xh_ctx *ctx;
    int D, P, cdim, cstart;
    double maxFreq;
    XhBuf d_re, d_im;            // cropped B-spline coefficient cubes [cdim]^3 double
    XhBuf d_pf, d_eul;           // per-call scratch
    XhPlanBufs<double> planD;
};

extern "C" {

int xh_fp_create(xh_ctx *ctx, const float *d_vol, int32_t D, double padding, double max_freq, int32_t degree, xh_fp **out)
{
    XH_CHECK(ctx && d_vol && out && D >= 4, XH_ERR_ARG, "xh_fp_create: bad argument");
    XH_CHECK(degree == 3, XH_ERR_UNSUPPORTED, "xh_fp_create: only cubic B-spline interpolation (degree 3) is on the device, got %d", degree);
    XH_CHECK(padding >= 1.0 && max_freq > 0 && max_freq <= 0.5, XH_ERR_ARG, "xh_fp_create: padding %g / max_freq %g out of range", padding, max_freq);
    const int P = (int)(padding * D);
    XH_CHECK(P <= 1024, XH_ERR_UNSUPPORTED, "xh_fp_create: padded size %d exceeds 1024", P);
    XH_HIP(hipSetDevice(ctx->device));
    xh_fp *fp = new xh_fp;
    fp->ctx = ctx; fp->D = D; fp->P = P; fp->maxFreq = max_freq;
    int idxMax = (int)(max_freq * P + 10);                    // L281: +10 is a safety guard
    const int lastP = -(P / 2) + P - 1, firstP = -(P / 2);
    idxMax = std::min(lastP, idxMax);
    const int idxMin = std::max(-idxMax, firstP);
    fp->cdim = idxMax - idxMin + 1;
    fp->cstart = idxMin;
    const size_t P3 = (size_t)P * P * P, c3 = (size_t)fp->cdim * fp->cdim * fp->cdim;
    XhBuf F, re, im;
    XhPlanBufs<double> planP;
    int rc = xh_buf_alloc(ctx, F, sizeof(xh_cd) * P3);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, re, sizeof(double) * P3);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, im, sizeof(double) * P3);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, P, planP);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, D, fp->planD);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, fp->d_re, sizeof(double) * c3);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, fp->d_im, sizeof(double) * c3);
    hipError_t e = hipSuccess;
    if (rc == XH_OK) {
        const unsigned gb = (unsigned)((P3 + 255) / 256);
        hipLaunchKernelGGL(k_fp_pad, dim3(gb), dim3(256), 0, ctx->stream, d_vol, (xh_cd *)F.p, D, P, -(D / 2) + P / 2);
        const int lpb = xh_plan_lpb(planP.plan, 64 * 1024, 8);
        const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << planP.plan.logM;
        const size_t nlines = (size_t)P * P;
        const unsigned gl = (unsigned)((nlines + lpb - 1) / lpb);
        // x lines: (k,i) -> offset (k*P+i)*P, element stride 1
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3(gl), dim3(256), smem, ctx->stream, (xh_cd *)F.p, planP.plan, nlines,
                           (size_t)1, (size_t)P, (size_t)0, (size_t)1, lpb);
        // y lines: (k,j) -> offset k*P*P + j, element stride P
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3(gl), dim3(256), smem, ctx->stream, (xh_cd *)F.p, planP.plan, nlines,
                           (size_t)P, (size_t)P * P, (size_t)1, (size_t)P, lpb);
        // z lines: (i,j) -> offset i*P + j, element stride P*P
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3(gl), dim3(256), smem, ctx->stream, (xh_cd *)F.p, planP.plan, nlines,
                           nlines, (size_t)0, (size_t)1, (size_t)P * P, lpb);
        hipLaunchKernelGGL(k_fp_center_split, dim3(gb), dim3(256), 0, ctx->stream, (const xh_cd *)F.p, (double *)re.p, (double *)im.p, P, D);
        // produceSplineCoefficients on both volumes: x (LDS row tiles), y (thread per column), z
        const int TR = std::max(1, std::min(32, (int)(60000 / ((P + 1) * sizeof(double)))));
        const int tiles = (P + TR - 1) / TR;
        for (XhBuf *b : {&re, &im}) {
            hipLaunchKernelGGL((k_pm_prefilter_rows<double, double>), dim3(P * tiles), dim3(64), sizeof(double) * TR * (P + 1), ctx->stream,
                               (const double *)b->p, (const int *)nullptr, (double *)b->p, P, TR, (const int *)nullptr);
            hipLaunchKernelGGL((k_pm_prefilter_cols<double>), dim3((P * P + 63) / 64), dim3(64), 0, ctx->stream, (double *)b->p, P, P,
                               (const int *)nullptr);
            hipLaunchKernelGGL(k_fp_prefilter_z, dim3((P * P + 63) / 64), dim3(64), 0, ctx->stream, (double *)b->p, P);
        }
        const unsigned gc = (unsigned)((c3 + 255) / 256);
        hipLaunchKernelGGL(k_fp_crop, dim3(gc), dim3(256), 0, ctx->stream, (const double *)re.p, (double *)fp->d_re.p, P, fp->cdim, idxMin - firstP);
        hipLaunchKernelGGL(k_fp_crop, dim3(gc), dim3(256), 0, ctx->stream, (const double *)im.p, (double *)fp->d_im.p, P, fp->cdim, idxMin - firstP);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    xh_buf_free(F); xh_buf_free(re); xh_buf_free(im);
    xh_plan_free(planP);
    if (rc == XH_OK && e != hipSuccess) { xh_set_error("xh_fp_create: %s", hipGetErrorString(e)); rc = XH_ERR_HIP; }
    if (rc != XH_OK) {
        xh_buf_free(fp->d_re); xh_buf_free(fp->d_im); xh_plan_free(fp->planD);
        delete fp;
        return rc;
    }
    *out = fp;
    return XH_OK;
}

int xh_fp_destroy(xh_fp *fp)
{
    if (!fp) return XH_OK;
    (void)hipSetDevice(fp->ctx->device);
    (void)hipStreamSynchronize(fp->ctx->stream);
    xh_buf_free(fp->d_re); xh_buf_free(fp->d_im); xh_buf_free(fp->d_pf); xh_buf_free(fp->d_eul);
    xh_plan_free(fp->planD);
    delete fp;
    return XH_OK;
}

int xh_fp_info(const xh_fp *fp, int32_t *P, int32_t *cdim, int32_t *cstart)
{
    XH_CHECK(fp, XH_ERR_ARG, "xh_fp_info: null handle");
    if (P) *P = fp->P;
    if (cdim) *cdim = fp->cdim;
    if (cstart) *cstart = fp->cstart;
    return XH_OK;
}

int xh_fp_coefs(const xh_fp *fp, double *h_re, double *h_im)
{
    XH_CHECK(fp && h_re && h_im, XH_ERR_ARG, "xh_fp_coefs: bad argument");
    XH_HIP(hipSetDevice(fp->ctx->device));
    XH_HIP(hipMemcpyAsync(h_re, fp->d_re.p, fp->d_re.bytes, hipMemcpyDeviceToHost, fp->ctx->stream));
    XH_HIP(hipMemcpyAsync(h_im, fp->d_im.p, fp->d_im.bytes, hipMemcpyDeviceToHost, fp->ctx->stream));
    XH_HIP(hipStreamSynchronize(fp->ctx->stream));
    return XH_OK;
}

int xh_fp_project(xh_fp *fp, const double *h_angles, int32_t n, const double *d_ctf, float *d_out)
{
    XH_CHECK(fp && h_angles && d_out && n >= 0, XH_ERR_ARG, "xh_fp_project: bad argument");
    XH_HIP(hipSetDevice(fp->ctx->device));
    if (n == 0) return XH_OK;
    xh_ctx *ctx = fp->ctx;
    const int D = fp->D, xh = D / 2 + 1;
    const size_t per = (size_t)D * xh;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(n, ((size_t)512 << 20) / (per * sizeof(xh_cd))));
    XH_TRY(xh_buf_reserve(ctx, fp->d_pf, sizeof(xh_cd) * per * chunk));
    XH_TRY(xh_buf_reserve(ctx, fp->d_eul, sizeof(double) * 9 * chunk));
    const XhPlan<double> &plan = fp->planD.plan;
    const int lpb = xh_plan_lpb(plan, 64 * 1024, 16);
    const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << plan.logM;
    std::vector<double> E(9 * (size_t)chunk);
    for (int p0 = 0; p0 < n; p0 += chunk) {
        const int m = std::min(chunk, n - p0);
        for (int p = 0; p < m; ++p) h_euler(h_angles[3 * (p0 + p)], h_angles[3 * (p0 + p) + 1], h_angles[3 * (p0 + p) + 2], &E[9 * (size_t)p]);
        XH_HIP(hipMemcpyAsync(fp->d_eul.p, E.data(), sizeof(double) * 9 * m, hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_fp_slice, dim3((unsigned)((per * m + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)fp->d_re.p,
                           (const double *)fp->d_im.p, (const double *)fp->d_eul.p, d_ctf, (xh_cd *)fp->d_pf.p, m, D, fp->P, fp->cdim,
                           fp->cstart, fp->maxFreq * fp->maxFreq);
        XH_LAUNCH_CHECK();
        // inverse along y: lines (p, j): offset p*per + j, element stride xh
        const size_t ncol = (size_t)m * xh;
        hipLaunchKernelGGL((xh_k_fft_lines<double, true>), dim3((unsigned)((ncol + lpb - 1) / lpb)), dim3(256), smem, ctx->stream,
                           (xh_cd *)fp->d_pf.p, plan, ncol, (size_t)xh, per, (size_t)1, (size_t)xh, lpb);
        XH_LAUNCH_CHECK();
        const size_t nrows = (size_t)m * D;
        hipLaunchKernelGGL(k_fp_c2r_rows, dim3((unsigned)((nrows + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, (const xh_cd *)fp->d_pf.p,
                           plan, d_out + (size_t)p0 * D * D, D, nrows, lpb);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

}  // extern "C"
