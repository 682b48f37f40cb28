// xh_fsc.hip -- Fourier ring / shell correlation of two images or volumes on gfx950: the numerical core of
// xmipp_resolution_fsc (reconstruction/resolution_fsc.cpp:179-203), i.e. xmippCore's frc_dpr. SURVEY.md 8f rank 2.
//
//   r2c 3-D FFT (fp64, line transforms of xh_plan.h: any length) of both inputs into [Z][Y][X/2+1]
//   one pass over the half-complex coefficients: shell = round(R * X), seven sums per shell accumulated in an
//     LDS table per workgroup (ds_add_f64), flushed once with global atomics; R-factor sums ride along
//   the host turns the sums into freq / frc / frc_noise / dpr / error_l2 (X/2+1 values each).
// Shell membership follows the scalar double arithmetic of frc_dpr exactly (no FMA contraction: this file is
// built with -ffp-contract=off); the sums differ from a sequential loop only by fp64 summation order.
// HBM bound: 2 volumes read once (8 B/voxel), 2 half-spectra written and re-read by three line passes.
#include "xh_common.h"
#include "xh_plan.h"
#include "../../include/xmipp_hip.h"

namespace {

// x lines: real input [nlines][X] -> half spectrum [nlines][xh], forward, un-normalised
__global__ void __launch_bounds__(256)
k_fsc_rows(const double *__restrict__ in, xh_cd *__restrict__ out, XhPlan<double> plan, size_t nlines, int X, int xh, int lpb)
{
    extern __shared__ __align__(16) unsigned char fsc_smem[];
    xh_cd *s = reinterpret_cast<xh_cd *>(fsc_smem);
    const int M = 1 << plan.logM;
    const int tid = threadIdx.x, nth = blockDim.x;
    const size_t line0 = (size_t)blockIdx.x * lpb;
    const int nl = (int)min((size_t)lpb, nlines - line0);
    for (int i = tid; i < lpb * X; i += nth) {
        const int l = i / X, e = i - l * X;
        const double v = l < nl ? in[(line0 + l) * X + e] : 0.0;
        s[l * M + xh_plan_pos(plan, e)] = xh_cd{v, 0.0};
    }
    __syncthreads();
    xh_plan_exec<double, false>(s, plan, lpb, tid, nth);
    for (int i = tid; i < nl * xh; i += nth) {
        const int l = i / xh, e = i - l * xh;
        out[(line0 + l) * xh + e] = s[l * M + e];
    }
}

// FFT_IDX2DIGFREQ_FAST (xmippCore xmipp_fft.h)
__device__ __forceinline__ double digfreq(int idx, int size) { return (double)(idx <= size / 2 ? idx : idx - size) / (double)size; }

// realWRAP (xmippCore xmipp_macros.h)
__device__ __forceinline__ double real_wrap(double x, double x0, double xF)
{
    if (x >= x0 && x <= xF) return x;
    if (x < x0) return x - (int)((x - x0) / (xF - x0) - 1) * (xF - x0);
    return x - (int)((x - xF) / (xF - x0) + 1) * (xF - x0);
}

enum { S_NUM = 0, S_DEN1, S_DEN2, S_L2, S_DPRN, S_DPRD, S_COUNT, S_NSUMS };

// sums [S_NSUMS][L] then [2] = (Rn, Rd)
__global__ void __launch_bounds__(256)
k_fsc_shells(const xh_cd *__restrict__ F1, const xh_cd *__restrict__ F2, int Z, int Y, int X, int xh, int L, double inv,
             int dodpr, int dorfactor, double minFreq, double maxFreq, double *__restrict__ sums)
{
    extern __shared__ __align__(16) unsigned char fsc_smem[];
    double *t = reinterpret_cast<double *>(fsc_smem);      // [S_NSUMS][L] + 2
    const int nt = S_NSUMS * L + 2;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) t[i] = 0.0;
    __syncthreads();
    const size_t total = (size_t)Z * Y * xh;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e % xh);
        const size_t r = e / xh;
        const int i = (int)(r % Y), k = (int)(r / Y);
        const double fz = digfreq(k, Z), fy = digfreq(i, Y), fx = digfreq(j, X);
        const double R2 = fz * fz + fy * fy + fx * fx;
        xh_cd z1 = F1[e], z2 = F2[e];
        z1.x *= inv; z1.y *= inv; z2.x *= inv; z2.y *= inv;
        const double absz1 = hypot(z1.x, z1.y), absz2 = hypot(z2.x, z2.y);
        if (dorfactor) {
            const double R = sqrt(R2);
            if (R >= minFreq && R <= maxFreq) {
                atomicAdd(&t[S_NSUMS * L], fabs(absz1 - absz2));
                atomicAdd(&t[S_NSUMS * L + 1], absz1);
            }
        }
        if (R2 > 0.25) continue;
        const int idx = (int)round(sqrt(R2) * X);
        if (idx >= L) continue;     // only reachable for non-cubic inputs; the reference writes past its arrays there
        atomicAdd(&t[S_NUM * L + idx], z1.x * z2.x + z1.y * z2.y);
        atomicAdd(&t[S_DEN1 * L + idx], absz1 * absz1);
        atomicAdd(&t[S_DEN2 * L + idx], absz2 * absz2);
        atomicAdd(&t[S_L2 * L + idx], hypot(z1.x - z2.x, z1.y - z2.y));
        if (dodpr) {
            const double dphi = real_wrap((atan2(z1.y, z1.x) - atan2(z2.y, z2.x)) * 180.0 / 3.14159265358979323846, -180, 180);
            atomicAdd(&t[S_DPRN * L + idx], (absz1 + absz2) * dphi * dphi);
            atomicAdd(&t[S_DPRD * L + idx], absz1 + absz2);
        }
        atomicAdd(&t[S_COUNT * L + idx], 1.0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x)
        if (t[i] != 0.0) atomicAdd(&sums[i], t[i]);
}

int fft3d_r2c(xh_ctx *ctx, const double *d_in, xh_cd *F, int Z, int Y, int X, const XhPlan<double> &px, const XhPlan<double> &py,
              const XhPlan<double> &pz)
{
    const int xh = X / 2 + 1;
    {
        const int lpb = xh_plan_lpb(px, 64 * 1024, 8);
        const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << px.logM, nlines = (size_t)Z * Y;
        hipLaunchKernelGGL(k_fsc_rows, dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, d_in, F, px, nlines, X, xh, lpb);
        XH_LAUNCH_CHECK();
    }
    if (Y > 1) {   // y lines: (k,j) -> offset k*Y*xh + j, element stride xh
        const int lpb = xh_plan_lpb(py, 64 * 1024, 8);
        const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << py.logM, nlines = (size_t)Z * xh;
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, F, py,
                           nlines, (size_t)xh, (size_t)Y * xh, (size_t)1, (size_t)xh, lpb);
        XH_LAUNCH_CHECK();
    }
    if (Z > 1) {   // z lines: (i,j) -> offset i*xh + j, element stride Y*xh
        const int lpb = xh_plan_lpb(pz, 64 * 1024, 8);
        const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << pz.logM, nlines = (size_t)Y * xh;
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, F, pz,
                           nlines, nlines, (size_t)0, (size_t)1, (size_t)Y * xh, lpb);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

}  // namespace

extern "C" int xh_frc_dpr(xh_ctx *ctx, const double *d_m1, const double *d_m2, int32_t Z, int32_t Y, int32_t X, double sampling_rate,
                          int32_t do_dpr, int32_t do_rfactor, double minFreq, double maxFreq, double *h_freq, double *h_frc,
                          double *h_frc_noise, double *h_dpr, double *h_error_l2, double *h_rfactor)
{
    XH_CHECK(ctx && d_m1 && d_m2 && h_freq && h_frc && h_frc_noise && h_error_l2, XH_ERR_ARG, "xh_frc_dpr: null argument");
    XH_CHECK(!do_dpr || h_dpr, XH_ERR_ARG, "xh_frc_dpr: do_dpr without an output array");
    XH_CHECK(!do_rfactor || h_rfactor, XH_ERR_ARG, "xh_frc_dpr: do_rfactor without an output");
    XH_CHECK(Z >= 1 && Y >= 1 && X >= 2, XH_ERR_ARG, "xh_frc_dpr: bad size %d x %d x %d", Z, Y, X);
    XH_CHECK(Z <= 1024 && Y <= 1024 && X <= 1024, XH_ERR_UNSUPPORTED, "xh_frc_dpr: sizes above 1024 are not supported (%d x %d x %d)", Z, Y, X);
    XH_CHECK(sampling_rate > 0, XH_ERR_ARG, "xh_frc_dpr: sampling rate must be positive");
    XH_HIP(hipSetDevice(ctx->device));
    const int xh = X / 2 + 1, L = X / 2 + 1, nt = S_NSUMS * L + 2;
    const size_t total = (size_t)Z * Y * xh;
    XhPlanBufs<double> px, py, pz;
    XhBuf F1, F2, sums;
    int rc = xh_plan_create<double>(ctx, X, px);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, Y, py);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, Z, pz);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, F1, sizeof(xh_cd) * total);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, F2, sizeof(xh_cd) * total);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, sums, sizeof(double) * nt);
    std::vector<double> h((size_t)nt, 0.0);
    if (rc == XH_OK) rc = fft3d_r2c(ctx, d_m1, (xh_cd *)F1.p, Z, Y, X, px.plan, py.plan, pz.plan);
    if (rc == XH_OK) rc = fft3d_r2c(ctx, d_m2, (xh_cd *)F2.p, Z, Y, X, px.plan, py.plan, pz.plan);
    if (rc == XH_OK) {
        hipError_t e = hipMemsetAsync(sums.p, 0, sums.bytes, ctx->stream);
        if (e == hipSuccess) {
            const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)ctx->num_cus * 8);
            hipLaunchKernelGGL(k_fsc_shells, dim3(grid), dim3(256), sizeof(double) * nt, ctx->stream, (const xh_cd *)F1.p, (const xh_cd *)F2.p,
                               Z, Y, X, xh, L, 1.0 / ((double)Z * Y * X), do_dpr, do_rfactor, minFreq, maxFreq, (double *)sums.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h.data(), sums.p, sums.bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { xh_set_error("xh_frc_dpr: %s", hipGetErrorString(e)); rc = XH_ERR_HIP; }
    }
    xh_buf_free(F1); xh_buf_free(F2); xh_buf_free(sums);
    xh_plan_free(px); xh_plan_free(py); xh_plan_free(pz);
    if (rc != XH_OK) return rc;
    for (int i = 0; i < L; ++i) {
        const double count = h[(size_t)S_COUNT * L + i];
        h_freq[i] = (double)i / (X * sampling_rate);
        h_frc[i] = h[(size_t)S_NUM * L + i] / std::sqrt(h[(size_t)S_DEN1 * L + i] * h[(size_t)S_DEN2 * L + i]);
        h_frc_noise[i] = 2 / std::sqrt(count);
        h_error_l2[i] = h[(size_t)S_L2 * L + i] / count;
        if (do_dpr) h_dpr[i] = std::sqrt(h[(size_t)S_DPRN * L + i] / h[(size_t)S_DPRD * L + i]);
    }
    if (do_rfactor) *h_rfactor = h[(size_t)S_NSUMS * L] / h[(size_t)S_NSUMS * L + 1];
    return XH_OK;
}
