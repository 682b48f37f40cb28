// xh_plan.h -- line FFTs of ANY length in LDS (gfx950): powers of two run the radix-2 kernels of
// xh_fft.h directly, every other length n runs Bluestein's chirp-z form on the next power of two
// M >= 2n-1:
//     X[k] = c[k] * sum_j (x[j] c[j]) b[k-j],   c[j] = exp(-i pi j^2 / n),  b = conj(c)
// i.e. chirp multiply, forward FFT_M (DIF, natural -> bit-reversed), multiply by the precomputed
// spectrum of b (stored bit-reversed, 1/M folded in), inverse FFT_M (DIT, bit-reversed -> natural),
// chirp multiply. The inverse transform conjugates both tables (b is even, so FFT(conj b) = conj FFT(b)).
// Replaces what FFTW does for xmippCore's FourierTransformer on non power-of-two boxes (SURVEY.md
// Appendix B); chirp phases use j^2 mod 2n in integers and long double trigonometry on the host.
#ifndef XH_PLAN_H
#define XH_PLAN_H
#include "xh_common.h"
#include "xh_fft.h"
#include <cmath>
#include <vector>

template <typename T> struct XhPlan {
    int n;                    // transform length
    int logM;                 // LDS line stride is M = 1 << logM (== n for powers of two)
    const xh_c2<T> *W;        // exp(-2 pi i j / M), j < M/2
    const xh_c2<T> *chirp;    // c[j], j < n; null for powers of two
    const xh_c2<T> *vhat;     // FFT_M(b wrapped) / M in bit-reversed order; null for powers of two
};

// where element e of a line goes when it is loaded into LDS
template <typename T> __device__ __forceinline__ int xh_plan_pos(const XhPlan<T> &p, int e)
{
    return p.chirp ? e : xh_bitrev(e, p.logM);
}

// Transforms nlines lines in place. Input: element e at s[l*M + xh_plan_pos(e)], e < n (for
// Bluestein lines the tail [n, M) need not be initialised), already visible to the whole workgroup
// (__syncthreads done). Output: X[k] at s[l*M + k], k < n, visible on return. Un-normalised.
template <typename T, bool INV>
__device__ __forceinline__ void xh_plan_exec(xh_c2<T> *s, const XhPlan<T> &p, int nlines, int tid, int nth)
{
    if (!p.chirp) {
        xh_fft_dit<T, INV>(s, p.logM, nlines, p.W, p.logM, tid, nth);
        return;
    }
    const int M = 1 << p.logM, n = p.n;
    for (int i = tid; i < nlines * M; i += nth) {
        const int e = i & (M - 1);
        xh_c2<T> v = xh_c2<T>{0, 0};
        if (e < n) {
            xh_c2<T> c = p.chirp[e];
            if (INV) c.y = -c.y;
            v = xh_cmul(s[i], c);
        }
        s[i] = v;
    }
    __syncthreads();
    xh_fft_dif<T, false>(s, p.logM, nlines, p.W, p.logM, tid, nth);
    for (int i = tid; i < nlines * M; i += nth) {
        xh_c2<T> h = p.vhat[i & (M - 1)];
        if (INV) h.y = -h.y;
        s[i] = xh_cmul(s[i], h);
    }
    __syncthreads();
    xh_fft_dit<T, true>(s, p.logM, nlines, p.W, p.logM, tid, nth);
    for (int i = tid; i < nlines * M; i += nth) {
        const int e = i & (M - 1);
        if (e < n) {
            xh_c2<T> c = p.chirp[e];
            if (INV) c.y = -c.y;
            s[i] = xh_cmul(s[i], c);
        }
    }
    __syncthreads();
}

// Generic in-place strided line transform (un-normalised): line l starts at
// (l / inner) * outerStride + (l % inner) * innerStride, its elements are elemStride apart; lpb lines per
// 256-thread workgroup in ((lpb * sizeof(xh_c2<T>)) << logM) bytes of dynamic LDS. Used for every separable
// 2-D / 3-D transform that is not fused into a neighbouring kernel.
template <typename T, bool INV>
__global__ void __launch_bounds__(256)
xh_k_fft_lines(xh_c2<T> *__restrict__ data, XhPlan<T> plan, size_t nlinesTotal, size_t inner, size_t outerStride,
               size_t innerStride, size_t elemStride, int lpb)
{
    extern __shared__ __align__(16) unsigned char xh_fft_lines_smem[];
    xh_c2<T> *s = reinterpret_cast<xh_c2<T> *>(xh_fft_lines_smem);
    const int n = plan.n, M = 1 << plan.logM;
    const int tid = threadIdx.x, nth = blockDim.x;
    const size_t line0 = (size_t)blockIdx.x * lpb;
    const int nl = (int)min((size_t)lpb, nlinesTotal - line0);
    for (int i = tid; i < lpb * n; i += nth) {
        const int e = i / lpb, l = i - e * lpb;  // consecutive threads -> consecutive lines
        xh_c2<T> v = xh_c2<T>{0, 0};
        if (l < nl) {
            const size_t ln = line0 + l;
            v = data[(ln / inner) * outerStride + (ln % inner) * innerStride + (size_t)e * elemStride];
        }
        s[l * M + xh_plan_pos(plan, e)] = v;
    }
    __syncthreads();
    xh_plan_exec<T, INV>(s, plan, lpb, tid, nth);
    for (int i = tid; i < lpb * n; i += nth) {
        const int e = i / lpb, l = i - e * lpb;
        if (l < nl) {
            const size_t ln = line0 + l;
            data[(ln / inner) * outerStride + (ln % inner) * innerStride + (size_t)e * elemStride] = s[l * M + e];
        }
    }
}

// ---- host side: tables of one plan, owned by the handle that created it
template <typename T> struct XhPlanBufs {
    XhBuf W, chirp, vhat;
    XhPlan<T> plan;
};

template <typename T> static int xh_plan_create(xh_ctx *ctx, int n, XhPlanBufs<T> &b)
{
    const long double PI = 3.14159265358979323846264338327950288L;
    const bool pow2 = xh_is_pow2(n);
    int M = 1;
    while (M < (pow2 ? n : 2 * n - 1)) M <<= 1;
    std::vector<xh_c2<T>> W(std::max(1, M / 2));
    for (int j = 0; j < M / 2; ++j) {
        const long double a = -2.0L * PI * j / M;
        W[j] = xh_c2<T>{(T)cosl(a), (T)sinl(a)};
    }
    XH_TRY(xh_buf_alloc(ctx, b.W, sizeof(xh_c2<T>) * W.size()));
    XH_HIP(hipMemcpy(b.W.p, W.data(), b.W.bytes, hipMemcpyHostToDevice));
    b.plan.n = n;
    b.plan.logM = xh_ilog2(M);
    b.plan.W = (const xh_c2<T> *)b.W.p;
    b.plan.chirp = nullptr;
    b.plan.vhat = nullptr;
    if (pow2) return XH_OK;
    // chirp c[j] = exp(-i pi j^2 / n): reduce j^2 mod 2n exactly
    std::vector<xh_c2<T>> c(n);
    std::vector<xh_c2<long double>> bl(M, xh_c2<long double>{0.L, 0.L});
    for (int j = 0; j < n; ++j) {
        const long long q = ((long long)j * j) % (2LL * n);
        const long double a = PI * (long double)q / (long double)n;
        c[j] = xh_c2<T>{(T)cosl(a), (T)(-sinl(a))};
        const xh_c2<long double> bv{cosl(a), sinl(a)};      // b[j] = conj(c[j])
        bl[j] = bv;
        if (j) bl[M - j] = bv;
    }
    // FFT_M(b) in long double (iterative radix 2), then /M and bit-reversed storage
    {
        const int logM = b.plan.logM;
        for (int i = 1, j = 0; i < M; ++i) {
            int bit = M >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(bl[i], bl[j]);
        }
        for (int len = 2; len <= M; len <<= 1)
            for (int i = 0; i < M; i += len)
                for (int j = 0; j < len / 2; ++j) {
                    const long double a = -2.0L * PI * j / len;
                    const long double wr = cosl(a), wi = sinl(a);
                    const xh_c2<long double> u = bl[i + j], v = bl[i + j + len / 2];
                    const xh_c2<long double> t{v.x * wr - v.y * wi, v.x * wi + v.y * wr};
                    bl[i + j] = xh_c2<long double>{u.x + t.x, u.y + t.y};
                    bl[i + j + len / 2] = xh_c2<long double>{u.x - t.x, u.y - t.y};
                }
        std::vector<xh_c2<T>> vh(M);
        for (int p = 0; p < M; ++p) {
            unsigned r = 0;
            for (int k = 0; k < logM; ++k)
                if (p & (1 << k)) r |= 1u << (logM - 1 - k);
            vh[p] = xh_c2<T>{(T)(bl[r].x / M), (T)(bl[r].y / M)};
        }
        XH_TRY(xh_buf_alloc(ctx, b.vhat, sizeof(xh_c2<T>) * M));
        XH_HIP(hipMemcpy(b.vhat.p, vh.data(), b.vhat.bytes, hipMemcpyHostToDevice));
    }
    XH_TRY(xh_buf_alloc(ctx, b.chirp, sizeof(xh_c2<T>) * n));
    XH_HIP(hipMemcpy(b.chirp.p, c.data(), b.chirp.bytes, hipMemcpyHostToDevice));
    b.plan.chirp = (const xh_c2<T> *)b.chirp.p;
    b.plan.vhat = (const xh_c2<T> *)b.vhat.p;
    return XH_OK;
}

template <typename T> static void xh_plan_free(XhPlanBufs<T> &b)
{
    xh_buf_free(b.W);
    xh_buf_free(b.chirp);
    xh_buf_free(b.vhat);
}

// lines per 256-thread workgroup for a plan, within `budget` bytes of LDS
template <typename T> static int xh_plan_lpb(const XhPlan<T> &p, size_t budget, int maxLines)
{
    const size_t line = sizeof(xh_c2<T>) << p.logM;
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)maxLines, budget / line));
}

#endif
