// xh_fft2d.hip -- complex 2-D FFTs whose lines do not fit one LDS transform (gfx950).
//
// FlexAlign transforms whole movie frames (4096 x 5760 K3 frames, reconstruction_adapt_cuda/
// movie_alignment_correlation_gpu.cpp:633-725 plans them with cuFFT); the line transforms of xh_plan.h hold a line in
// LDS and stop at 1024 points (4096 for powers of two). A longer line of N = N1 N2 points is done in four steps
// (Bailey), every step a pass of kernels that already exist or a plain streaming kernel:
//     n = N2 n1 + n2,  k = k1 + N1 k2
//     1. N2 transforms of length N1 over n1 (elements N2 apart)                  xh_k_fft_lines, strided
//     2. multiply element (k1, n2) by exp(-+2 pi i k1 n2 / N)                    k_fft2d_twiddle
//     3. N1 transforms of length N2 over n2 (contiguous)                         xh_k_fft_lines
//     4. X[k1 + N1 k2] sits at N2 k1 + k2: transpose the N1 x N2 matrix          k_fft2d_untangle (out of place)
// When N = (odd factor <= 64) x (power of two), step 1 is a direct DFT of the odd length instead (k_fft2d_small: 45 points cost 45
// multiply-adds per output, against three 128-point transforms of the chirp-z form): 5760 = 45 x 128, the K3 frame's long side.
// Rows (contiguous lines) and columns (lines nx apart) take the same four steps with different strides. Un-normalised
// like the line kernels; xh_fft2d_exec divides by ny nx on the inverse.
#include "xh_common.h"
#include "xh_fft.h"
#include "xh_plan.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

typedef xh_c2<float> xh_cf;

namespace {

// one axis: direct (n2 == 1) or four-step
struct Axis {
    int n = 0, n1 = 0, n2 = 1;
    XhPlanBufs<float> p1, p2;        // length n1 (or n when direct), length n2
    XhBuf tw;                        // exp(-2 pi i j / n), j < n (four-step only)
    bool small1 = false;             // step 1 as a direct DFT of n1 <= 64 points (k_fft2d_small)
    XhBuf tw1;                       // exp(-2 pi i j / n1), j < n1
};

// largest line xh_k_fft_lines takes in 64 KB of LDS with at least four lines per workgroup
bool direct_ok(int n)
{
    int M = 1;
    while (M < (xh_is_pow2(n) ? n : 2 * n - 1)) M <<= 1;
    return M <= 2048;
}

// n = n1 * n2 with both factors direct_ok, as balanced as possible, powers of two preferred for n1
bool factorise(int n, int &n1, int &n2)
{
    long best = -1;
    for (int a = 2; (long)a * a <= (long)n * 4 && a <= n / 2; ++a) {
        if (n % a) continue;
        const int b = n / a;
        if (!direct_ok(a) || !direct_ok(b)) continue;
        // cost: LDS line lengths the two passes run at (Bluestein lines cost their padded length three times)
        // (+ 64 per line whatever its length: measured on the 4092 rows of a K3 frame, where 4 x 1023 -- 4-point lines that cost as much
        // as 128-point ones, and Bluestein lines of 2048 -- lost to 44 x 93 by 1.3 ms per 40-frame movie)
        auto cost = [](int m) { int M = 1; while (M < (xh_is_pow2(m) ? m : 2 * m - 1)) M <<= 1; return (long)(xh_is_pow2(m) ? M : 3 * M) + 64; };
        const long c = cost(a) * b + cost(b) * a;
        if (best < 0 || c < best) { best = c; n1 = a; n2 = b; }
    }
    return best >= 0;
}

__global__ void __launch_bounds__(256)
k_fft2d_twiddle(xh_cf *__restrict__ data, const xh_cf *__restrict__ tw, size_t total, int n, int n2, size_t inner, size_t outerStride,
                size_t innerStride, size_t elemStride, int inverse)
{
    // element e = N2 k1 + n2 of line l (same line addressing as xh_k_fft_lines); consecutive threads, consecutive lines
    // of the inner index when the lines are columns, consecutive elements when they are rows
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t l, e;
    if (elemStride == 1) { l = i / n; e = i - l * n; }
    else { e = i / inner % n; l = (i / inner / n) * inner + i % inner; }
    const int k1 = (int)(e / n2), m2 = (int)(e - (size_t)k1 * n2);
    xh_cf w = tw[(size_t)((long long)k1 * m2 % n)];
    if (inverse) w.y = -w.y;
    xh_cf *p = data + (l / inner) * outerStride + (l % inner) * innerStride + e * elemStride;
    *p = xh_cmul(*p, w);
}

// out[line][k1 + n1 k2] = in[line][n2 k1 + k2]
__global__ void __launch_bounds__(256)
k_fft2d_untangle(const xh_cf *__restrict__ in, xh_cf *__restrict__ out, size_t total, int n, int n1, int n2, size_t inner,
                 size_t outerStride, size_t innerStride, size_t elemStride, float scale)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t l, k;                                  // k: output element, consecutive threads write consecutive addresses
    if (elemStride == 1) { l = i / n; k = i - l * n; }
    else { k = i / inner % n; l = (i / inner / n) * inner + i % inner; }
    const int k2 = (int)(k / n1), k1 = (int)(k - (size_t)k2 * n1);
    const size_t base = (l / inner) * outerStride + (l % inner) * innerStride;
    xh_cf v = in[base + ((size_t)k1 * n2 + k2) * elemStride];
    v.x *= scale; v.y *= scale;
    out[base + k * elemStride] = v;
}

__global__ void __launch_bounds__(256) k_fft2d_scale(xh_cf *__restrict__ d, size_t total, float scale)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { d[i].x *= scale; d[i].y *= scale; }
}

// Direct DFT of short lines (n <= 64 points, elemStride apart; consecutive lines are innerStride apart, `inner` of them per outer
// step, like xh_k_fft_lines). A block stages 128 lines in LDS ([n][128]: a wave reads 64 neighbouring lines of one element, no
// conflicts) and its two halves share the outputs of a line (k = half, half + 2, ...); the twiddle of a (j, k) is the same for the
// whole wave. In place.
template <bool INV>
__global__ void __launch_bounds__(256) k_fft2d_small(xh_cf *__restrict__ data, const xh_cf *__restrict__ tw, int n, size_t nlines, size_t inner, size_t outerStride,
                                                     size_t innerStride, size_t elemStride)
{
    extern __shared__ xh_cf ssm[];            // [n][128] samples, then [n] twiddles
    xh_cf *sx = ssm, *sw = ssm + (size_t)n * 128;
    const size_t l0 = (size_t)blockIdx.x * 128;
    for (int j = threadIdx.x; j < n; j += 256) { xh_cf w = tw[j]; if (INV) w.y = -w.y; sw[j] = w; }
    for (int i = threadIdx.x; i < n * 128; i += 256) {
        const int j = i >> 7, ll = i & 127;
        const size_t l = l0 + ll;
        sx[i] = l < nlines ? data[(l / inner) * outerStride + (l % inner) * innerStride + (size_t)j * elemStride] : xh_cf{0.f, 0.f};
    }
    __syncthreads();
    const int ll = threadIdx.x & 127, half = threadIdx.x >> 7;
    const size_t l = l0 + ll;
    if (l >= nlines) return;
    xh_cf *dst = data + (l / inner) * outerStride + (l % inner) * innerStride;
    for (int k = half; k < n; k += 2) {
        float re = 0.f, im = 0.f;
        int m = 0;
        for (int j = 0; j < n; ++j) {
            const xh_cf x = sx[(j << 7) + ll], w = sw[m];
            re += x.x * w.x - x.y * w.y; im += x.x * w.y + x.y * w.x;
            m += k; if (m >= n) m -= n;
        }
        dst[(size_t)k * elemStride] = xh_cf{re, im};
    }
}

// The first two steps of a four-step ROW transform whose input is a real frame, two rows per complex row (row 2r real part, row
// 2r + 1 imaginary part, (frame - dark) * gain like loadFrame): the n1-point DFTs over n1 (n1 odd, <= 64), times the step-2 factor
// exp(-2 pi i k1 m2 / n) -- the pass k_fa_load2 + k_fft2d_small + k_fft2d_twiddle make in three trips through memory.  out[r][k1 n2 + m2].
__global__ void __launch_bounds__(256) k_fft2d_small_pairs(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain, int Y, int X,
                                                           xh_cf *__restrict__ out, const xh_cf *__restrict__ tw1, const xh_cf *__restrict__ twN, int n1, int n2,
                                                           size_t nlines)
{
    extern __shared__ xh_cf ssm[];            // [n1][128] samples, then [n1] twiddles
    xh_cf *sx = ssm, *sw = ssm + (size_t)n1 * 128;
    const size_t l0 = (size_t)blockIdx.x * 128;
    for (int j = threadIdx.x; j < n1; j += 256) sw[j] = tw1[j];
    for (int i = threadIdx.x; i < n1 * 128; i += 256) {
        const int j = i >> 7, ll = i & 127;
        const size_t l = l0 + ll;
        xh_cf v = xh_cf{0.f, 0.f};
        if (l < nlines) {
            const size_t r = l / n2, m2 = l - r * n2;
            const size_t s0 = (2 * r) * (size_t)X + m2 + (size_t)j * n2, s1 = s0 + X;
            float v0 = frame[s0], v1 = 0.f;
            if (dark) v0 -= dark[s0];
            if (gain) v0 *= gain[s0];
            if (2 * r + 1 < (size_t)Y) {
                v1 = frame[s1];
                if (dark) v1 -= dark[s1];
                if (gain) v1 *= gain[s1];
            }
            v = xh_cf{v0, v1};
        }
        sx[i] = v;
    }
    __syncthreads();
    const int ll = threadIdx.x & 127, half = threadIdx.x >> 7;
    const size_t l = l0 + ll;
    if (l >= nlines) return;
    const size_t r = l / n2, m2 = l - r * n2;
    xh_cf *dst = out + r * (size_t)X + m2;
    for (int k = half; k < n1; k += 2) {
        float re = 0.f, im = 0.f;
        int m = 0;
        for (int j = 0; j < n1; ++j) {
            const xh_cf x = sx[(j << 7) + ll], w = sw[m];
            re += x.x * w.x - x.y * w.y; im += x.x * w.y + x.y * w.x;
            m += k; if (m >= n1) m -= n1;
        }
        dst[(size_t)k * n2] = xh_cmul(xh_cf{re, im}, twN[(size_t)k * m2]);         // k m2 < n1 n2 = n
    }
}


// ---- 45 points in registers: 45 = 5 x 9 (Cooley-Tukey, every index a compile-time constant).  W45^j = exp(-2 pi i j / 45).
__device__ __forceinline__ xh_cf xh_w45(int j)
{
    constexpr float c[45] = {1.000000000e+00f, 9.902680687e-01f, 9.612616959e-01f, 9.135454576e-01f, 8.480480962e-01f, 7.660444431e-01f, 6.691306064e-01f, 5.591929035e-01f, 4.383711468e-01f, 3.090169944e-01f, 1.736481777e-01f, 3.489949670e-02f, -1.045284633e-01f, -2.419218956e-01f, -3.746065934e-01f, -5.000000000e-01f, -6.156614753e-01f, -7.193398003e-01f, -8.090169944e-01f, -8.829475929e-01f, -9.396926208e-01f, -9.781476007e-01f, -9.975640503e-01f, -9.975640503e-01f, -9.781476007e-01f, -9.396926208e-01f, -8.829475929e-01f, -8.090169944e-01f, -7.193398003e-01f, -6.156614753e-01f, -5.000000000e-01f, -3.746065934e-01f, -2.419218956e-01f, -1.045284633e-01f, 3.489949670e-02f, 1.736481777e-01f, 3.090169944e-01f, 4.383711468e-01f, 5.591929035e-01f, 6.691306064e-01f, 7.660444431e-01f, 8.480480962e-01f, 9.135454576e-01f, 9.612616959e-01f, 9.902680687e-01f};
    constexpr float s[45] = {-0.000000000e+00f, -1.391731010e-01f, -2.756373558e-01f, -4.067366431e-01f, -5.299192642e-01f, -6.427876097e-01f, -7.431448255e-01f, -8.290375726e-01f, -8.987940463e-01f, -9.510565163e-01f, -9.848077530e-01f, -9.993908270e-01f, -9.945218954e-01f, -9.702957263e-01f, -9.271838546e-01f, -8.660254038e-01f, -7.880107536e-01f, -6.946583705e-01f, -5.877852523e-01f, -4.694715628e-01f, -3.420201433e-01f, -2.079116908e-01f, -6.975647374e-02f, 6.975647374e-02f, 2.079116908e-01f, 3.420201433e-01f, 4.694715628e-01f, 5.877852523e-01f, 6.946583705e-01f, 7.880107536e-01f, 8.660254038e-01f, 9.271838546e-01f, 9.702957263e-01f, 9.945218954e-01f, 9.993908270e-01f, 9.848077530e-01f, 9.510565163e-01f, 8.987940463e-01f, 8.290375726e-01f, 7.431448255e-01f, 6.427876097e-01f, 5.299192642e-01f, 4.067366431e-01f, 2.756373558e-01f, 1.391731010e-01f};
    return xh_cf{c[j], s[j]};
}
// a[9 n1 + n2] = x[9 n1 + n2] on entry; X[k1 + 5 k2] = a[9 k1 + k2] on return
__device__ __forceinline__ void xh_dft45(xh_cf (&a)[45])
{
#pragma unroll
    for (int n2 = 0; n2 < 9; ++n2) {
        xh_cf t[5];
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) {
            xh_cf acc = a[n2];
#pragma unroll
            for (int n1 = 1; n1 < 5; ++n1) {
                const int e = (9 * n1 * k1) % 45;
                const xh_cf x = a[9 * n1 + n2];
                if (e == 0) { acc.x += x.x; acc.y += x.y; }
                else { const xh_cf w = xh_w45(e); acc.x += x.x * w.x - x.y * w.y; acc.y += x.x * w.y + x.y * w.x; }
            }
            const int e2 = (n2 * k1) % 45;
            t[k1] = e2 == 0 ? acc : xh_cmul(acc, xh_w45(e2));
        }
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) a[9 * k1 + n2] = t[k1];
    }
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
        xh_cf t[9];
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) {
            xh_cf acc = a[9 * k1];
#pragma unroll
            for (int n2 = 1; n2 < 9; ++n2) {
                const int e = (5 * n2 * k2) % 45;
                const xh_cf x = a[9 * k1 + n2];
                if (e == 0) { acc.x += x.x; acc.y += x.y; }
                else { const xh_cf w = xh_w45(e); acc.x += x.x * w.x - x.y * w.y; acc.y += x.x * w.y + x.y * w.x; }
            }
            t[k2] = acc;
        }
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) a[9 * k1 + k2] = t[k2];
    }
}

// k_fft2d_small_pairs for n1 = 45 (a K3 frame's 5760 = 45 x 128): a thread owns a line -- its 45 samples are 45 coalesced loads (lanes
// are neighbouring m2), the transform runs in registers with constant factors (630 complex multiply-adds instead of 2025, no LDS), the
// 45 results are 45 coalesced stores.
__global__ void __launch_bounds__(256) k_fft2d_45_pairs(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain, int Y, int X,
                                                        xh_cf *__restrict__ out, const xh_cf *__restrict__ twN, int n2, size_t nlines)
{
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines) return;
    const size_t r = l / n2, m2 = l - r * n2;
    const bool two = 2 * r + 1 < (size_t)Y;
    const size_t s0 = (2 * r) * (size_t)X + m2;
    xh_cf a[45];
#pragma unroll
    for (int j = 0; j < 45; ++j) {
        const size_t o = s0 + (size_t)j * n2;
        float v0 = frame[o], v1 = two ? frame[o + X] : 0.f;
        if (dark) { v0 -= dark[o]; if (two) v1 -= dark[o + X]; }
        if (gain) { v0 *= gain[o]; if (two) v1 *= gain[o + X]; }
        a[j] = xh_cf{v0, v1};
    }
    xh_dft45(a);
    xh_cf *dst = out + r * (size_t)X + m2;
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) {
            const int k = k1 + 5 * k2;
            dst[(size_t)k * n2] = xh_cmul(a[9 * k1 + k2], twN[(size_t)k * m2]);         // k m2 < 45 n2 = n
        }
}

// The whole row pass of a frame whose rows are 5760 = 45 x 128 points when only the nc <= 45 PL lowest frequencies of every row are kept
// (FlexAlign's reduced frame keeps 653 of a K3 frame's 2881): one workgroup per pair of rows (row 2r real part, row 2r + 1 imaginary
// part), nothing but the frame read and the kept columns written.
//   1. the 45-point transform over j of z[128 j + m] (a lane per pair of lines m, m + 64, a wave per quarter of the outputs k1, the factors
//      of a j by scalar loads from W45[j][48], packed multiply-adds), times exp(-2 pi i k1 m / 5760), into LDS T[m][k1];
//   2. the 128-point transform over m for the 2 PL kept k2 only (k = k1 + 45 k2: k2 < PL and k2 >= 128 - PL, the latter for the
//      mirror frequencies the separation of the two rows needs): a lane per k1, eight k2 per wave, the factors exp(-2 pi i m k2 / 128) of
//      an m wave-uniform (scalar loads from Wq[m][32]), every complex multiply-add two packed fused multiply-adds;
//   3. the two rows apart, F(a)[k] = (Z[k] + conj Z[-k]) / 2, F(b)[k] = (Z[k] - conj Z[-k]) / 2i, written as C[2r][k], C[2r + 1][k], k < nc.
// 1.8 G packed multiply-adds and 115 MB of traffic per K3 frame instead of three kernels and 400 MB.
typedef float xh_v2 __attribute__((ext_vector_type(2)));
template <bool CG>          // CG: dark and / or gain given
__global__ void __launch_bounds__(256) k_fft2d_45x128_rows_kept(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain, int Y, int X,
                                                                const xh_cf *__restrict__ twN, const xh_v2 *__restrict__ W45, const xh_v2 *__restrict__ Wq, int PL, int nc, xh_cf *__restrict__ C)
{
    __shared__ xh_v2 T[128 * 45];
    const int r = blockIdx.x, t = threadIdx.x;
    const bool two = 2 * r + 1 < Y;
    {
        // a lane per pair of lines (m, m + 64), a wave per quarter of the 45 outputs: 2 x 45 samples in, 2 x 12 sums; the twelve factors
        // W45[j][k1] of a j are the same for the whole wave and serve 48 packed multiply-adds (with one line per lane and 24 outputs per
        // wave it was 24 factors for the same 48: the wave waited for its scalar loads)
        const int m = t & 63, hq = __builtin_amdgcn_readfirstlane(t >> 6);
        const size_t s0 = (size_t)(2 * r) * X + m;
        xh_v2 acc[2][12];
#pragma unroll
        for (int q = 0; q < 12; ++q) { acc[0][q] = xh_v2{0.f, 0.f}; acc[1][q] = xh_v2{0.f, 0.f}; }
        const xh_v2 *w = W45 + hq * 12;
        auto fetch = [&](int j, float (&v)[4]) {
            const size_t o = s0 + (size_t)j * 128;
            v[0] = frame[o]; v[1] = two ? frame[o + X] : 0.f; v[2] = frame[o + 64]; v[3] = two ? frame[o + 64 + X] : 0.f;
            if (CG) {
                if (dark) { v[0] -= dark[o]; v[2] -= dark[o + 64]; if (two) { v[1] -= dark[o + X]; v[3] -= dark[o + 64 + X]; } }
                if (gain) { v[0] *= gain[o]; v[2] *= gain[o + 64]; if (two) { v[1] *= gain[o + X]; v[3] *= gain[o + 64 + X]; } }
            }
        };
        // three samples ahead: the loads of group g + 1 fly while group g is summed
        float c[3][4], n[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u) fetch(u, c[u]);
        for (int g = 0; g < 15; ++g) {
            if (g < 14) {
#pragma unroll
                for (int u = 0; u < 3; ++u) fetch(3 * (g + 1) + u, n[u]);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u, w += 48) {
                const xh_v2 ar = xh_v2{c[u][0], c[u][0]}, ai = xh_v2{-c[u][1], c[u][1]}, br = xh_v2{c[u][2], c[u][2]}, bi = xh_v2{-c[u][3], c[u][3]};
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const xh_v2 f = w[q], fs = xh_v2{f.y, f.x};
                    acc[0][q] = __builtin_elementwise_fma(ar, f, acc[0][q]);
                    acc[0][q] = __builtin_elementwise_fma(ai, fs, acc[0][q]);
                    acc[1][q] = __builtin_elementwise_fma(br, f, acc[1][q]);
                    acc[1][q] = __builtin_elementwise_fma(bi, fs, acc[1][q]);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) c[u][e] = n[u][e];
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int k = hq * 12 + q;
            if (k < 45) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int mm = m + 64 * e;
                    const xh_cf v = xh_cmul(xh_cf{acc[e][q].x, acc[e][q].y}, twN[(size_t)k * mm]);         // k m < 45 x 128
                    T[mm * 45 + k] = xh_v2{v.x, v.y};
                }
            }
        }
    }
    __syncthreads();
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), k1 = min(lane, 44);
    xh_v2 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = xh_v2{0.f, 0.f};
    const xh_v2 *w = Wq + wv * 8;
    for (int m0 = 0; m0 < 128; m0 += 4) {
        xh_v2 z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) z[u] = T[(m0 + u) * 45 + k1];
#pragma unroll
        for (int u = 0; u < 4; ++u, w += 32) {
            const xh_v2 vr = xh_v2{z[u].x, z[u].x}, vi = xh_v2{-z[u].y, z[u].y};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const xh_v2 f = w[q];
                acc[q] = __builtin_elementwise_fma(vr, f, acc[q]);
                acc[q] = __builtin_elementwise_fma(vi, xh_v2{f.y, f.x}, acc[q]);
            }
        }
    }
    __syncthreads();
    // Z of the kept frequencies over T: Zs[q][k1], q < PL: k = k1 + 45 q; PL <= q < 2 PL: k = k1 + 45 (128 - 2 PL + q)
    if (lane < 45)
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(wv * 8 + q) * 45 + lane] = acc[q];
    __syncthreads();
    for (int k = t; k < nc; k += 256) {
        const xh_v2 zk = T[k];                                                        // q = k / 45, k1 = k % 45: q 45 + k1 = k
        const int km = k ? X - k : 0;                                                 // frequency -k
        const xh_v2 zm = T[k ? km - 45 * (128 - 2 * PL) : 0];
        C[(size_t)(2 * r) * nc + k] = xh_cf{0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        if (two) C[(size_t)(2 * r + 1) * nc + k] = xh_cf{0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};
    }
}

int small_lines(xh_ctx *ctx, xh_cf *data, const Axis &A, size_t nlines, size_t inner, size_t outerStride, size_t innerStride, size_t elemStride, bool inverse)
{
    const size_t smem = sizeof(xh_cf) * ((size_t)A.n1 * 128 + A.n1);
    const unsigned grid = (unsigned)((nlines + 127) / 128);
    if (inverse)
        hipLaunchKernelGGL((k_fft2d_small<true>), dim3(grid), dim3(256), smem, ctx->stream, data, (const xh_cf *)A.tw1.p, A.n1, nlines, inner, outerStride, innerStride, elemStride);
    else
        hipLaunchKernelGGL((k_fft2d_small<false>), dim3(grid), dim3(256), smem, ctx->stream, data, (const xh_cf *)A.tw1.p, A.n1, nlines, inner, outerStride, innerStride, elemStride);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

int axis_create(xh_ctx *ctx, int n, Axis &A)
{
    A.n = n;
    if (direct_ok(n)) {
        A.n1 = n; A.n2 = 1;
        return xh_plan_create<float>(ctx, n, A.p1);
    }
    // an odd factor of at most 64 points times a power of two: the odd part directly, the rest by the radix kernels
    int odd = n, p2 = 1;
    while ((odd & 1) == 0) { odd >>= 1; p2 <<= 1; }
    if (odd > 1 && odd <= 64 && p2 >= 2 && direct_ok(p2) && !xh_debug_env("XH_FFT2D_NO_SMALL")) { A.n1 = odd; A.n2 = p2; A.small1 = true; }
    else {
        XH_CHECK(factorise(n, A.n1, A.n2), XH_ERR_UNSUPPORTED, "xh_fft2d: %d has no factorisation into two LDS-sized line lengths", n);
        if (const char *f = xh_debug_env("XH_FFT2D_N1")) {              // A/B runs: force the first factor
            const int a = atoi(f);
            if (a >= 2 && n % a == 0 && direct_ok(a) && direct_ok(n / a)) { A.n1 = a; A.n2 = n / a; }
        }
    }
    if (A.small1) {
        const long double PI1 = 3.14159265358979323846264338327950288L;
        std::vector<xh_cf> w1(A.n1);
        for (int j = 0; j < A.n1; ++j) { const long double a = -2.0L * PI1 * j / A.n1; w1[j] = xh_cf{(float)cosl(a), (float)sinl(a)}; }
        XH_TRY(xh_buf_alloc(ctx, A.tw1, sizeof(xh_cf) * A.n1));
        XH_HIP(hipMemcpy(A.tw1.p, w1.data(), A.tw1.bytes, hipMemcpyHostToDevice));
    } else XH_TRY(xh_plan_create<float>(ctx, A.n1, A.p1));
    XH_TRY(xh_plan_create<float>(ctx, A.n2, A.p2));
    const long double PI = 3.14159265358979323846264338327950288L;
    std::vector<xh_cf> w(n);
    for (int j = 0; j < n; ++j) {
        const long double a = -2.0L * PI * j / n;
        w[j] = xh_cf{(float)cosl(a), (float)sinl(a)};
    }
    XH_TRY(xh_buf_alloc(ctx, A.tw, sizeof(xh_cf) * n));
    XH_HIP(hipMemcpy(A.tw.p, w.data(), A.tw.bytes, hipMemcpyHostToDevice));
    return XH_OK;
}

void axis_free(Axis &A)
{
    xh_plan_free(A.p1);
    xh_plan_free(A.p2);
    xh_buf_free(A.tw);
    xh_buf_free(A.tw1);
}

int lines(xh_ctx *ctx, xh_cf *data, const XhPlan<float> &plan, size_t nlines, size_t inner, size_t outerStride, size_t innerStride,
          size_t elemStride, bool inverse)
{
    const int lpb = xh_plan_lpb(plan, 64 * 1024, 16);
    const size_t smem = ((size_t)lpb * sizeof(xh_cf)) << plan.logM;
    const unsigned grid = (unsigned)((nlines + lpb - 1) / lpb);
    if (inverse)
        hipLaunchKernelGGL((xh_k_fft_lines<float, true>), dim3(grid), dim3(256), smem, ctx->stream, data, plan, nlines, inner, outerStride,
                           innerStride, elemStride, lpb);
    else
        hipLaunchKernelGGL((xh_k_fft_lines<float, false>), dim3(grid), dim3(256), smem, ctx->stream, data, plan, nlines, inner, outerStride,
                           innerStride, elemStride, lpb);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

// Transforms `count` lines of A.n points: line l starts at (l / inner) * outerStride + (l % inner) * innerStride, elements
// elemStride apart. Result in `data` (direct) or in `other` (four-step: the untangling pass is out of place); *inData says which.
int axis_exec(xh_ctx *ctx, const Axis &A, xh_cf *data, xh_cf *other, size_t count, size_t inner, size_t outerStride, size_t innerStride,
              size_t elemStride, bool inverse, float scale, bool *inData)
{
    if (A.n2 == 1) {
        XH_TRY(lines(ctx, data, A.p1.plan, count, inner, outerStride, innerStride, elemStride, inverse));
        *inData = true;
        return XH_OK;
    }
    const size_t n1 = A.n1, n2 = A.n2;
    if (elemStride == 1) {
        // rows: sub-line (l, n2) starts at base(l) + n2, elements n2 apart; needs inner == count (one row after the other)
        XH_CHECK(inner == count || innerStride == outerStride / inner, XH_ERR_ARG, "xh_fft2d: rows must be evenly spaced");
        const size_t rowStride = inner == count ? innerStride : outerStride / inner;
        if (A.small1) XH_TRY(small_lines(ctx, data, A, count * n2, n2, rowStride, 1, n2, inverse));
        else XH_TRY(lines(ctx, data, A.p1.plan, count * n2, n2, rowStride, 1, n2, inverse));
        const size_t total = count * A.n;
        hipLaunchKernelGGL(k_fft2d_twiddle, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, data, (const xh_cf *)A.tw.p, total,
                           A.n, A.n2, (size_t)1, rowStride, (size_t)0, (size_t)1, inverse ? 1 : 0);
        XH_LAUNCH_CHECK();
        XH_TRY(lines(ctx, data, A.p2.plan, count * n1, n1, rowStride, n2, 1, inverse));
        hipLaunchKernelGGL(k_fft2d_untangle, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const xh_cf *)data, other, total,
                           A.n, A.n1, A.n2, (size_t)1, rowStride, (size_t)0, (size_t)1, scale);
        XH_LAUNCH_CHECK();
    } else {
        // columns of a row-major array: count = nx columns, elemStride = nx
        XH_CHECK(inner == count && innerStride == 1, XH_ERR_ARG, "xh_fft2d: columns must be neighbours");
        const size_t nx = elemStride;
        if (A.small1) XH_TRY(small_lines(ctx, data, A, count * n2, count, nx, 1, nx * n2, inverse));
        else XH_TRY(lines(ctx, data, A.p1.plan, count * n2, count, nx, 1, nx * n2, inverse));            // (x, n2): start n2 nx + x
        const size_t total = count * A.n;
        hipLaunchKernelGGL(k_fft2d_twiddle, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, data, (const xh_cf *)A.tw.p, total,
                           A.n, A.n2, count, (size_t)0, (size_t)1, nx, inverse ? 1 : 0);
        XH_LAUNCH_CHECK();
        XH_TRY(lines(ctx, data, A.p2.plan, count * n1, count, nx * n2, 1, nx, inverse));            // (x, k1): start k1 n2 nx + x
        hipLaunchKernelGGL(k_fft2d_untangle, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const xh_cf *)data, other, total,
                           A.n, A.n1, A.n2, count, (size_t)0, (size_t)1, nx, scale);
        XH_LAUNCH_CHECK();
    }
    *inData = false;
    return XH_OK;
}

}  // namespace

struct xh_fft2d {
    xh_ctx *ctx;
    int ny, nx;
    Axis ax, ay;
    XhBuf tmp;
    XhBuf keptW;         // xh_fft2d_rows_of_real_pairs_kept: exp(-2 pi i m k2 / 128) of the kept k2, [128][32]
    int keptPL = 0;
    XhBuf user;          // scratch of the callers that transform frame after frame with one plan (xh_fft2d_user_scratch): grow-only
};

// internal (xh_common.h): a device buffer of at least `bytes` that lives with the plan
int xh_fft2d_user_scratch(xh_fft2d *f, size_t bytes, void **p)
{
    XH_CHECK(f && p, XH_ERR_ARG, "xh_fft2d_user_scratch: bad argument");
    XH_TRY(xh_buf_reserve(f->ctx, f->user, bytes));
    *p = f->user.p;
    return XH_OK;
}

// internal (xh_common.h): the forward transform of the rows of a real frame, two rows per complex row of the plan (ny = (Y + 1) / 2,
// nx = X), steps 1-3 of the four steps only: d_work[r][n2 k1 + k2] = X_r[k1 + n1 k2] -- the caller reads the columns it keeps through
// that map instead of paying for the untangling pass.  *n1 = 0: this plan's rows are not (odd <= 64) x (power of two); nothing done.
int xh_fft2d_rows_of_real_pairs(xh_fft2d *f, const float *d_frame, const float *d_dark, const float *d_gain, int Y, float *d_work, int *n1, int *n2)
{
    XH_CHECK(f && d_frame && d_work && n1 && n2 && (Y + 1) / 2 == f->ny, XH_ERR_ARG, "xh_fft2d_rows_of_real_pairs: bad argument");
    const Axis &A = f->ax;
    *n1 = 0; *n2 = 0;
    if (A.n2 == 1 || !A.small1) return XH_OK;
    xh_ctx *ctx = f->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t nlines = (size_t)f->ny * A.n2;
    const size_t smem = sizeof(xh_cf) * ((size_t)A.n1 * 128 + A.n1);
    if (A.n1 == 45 && !xh_debug_env("XH_FFT2D_NO_45"))
        hipLaunchKernelGGL(k_fft2d_45_pairs, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, ctx->stream, d_frame, d_dark, d_gain, Y, f->nx, (xh_cf *)d_work,
                           (const xh_cf *)A.tw.p, A.n2, nlines);
    else
    hipLaunchKernelGGL(k_fft2d_small_pairs, dim3((unsigned)((nlines + 127) / 128)), dim3(256), smem, ctx->stream, d_frame, d_dark, d_gain, Y, f->nx, (xh_cf *)d_work,
                       (const xh_cf *)A.tw1.p, (const xh_cf *)A.tw.p, A.n1, A.n2, nlines);
    XH_LAUNCH_CHECK();
    XH_TRY(lines(ctx, (xh_cf *)d_work, A.p2.plan, (size_t)f->ny * A.n1, (size_t)A.n1, (size_t)f->nx, (size_t)A.n2, 1, false));
    *n1 = A.n1; *n2 = A.n2;
    return XH_OK;
}

// internal (xh_common.h): the row pass of a real frame for a caller that keeps the nc lowest frequencies of every row only, the two rows of
// a complex row already apart: d_C [Y][nc].  *done = 0 (nothing launched) unless the rows are 45 x 128 points and nc <= 720.
int xh_fft2d_rows_of_real_pairs_kept(xh_fft2d *f, const float *d_frame, const float *d_dark, const float *d_gain, int Y, int nc, float *d_C, int *done)
{
    XH_CHECK(f && d_frame && d_C && done && (Y + 1) / 2 == f->ny, XH_ERR_ARG, "xh_fft2d_rows_of_real_pairs_kept: bad argument");
    *done = 0;
    const Axis &A = f->ax;
    if (A.n2 != 128 || !A.small1 || A.n1 != 45 || nc < 1 || nc > 720 || xh_debug_env("XH_FFT2D_NO_45")) return XH_OK;
    xh_ctx *ctx = f->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int PL = (nc + 44) / 45;
    if (f->keptPL != PL) {
        const long double PI1 = 3.14159265358979323846264338327950288L;
        std::vector<xh_cf> W((size_t)128 * 32 + 45 * 48, xh_cf{0.f, 0.f});
        for (int j = 0; j < 45; ++j)
            for (int k = 0; k < 45; ++k) {
                const long double a = -2.0L * PI1 * ((j * k) % 45) / 45;
                W[(size_t)128 * 32 + (size_t)j * 48 + k] = xh_cf{(float)cosl(a), (float)sinl(a)};
            }
        for (int m = 0; m < 128; ++m)
            for (int q = 0; q < 2 * PL; ++q) {
                const int k2 = q < PL ? q : 128 - 2 * PL + q;
                const long double a = -2.0L * PI1 * ((m * k2) % 128) / 128;
                W[(size_t)m * 32 + q] = xh_cf{(float)cosl(a), (float)sinl(a)};
            }
        XH_TRY(xh_buf_reserve(ctx, f->keptW, sizeof(xh_cf) * W.size()));
        XH_HIP(hipMemcpyAsync(f->keptW.p, W.data(), sizeof(xh_cf) * W.size(), hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(hipStreamSynchronize(ctx->stream));
        f->keptPL = PL;
    }
    if (d_dark || d_gain)
        hipLaunchKernelGGL((k_fft2d_45x128_rows_kept<true>), dim3((unsigned)f->ny), dim3(256), 0, ctx->stream, d_frame, d_dark, d_gain, Y, f->nx, (const xh_cf *)A.tw.p,
                           (const xh_v2 *)f->keptW.p + 128 * 32, (const xh_v2 *)f->keptW.p, PL, nc, (xh_cf *)d_C);
    else
        hipLaunchKernelGGL((k_fft2d_45x128_rows_kept<false>), dim3((unsigned)f->ny), dim3(256), 0, ctx->stream, d_frame, d_dark, d_gain, Y, f->nx, (const xh_cf *)A.tw.p,
                           (const xh_v2 *)f->keptW.p + 128 * 32, (const xh_v2 *)f->keptW.p, PL, nc, (xh_cf *)d_C);
    XH_LAUNCH_CHECK();
    *done = 1;
    return XH_OK;
}

extern "C" {

int xh_fft2d_create(xh_ctx *ctx, int32_t ny, int32_t nx, xh_fft2d **out)
{
    XH_CHECK(ctx && out && ny >= 1 && nx >= 1, XH_ERR_ARG, "xh_fft2d_create: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    xh_fft2d *f = new xh_fft2d;
    f->ctx = ctx; f->ny = ny; f->nx = nx;
    int rc = axis_create(ctx, nx, f->ax);
    if (rc == XH_OK) rc = axis_create(ctx, ny, f->ay);
    if (rc == XH_OK && (f->ax.n2 > 1 || f->ay.n2 > 1)) rc = xh_buf_alloc(ctx, f->tmp, sizeof(xh_cf) * (size_t)ny * nx);
    if (rc != XH_OK) { axis_free(f->ax); axis_free(f->ay); xh_buf_free(f->tmp); delete f; return rc; }
    *out = f;
    return XH_OK;
}

int xh_fft2d_destroy(xh_fft2d *f)
{
    if (!f) return XH_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    axis_free(f->ax); axis_free(f->ay); xh_buf_free(f->tmp);
    xh_buf_free(f->user); xh_buf_free(f->keptW); delete f;
    return XH_OK;
}

int xh_fft2d_factors(const xh_fft2d *f, int32_t *h_factors)
{
    XH_CHECK(f && h_factors, XH_ERR_ARG, "xh_fft2d_factors: bad argument");
    h_factors[0] = f->ay.n1; h_factors[1] = f->ay.n2; h_factors[2] = f->ax.n1; h_factors[3] = f->ax.n2;
    return XH_OK;
}

int xh_fft2d_exec(xh_fft2d *f, float *d_data, int32_t inverse)
{
    XH_CHECK(f && d_data, XH_ERR_ARG, "xh_fft2d_exec: bad argument");
    xh_ctx *ctx = f->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t ny = f->ny, nx = f->nx;
    xh_cf *cur = (xh_cf *)d_data, *oth = (xh_cf *)f->tmp.p;
    const float scale = inverse ? 1.0f / ((float)ny * (float)nx) : 1.0f;
    bool scaled = false, inCur = true;
    // rows: ny lines of nx contiguous points, nx apart
    XH_TRY(axis_exec(ctx, f->ax, cur, oth, ny, ny, 0, nx, 1, inverse != 0, f->ax.n2 > 1 ? scale : 1.0f, &inCur));
    if (f->ax.n2 > 1) scaled = true;
    if (!inCur) std::swap(cur, oth);
    // columns: nx lines of ny points nx apart
    const bool scaleHere = !scaled && f->ay.n2 > 1;
    XH_TRY(axis_exec(ctx, f->ay, cur, oth, nx, nx, 0, 1, nx, inverse != 0, scaleHere ? scale : 1.0f, &inCur));
    if (scaleHere) scaled = true;
    if (!inCur) std::swap(cur, oth);
    if (cur != (xh_cf *)d_data) XH_HIP(hipMemcpyAsync(d_data, cur, sizeof(xh_cf) * ny * nx, hipMemcpyDeviceToDevice, ctx->stream));
    if (inverse && !scaled) {
        hipLaunchKernelGGL(k_fft2d_scale, dim3((unsigned)((ny * nx + 255) / 256)), dim3(256), 0, ctx->stream, (xh_cf *)d_data, ny * nx, scale);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

// One axis only, un-normalised in both directions: axis 0 = the ny rows, axis 1 = the nx columns. FlexAlign's frame transform
// packs two real rows into one complex row, transforms the rows, keeps the columns the reduced frame needs and transforms
// those down the other axis on a narrower array (xh_flexalign.hip).
int xh_fft2d_exec_axis(xh_fft2d *f, float *d_data, int32_t inverse, int32_t axis)
{
    XH_CHECK(f && d_data && (axis == 0 || axis == 1), XH_ERR_ARG, "xh_fft2d_exec_axis: bad argument");
    xh_ctx *ctx = f->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const size_t ny = f->ny, nx = f->nx;
    xh_cf *cur = (xh_cf *)d_data, *oth = (xh_cf *)f->tmp.p;
    bool inCur = true;
    if (axis == 0) XH_TRY(axis_exec(ctx, f->ax, cur, oth, ny, ny, 0, nx, 1, inverse != 0, 1.0f, &inCur));
    else XH_TRY(axis_exec(ctx, f->ay, cur, oth, nx, nx, 0, 1, nx, inverse != 0, 1.0f, &inCur));
    if (!inCur) XH_HIP(hipMemcpyAsync(d_data, oth, sizeof(xh_cf) * ny * nx, hipMemcpyDeviceToDevice, ctx->stream));
    return XH_OK;
}

}  // extern "C"
