// xh_flexalign.hip -- FlexAlign, global alignment of a movie (BASELINE config 5: first slice).
//
// Replaces ProgMovieAlignmentCorrelationGPU<T>::computeGlobalAlignment (reconstruction_adapt_cuda/
// movie_alignment_correlation_gpu.cpp:633-725) with the arithmetic of the CPU program the reference can be compared with
// (ProgMovieAlignmentCorrelation<double>, reconstruction/movie_alignment_correlation.cpp:45-157; base class
// movie_alignment_correlation_base.cpp:152-320,399-418; EquationSystemSolver::solve, eq_system_solver.cpp:35-106):
//
//   per frame   (frame - dark) * gain                       loadFrame, base.cpp:152-176
//               2-D FFT of the whole frame (xh_fft2d: 4096 x 5760 lines in four steps)
//               crop to the reduced size + low-pass filter + the normalisations of the reference's two transforms
//                                                            scaleToSizeFourier + createLPF / scaleLPF (loadData, :79-127):
//               the CPU path goes back to real space between crop and filter; a c2r transform drops what is not Hermitian on the
//               DC and Nyquist columns of the cropped half spectrum, so those two columns are symmetrised here instead
//               (k_fa_reduce) -- same spectrum, no round trip. This is what the reference's scaleFFT2DKernel fuses too.
//   per pair    S_a conj(S_b) dSize, inverse transform, centre, bestShift within --maxShift (computeShifts, :129-157;
//               correlation_matrix + bestShift, data/filters.cpp:1593-1719,1745-1754)
//   host        least squares over all pairs with one round of 3-sigma outlier rejection, reference frame = minimax of the X
//               shifts, total shifts from it (solve / findReferenceImage / computeTotalShift)
//
// The spectra of the reduced frames stay resident ([N][nY][nX] complex<float>: 0.39 GB for 40 K3 frames at the default
// 30 A); transforms in fp32 (reference: double), peak statistics and centre of mass in double.
// The patch (local) alignment of movie_alignment_correlation_gpu.cpp:289-430 is not here: the reference has no CPU form of it
// to compare with (movie_alignment_correlation.cpp:63-76 throw "Not implemented").
#include <map>
#include <string>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

#include "xh_common.h"
#include "xh_bspline.h"

namespace {
typedef float2 fa_cf;
typedef float fa_v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ fa_cf fa_conj(fa_cf a) { return fa_cf{a.x, -a.y}; }

__global__ void __launch_bounds__(256) k_fa_load(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain,
                                                 fa_cf *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tot) return;
    float v = frame[t];
    if (dark) v -= dark[t];
    if (gain) v *= gain[t];
    out[t] = fa_cf{v, 0.f};
}

// cropped half spectrum of scaleToSizeFourier: rows 0 .. ihalf-1 from the top, the last ihalf-2 rows from the bottom
__device__ __forceinline__ fa_cf d_fa_crop(const fa_cf *__restrict__ B, int Y, int X, int nY, int ihalf, int i, int j)
{
    if (i < ihalf) return B[(size_t)i * X + j];
    const int n = nY - i;                       // 1 .. : row nY - n comes from row Y - n
    if (n >= 1 && n < ihalf - 1) return B[(size_t)(Y - n) * X + j];
    return fa_cf{0.f, 0.f};
}
// the same with what a c2r transform keeps of the DC / Nyquist column
__device__ __forceinline__ fa_cf d_fa_crop_sym(const fa_cf *__restrict__ B, int Y, int X, int nY, int nX, int ihalf, int i, int j)
{
    const fa_cf c = d_fa_crop(B, Y, X, nY, ihalf, i, j);
    if (j == 0 || ((nX & 1) == 0 && j == nX / 2)) {
        const fa_cf m = d_fa_crop(B, Y, X, nY, ihalf, (nY - i) % nY, j);
        return fa_cf{0.5f * (c.x + m.x), 0.5f * (c.y - m.y)};
    }
    return c;
}

// full spectrum of the reduced, filtered frame: S(i, j) = crop_sym(i, j) * lpf(i, j) / (Y X) for j <= nX/2, Hermitian beyond
__global__ void __launch_bounds__(256) k_fa_reduce(const fa_cf *__restrict__ B, int Y, int X, fa_cf *__restrict__ S, int nY, int nX,
                                                   const float *__restrict__ lpf, float inorm)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)nY * nX) return;
    B += (size_t)blockIdx.y * Y * X; S += (size_t)blockIdx.y * nY * nX;            // blockIdx.y: frame of a batch
    int i = (int)(t / nX), j = (int)(t - (size_t)i * nX);
    const bool mirror = j > nX / 2;
    if (mirror) { j = nX - j; i = (nY - i) % nY; }
    const int ihalf = min(nY / 2 + 1, Y / 2 + 1);
    fa_cf v = d_fa_crop_sym(B, Y, X, nY, nX, ihalf, i, j);
    const float f = lpf[(size_t)i * (nX / 2 + 1) + j] * inorm;
    v = fa_cf{v.x * f, v.y * f};
    S[t] = mirror ? fa_conj(v) : v;
}

__global__ void __launch_bounds__(256) k_fa_pair(const fa_cf *__restrict__ A, const fa_cf *__restrict__ Bs, fa_cf *__restrict__ P, size_t tot, float scale)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tot) return;
    const fa_cf a = A[t], b = Bs[t];
    // element 0 is the mean of the correlation map, which bestShift subtracts again (statisticsAdjust) and which is orders of magnitude
    // above every other coefficient: left at zero, the inverse transform rounds at the signal's magnitude instead of the mean's
    P[t] = t == 0 ? fa_cf{0.f, 0.f} : fa_cf{(a.x * b.x + a.y * b.y) * scale, (a.y * b.x - a.x * b.y) * scale};
}

// Two real rows per complex row: row 2r in the real part, row 2r+1 in the imaginary part (one transform along x for both)
__global__ void __launch_bounds__(256) k_fa_load2(const float *__restrict__ frame, const float *__restrict__ dark, const float *__restrict__ gain,
                                                  fa_cf *__restrict__ out, int Y, int X)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int Yh = (Y + 1) / 2;
    if (t >= (size_t)Yh * X) return;
    const int r = (int)(t / X), x = (int)(t - (size_t)r * X);
    const size_t s0 = (size_t)(2 * r) * X + x, s1 = s0 + X;
    float v0 = frame[s0], v1 = 0.f;
    if (dark) v0 -= dark[s0];
    if (gain) v0 *= gain[s0];
    if (2 * r + 1 < Y) {
        v1 = frame[s1];
        if (dark) v1 -= dark[s1];
        if (gain) v1 *= gain[s1];
    }
    out[t] = fa_cf{v0, v1};
}

// ... and apart again for the nc columns the reduced frame keeps: with Z = F(a + i b), F(a)[k] = (Z[k] + conj Z[-k]) / 2,
// F(b)[k] = (Z[k] - conj Z[-k]) / 2i. C: [Y][nc]
// (n1 > 0: Z as xh_fft2d_rows_of_real_pairs leaves it, frequency k of a row at n2 (k % n1) + k / n1)
__global__ void __launch_bounds__(256) k_fa_unpack(const fa_cf *__restrict__ Z, fa_cf *__restrict__ C, int Y, int X, int nc, int n1, int n2)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int Yh = (Y + 1) / 2;
    if (t >= (size_t)Yh * nc) return;
    const int r = (int)(t / nc), k = (int)(t - (size_t)r * nc);
    int pk = k, pm = k ? X - k : 0;
    if (n1 > 0) { pk = n2 * (pk % n1) + pk / n1; pm = n2 * (pm % n1) + pm / n1; }
    const fa_cf zk = Z[(size_t)r * X + pk], zm = Z[(size_t)r * X + pm];
    C[(size_t)(2 * r) * nc + k] = fa_cf{0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
    if (2 * r + 1 < Y) C[(size_t)(2 * r + 1) * nc + k] = fa_cf{0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};
}

// Pair correlation inside the window the shift is looked for in only. With c = S_a conj(S_b) the map the full path returns is
// M = dSize sum_k c_k e^{2 pi i k x}; its mean is dSize c_0 and its variance dSize^2 sum_{k != 0} |c_k|^2 (Parseval), so
// statisticsAdjust needs no map, and bestShift looks at the disc of radius maxShift and at the square it grows around the
// maximum: rows / columns -h .. h of the centred map, transformed as two pruned DFTs. First along y:
// U[pair][yy][kx] = sum_ky c[ky][kx] twY[ky][yy], kx <= nX/2; RW rows per thread.
template <int RW>
__global__ void __launch_bounds__(256) k_fa_pairwin_a(const fa_cf *__restrict__ S, int N, int nY, int nX, const fa_cf *__restrict__ twY, int wy,
                                                      fa_cf *__restrict__ U, double *__restrict__ stat)
{
    __shared__ float red[256];
    const int nxh = nX / 2 + 1;
    const int kx = blockIdx.x * 256 + threadIdx.x;
    int a = 0, rem = blockIdx.y;
    while (rem >= N - 1 - a) { rem -= N - 1 - a; ++a; }
    const int b = a + 1 + rem;
    const size_t small = (size_t)nY * nX;
    const fa_cf *Sa = S + (size_t)a * small, *Sb = S + (size_t)b * small;
    const int yy0 = blockIdx.z * RW;
    const bool first = blockIdx.z == 0;
    fa_cf acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = fa_cf{0.f, 0.f};
    float ss = 0.f, c0 = 0.f;
    if (kx < nxh) {
        const float wgt = (kx == 0 || 2 * kx == nX) ? 1.f : 2.f;
        for (int ky = 0; ky < nY; ++ky) {
            const fa_cf p = Sa[(size_t)ky * nX + kx], q = Sb[(size_t)ky * nX + kx];
            float pr = p.x * q.x + p.y * q.y, pi = p.y * q.x - p.x * q.y;
            if (first) {
                if (ky == 0 && kx == 0) c0 = pr;
                else ss += wgt * (pr * pr + pi * pi);
            }
            // The mean of the map (its (0, 0) coefficient) stays out of the window: bestShift subtracts it again (statisticsAdjust), and it
            // is orders of magnitude above everything else -- summed in, every later term is rounded at ITS magnitude (k_fa_pairwin_b)
            if (ky == 0 && kx == 0) { pr = 0.f; pi = 0.f; }
            const fa_cf *w = twY + (size_t)ky * wy + yy0;
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (yy0 + r < wy) { const fa_cf t = w[r]; acc[r].x += pr * t.x - pi * t.y; acc[r].y += pr * t.y + pi * t.x; }
        }
        fa_cf *u = U + ((size_t)blockIdx.y * wy + yy0) * nxh + kx;
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (yy0 + r < wy) u[(size_t)r * nxh] = acc[r];
    }
    if (first) {
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) atomicAdd(&stat[2 * blockIdx.y + 1], (double)red[0]);
        if (kx == 0) stat[2 * blockIdx.y] = (double)c0;
    }
}

// The same, one wave per workgroup, RW rows per thread out of a table padded to whole groups of RW rows (zeros beyond wy): no
// conditions inside the loop, the RW factors of a ky are wave-uniform and arrive by scalar loads, and every complex multiply-add is two
// packed fused multiply-adds (v_pk_fma_f32): (re, im) += (pr, pr) (tx, ty); (re, im) += (-pi, pi) (ty, tx).
template <int RW>
__global__ void __launch_bounds__(64) k_fa_pairwin_a2(const fa_cf *__restrict__ S, int N, int nY, int nX, const fa_v2 *__restrict__ twYp, int wy, int wyp,
                                                      fa_cf *__restrict__ U, double *__restrict__ stat)
{
    const int nxh = nX / 2 + 1;
    const int kx = blockIdx.x * 64 + threadIdx.x;
    int a = 0, rem = blockIdx.y;
    while (rem >= N - 1 - a) { rem -= N - 1 - a; ++a; }
    const int b = a + 1 + rem;
    const size_t small = (size_t)nY * nX;
    const int kxc = min(kx, nxh - 1);
    const fa_cf *Sa = S + (size_t)a * small + kxc, *Sb = S + (size_t)b * small + kxc;
    const int yy0 = blockIdx.z * RW;
    const bool first = blockIdx.z == 0;
    fa_v2 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = fa_v2{0.f, 0.f};
    float ss = 0.f, c0 = 0.f;
    const float wgt = (kx == 0 || 2 * kx == nX) ? 1.f : 2.f;
    const fa_v2 *w = twYp + yy0;
    for (int ky = 0; ky < nY; ++ky, w += wyp) {
        const fa_cf p = Sa[(size_t)ky * nX], q = Sb[(size_t)ky * nX];
        float pr = p.x * q.x + p.y * q.y, pi = p.y * q.x - p.x * q.y;
        if (first) {
            if (ky == 0) c0 = pr;
            else ss += wgt * (pr * pr + pi * pi);
            if (ky == 0 && kx != 0) ss += wgt * (pr * pr + pi * pi);
        }
        if (ky == 0 && kx == 0) { pr = 0.f; pi = 0.f; }     // the map's mean stays out of the window (see k_fa_pairwin_a)
        const fa_v2 vr = fa_v2{pr, pr}, vi = fa_v2{-pi, pi};
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const fa_v2 t = w[r];
            acc[r] = __builtin_elementwise_fma(vr, t, acc[r]);
            acc[r] = __builtin_elementwise_fma(vi, fa_v2{t.y, t.x}, acc[r]);
        }
    }
    if (kx < nxh) {
        fa_cf *u = U + ((size_t)blockIdx.y * wy + yy0) * nxh + kx;
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (yy0 + r < wy) u[(size_t)r * nxh] = fa_cf{acc[r].x, acc[r].y};
    }
    if (first) {
        if (kx >= nxh) ss = 0.f;
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if (threadIdx.x == 0) atomicAdd(&stat[2 * blockIdx.y + 1], (double)ss);
        if (kx == 0) stat[2 * blockIdx.y] = (double)c0;
    }
}

// A DFT of at most 32 outputs per line, lines side by side: out[j][col] = sum_i W[i][j] in[i][col], a lane per column, the JP (= J rounded
// up to a multiple of four, zeros beyond J) factors of an i wave-uniform and fetched by scalar loads, every complex multiply-add two
// packed fused multiply-adds (the pattern of k_fa_pairwin_a2). blockIdx.y / .z: two batch levels (in, out and -- y only -- W move).
// The two steps of the pruned column pass of the frame transform (xh_fa_global_alignment).
template <int JP>
__global__ void __launch_bounds__(64) k_fa_small_dft(const fa_v2 *__restrict__ in, size_t inI, size_t inZ0, size_t inZ1, fa_v2 *__restrict__ out, size_t outJ, size_t outZ0,
                                                     size_t outZ1, const fa_v2 *__restrict__ W, size_t wZ0, int I, int J, int ncols)
{
    const int col = blockIdx.x * 64 + threadIdx.x, colc = min(col, ncols - 1);
    in += inZ0 * blockIdx.y + inZ1 * blockIdx.z + colc;
    W += wZ0 * blockIdx.y;
    fa_v2 acc[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) acc[j] = fa_v2{0.f, 0.f};
#pragma unroll 2
    for (int i = 0; i < I; ++i) {
        const fa_v2 t = in[(size_t)i * inI];
        const fa_v2 vr = fa_v2{t.x, t.x}, vi = fa_v2{-t.y, t.y};
        const fa_v2 *w = W + (size_t)i * JP;
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            const fa_v2 f = w[j];
            acc[j] = __builtin_elementwise_fma(vr, f, acc[j]);
            acc[j] = __builtin_elementwise_fma(vi, fa_v2{f.y, f.x}, acc[j]);
        }
    }
    if (col >= ncols) return;
    out += outZ0 * blockIdx.y + outZ1 * blockIdx.z + col;
#pragma unroll
    for (int j = 0; j < JP; ++j)
        if (j < J) out[(size_t)j * outJ] = acc[j];
}

// sum and sum of squares of the correlation map (real part of the inverse transform), one partial per block
__global__ void __launch_bounds__(256) k_fa_stats(const fa_cf *__restrict__ M, size_t tot, double *__restrict__ part)
{
    __shared__ double s1[256], s2[256];
    double a = 0, b = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (size_t)gridDim.x * 256) { const double v = M[t].x; a += v; b += v * v; }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}

// bestShift (data/filters.cpp:1593-1719, mask == nullptr) on the centred map: statisticsAdjust(0, 1), first maximum in raster order
// within the disc of radius maxShift, neighbourhood growth while every value stays above max / 1.414, centre of mass.
// Logical index l of CenterFFT(R, true) <-> transform index l >= 0 ? l : l + n. One block.
__global__ void __launch_bounds__(256) k_fa_bestshift(const fa_cf *__restrict__ M, int nY, int nX, int maxShift, const double *__restrict__ part, int nparts,
                                                      double *__restrict__ out)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    double sum = 0, sum2 = 0;
    for (int p = 0; p < nparts; ++p) { sum += part[2 * p]; sum2 += part[2 * p + 1]; }
    const double N = (double)nY * (double)nX;
    const double avg = sum / N;
    double sd = sqrt(fabs(sum2 / N - avg * avg));
    double a = 0, b = 0;
    if (sd != 0) { a = 1.0 / sd; b = -avg * a; }
    const int starty = -(nY / 2), startx = -(nX / 2), finy = starty + nY - 1, finx = startx + nX - 1;
    auto val = [&](int i, int j) { return a * (double)M[(size_t)(i >= 0 ? i : i + nY) * nX + (j >= 0 ? j : j + nX)].x + b; };
    // maximum within the disc, first in raster order (i outer, j inner) among equals
    const int w = 2 * maxShift + 1;
    double best = -1.79769313486231570815e+308;
    int bestIdx = 0x7fffffff;
    for (int t = threadIdx.x; t < w * w; t += 256) {
        const int i = t / w - maxShift, j = t % w - maxShift;
        if (i * i + j * j > maxShift * maxShift) continue;
        const double v = val(i, j);
        if (v > best) { best = v; bestIdx = t; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bestIdx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = sv[threadIdx.x + o];
            const int k = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && k < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double shiftX = 0, shiftY = 0;
    if (si[0] == 0x7fffffff) { out[0] = 0; out[1] = 0; out[2] = -1; return; }
    const int imax = si[0] / w - maxShift, jmax = si[0] % w - maxShift;
    const double mx = sv[0];
    int n_max = -1;
    bool neighbourhood = true;
    while (neighbourhood) {
        n_max++;
        for (int i = -n_max; i <= n_max && neighbourhood; i++) {
            const int ia = i + imax;
            if (ia < starty || ia > finy) { neighbourhood = false; break; }
            for (int j = -n_max; j <= n_max && neighbourhood; j++) {
                const int ja = j + jmax;
                if (ja < startx || ja > finx) { neighbourhood = false; break; }
                else if (mx / 1.414 > val(ia, ja)) { neighbourhood = false; break; }
            }
        }
    }
    if (imax - n_max < starty) n_max = min(imax - starty, n_max);
    if (imax + n_max > finy) n_max = min(finy - imax, n_max);
    if (jmax - n_max < starty) n_max = min(jmax - startx, n_max);       // (the reference compares with STARTINGY / FINISHINGY here too)
    if (jmax + n_max > finy) n_max = min(finx - jmax, n_max);
    double xs = 0, ys = 0, sc = 0;
    for (int i = -n_max; i <= n_max; i++)
        for (int j = -n_max; j <= n_max; j++) {
            const int ia = i + imax, ja = j + jmax;
            const double v = val(ia, ja);
            ys += ia * v; xs += ja * v; sc += v;
        }
    if (sc != 0) { shiftX = xs / sc; shiftY = ys / sc; }
    out[0] = shiftX; out[1] = shiftY; out[2] = mx;
}

// ... then along x (real part: the spectra are Hermitian) into the block's window W[wy][wx], and bestShift as k_fa_bestshift does
// it on the full map. out[pair] = (shiftX, shiftY, max, overflow): overflow = 1 when the square grown around the maximum reaches
// beyond the window; the host repeats such a pair through the full transform.
#define XH_FA_PWB_KC 32
__global__ void __launch_bounds__(256) k_fa_pairwin_b(const fa_cf *__restrict__ Uall, const fa_cf *__restrict__ twX, const double *__restrict__ stat, int nY, int nX,
                                                      int hy, int hx, int maxShift, float *__restrict__ Wall, double *__restrict__ out, int ldsRows)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const int nxh = nX / 2 + 1, wy = 2 * hy + 1, wx = 2 * hx + 1;
    const fa_cf *U = Uall + (size_t)blockIdx.x * wy * nxh;
    float *W = Wall + (size_t)blockIdx.x * wy * wx;
    const double dSize = (double)nY * (double)nX;
    if (ldsRows > 0) {
        // a lane per column xx of the window and group of up to 16 rows (256 / wx groups side by side); kx in chunks of XH_FA_PWB_KC whose
        // U values are staged through LDS (coalesced in, broadcast out), the lane's factor twX[kx][xx] loaded once per kx
        extern __shared__ fa_cf sU[];                                                  // [ldsRows][XH_FA_PWB_KC + 1]
        constexpr int KC = XH_FA_PWB_KC, LD = KC + 1;
        const int G = max(1, min(min(256 / wx, wy), ldsRows / 16)), grp = threadIdx.x / wx, xx = threadIdx.x - grp * wx;
        for (int yb = 0; yb < wy; yb += 16 * G) {
            const int yq = yb + grp * 16, nr = min(wy - yb, 16 * G);
            const bool act = grp < G && yq < wy;
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            for (int k0 = 0; k0 < nxh; k0 += KC) {
                __syncthreads();
                for (int i = threadIdx.x; i < nr * KC; i += 256) {
                    const int rr = i / KC, kk = i - rr * KC;
                    sU[rr * LD + kk] = k0 + kk < nxh ? U[(size_t)(yb + rr) * nxh + k0 + kk] : fa_cf{0.f, 0.f};
                }
                __syncthreads();
                if (act) {
                    const int kn = min(KC, nxh - k0);
                    for (int kk = 0; kk < kn; ++kk) {
                        const int kx = k0 + kk;
                        fa_cf t = twX[(size_t)kx * wx + xx];
                        const float wgt = (kx == 0 || 2 * kx == nX) ? 1.f : 2.f;
                        t.x *= wgt; t.y *= -wgt;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const fa_cf v = sU[min(grp * 16 + i, nr - 1) * LD + kk];
                            acc[i] = fmaf(v.x, t.x, acc[i]);
                            acc[i] = fmaf(v.y, t.y, acc[i]);
                        }
                    }
                }
            }
            if (act)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (yq + i < wy) W[(size_t)(yq + i) * wx + xx] = acc[i] * (float)dSize;
        }
    } else
    for (int o = threadIdx.x; o < wy * wx; o += 256) {
        const int yy = o / wx, xx = o - yy * wx;
        const fa_cf *u = U + (size_t)yy * nxh;
        const fa_cf *w = twX + xx;
        float acc = 0.f;
        for (int kx = 0; kx < nxh; ++kx) {
            const fa_cf t = w[(size_t)kx * wx], v = u[kx];
            const float r = v.x * t.x - v.y * t.y;
            acc += (kx == 0 || 2 * kx == nX) ? r : 2.f * r;
        }
        W[o] = acc * (float)dSize;
    }
    __syncthreads();
    // statisticsAdjust(0, 1): (value - mean) / sd.  The window was summed WITHOUT the map's mean (the (0, 0) coefficient was left out by
    // k_fa_pairwin_a / a2; stat[0] still holds it), so there is nothing left to subtract
    const double sd = dSize * sqrt(stat[2 * blockIdx.x + 1]);
    double a = 0;
    if (sd != 0) a = 1.0 / sd;
    const int starty = -(nY / 2), startx = -(nX / 2), finy = starty + nY - 1, finx = startx + nX - 1;
    auto val = [&](int i, int j) { return a * (double)W[(size_t)(i + hy) * wx + (j + hx)]; };
    const int w = 2 * maxShift + 1;
    double best = -1.79769313486231570815e+308;
    int bestIdx = 0x7fffffff;
    for (int t = threadIdx.x; t < w * w; t += 256) {
        const int i = t / w - maxShift, j = t % w - maxShift;
        if (i * i + j * j > maxShift * maxShift) continue;
        const double v = val(i, j);
        if (v > best) { best = v; bestIdx = t; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bestIdx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = sv[threadIdx.x + o];
            const int k = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && k < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double *o4 = out + 4 * (size_t)blockIdx.x;
    if (si[0] == 0x7fffffff) { o4[0] = 0; o4[1] = 0; o4[2] = -1; o4[3] = 0; return; }
    const int imax = si[0] / w - maxShift, jmax = si[0] % w - maxShift;
    const double mx = sv[0];
    bool overflow = false;
    auto inwin = [&](int i, int j) { return i >= -hy && i <= hy && j >= -hx && j <= hx; };
    int n_max = -1;
    bool neighbourhood = true;
    while (neighbourhood) {
        n_max++;
        for (int i = -n_max; i <= n_max && neighbourhood; i++) {
            const int ia = i + imax;
            if (ia < starty || ia > finy) { neighbourhood = false; break; }
            for (int j = -n_max; j <= n_max && neighbourhood; j++) {
                const int ja = j + jmax;
                if (ja < startx || ja > finx) { neighbourhood = false; break; }
                if (!inwin(ia, ja)) { overflow = true; neighbourhood = false; break; }
                if (mx / 1.414 > val(ia, ja)) { neighbourhood = false; break; }
            }
        }
    }
    if (imax - n_max < starty) n_max = min(imax - starty, n_max);
    if (imax + n_max > finy) n_max = min(finy - imax, n_max);
    if (jmax - n_max < starty) n_max = min(jmax - startx, n_max);
    if (jmax + n_max > finy) n_max = min(finx - jmax, n_max);
    double xs = 0, ys = 0, sc = 0;
    for (int i = -n_max; i <= n_max && !overflow; i++)
        for (int j = -n_max; j <= n_max; j++) {
            const int ia = i + imax, ja = j + jmax;
            if (!inwin(ia, ja)) { overflow = true; break; }
            const double v = val(ia, ja);
            ys += ia * v; xs += ja * v; sc += v;
        }
    double shiftX = 0, shiftY = 0;
    if (sc != 0) { shiftX = xs / sc; shiftY = ys / sc; }
    o4[0] = shiftX; o4[1] = shiftY; o4[2] = mx; o4[3] = overflow ? 1.0 : 0.0;
}

// ---- local (patch) alignment: computeLocalAlignment, movie_alignment_correlation_gpu.cpp:288-430 ------------------------------
// One patch position at a time, all N frames of it:
//   k_fa_gather      the patch window of every frame at the frame's rounded global shift, dark / gain applied (getPatchData, :166-202)
//   k_fa_gemm        the part of the patch spectrum the correlation keeps, as two pruned DFTs written as matrix products:
//                    along x for the cxh = C/2+1 kept columns, along y for the C kept rows (performFFTAndScale /
//                    scaleFFT2DKernel, cuda_flexalign_scale.cpp:58-77 + cuda_gpu_movie_alignment_correlation_kernels.cu, take an
//                    FFT of P x P and drop all but C x cxh values: 58 x 114 of 251 x 500 at the defaults)
//   k_fa_patch_sum   the patchesAvg frames around t summed (the transform is linear: getPatchData sums the pixels), low-pass
//   k_fa_patch_corr  per frame pair: S_a conj(S_b) (-1)^(x+y), inverse transform of the window the maximum is searched in only
//                    (2 maxDist + 3 rows and columns about the centre, not all C x C), first maximum within maxDist, centre of
//                    mass of the 3 x 3 values around it (computeCorrelations / sFindMax2DAroundCenter / refineLocation)
__global__ void __launch_bounds__(256) k_fa_gather(const float *__restrict__ frames, const float *__restrict__ dark, const float *__restrict__ gain,
                                                   const int *__restrict__ offs, float *__restrict__ out, int N, int nFrames, int Y, int X, int PY, int PX)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * PY * PX) return;
    const int x = (int)(t % PX), y = (int)((t / PX) % PY), f = (int)(t / ((size_t)PX * PY));          // f: (patch of the batch, frame)
    const size_t src = (size_t)(offs[2 * f + 1] + y) * X + offs[2 * f] + x;
    float v = frames[(size_t)(f % nFrames) * Y * X + src];
    if (dark) v -= dark[src];
    if (gain) v *= gain[src];
    out[t] = v;
}

// C[b][m][n] = sum_k A[b][m][k] B[b][k][n], B and C complex, A real or complex; 64 x 32 tiles of C per block, 4 x 2 per thread
template <bool ACPLX>
__global__ void __launch_bounds__(256) k_fa_gemm(const float *__restrict__ A, size_t lda, size_t sA, const fa_cf *__restrict__ B, size_t ldb, size_t sB,
                                                 fa_cf *__restrict__ C, size_t ldc, size_t sC, int M, int Nc, int K)
{
    constexpr int TM = 64, TN = 32, TK = 16;
    __shared__ fa_cf As[TK][TM + 1];
    __shared__ fa_cf Bs[TK][TN];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    A += sA * blockIdx.z; B += sB * blockIdx.z; C += sC * blockIdx.z;
    fa_cf acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = fa_cf{0.f, 0.f}; acc[i][1] = fa_cf{0.f, 0.f}; }
    for (int k0 = 0; k0 < K; k0 += TK) {
        {
            const int r = t >> 2, kc = (t & 3) * 4, m = m0 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kc + q;
                fa_cf v = fa_cf{0.f, 0.f};
                if (m < M && k < K) {
                    if (ACPLX) v = reinterpret_cast<const fa_cf *>(A)[(size_t)m * lda + k];
                    else v.x = A[(size_t)m * lda + k];
                }
                As[kc + q][r] = v;
            }
            const int kb = t >> 4, nb = (t & 15) * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + kb, n = n0 + nb + q;
                Bs[kb][nb + q] = (k < K && n < Nc) ? B[(size_t)k * ldb + n] : fa_cf{0.f, 0.f};
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            const fa_cf b0 = Bs[kk][tx * 2], b1 = Bs[kk][tx * 2 + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const fa_cf a = As[kk][ty * 4 + i];
                acc[i][0].x += a.x * b0.x; acc[i][0].y += a.x * b0.y;
                acc[i][1].x += a.x * b1.x; acc[i][1].y += a.x * b1.y;
                if (ACPLX) {
                    acc[i][0].x -= a.y * b0.y; acc[i][0].y += a.y * b0.x;
                    acc[i][1].x -= a.y * b1.y; acc[i][1].y += a.y * b1.x;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = n0 + tx * 2 + q;
            if (n < Nc) C[(size_t)m * ldc + n] = acc[i][q];
        }
    }
}

// The same products on the matrix cores (v_mfma_f32_32x32x2_f32, fp32 in, fp32 accumulate): a complex B [K][Nc] is a real matrix of
// 2 Nc columns as it lies in memory, so real A x complex B is one real product; a complex A splits into Ar B + Ai (i B), i.e. a
// real product over 2 K with the rows of i B = (-Bi, Br) formed while B is staged. Block = 128 x 128 of the real C, four waves of
// 64 x 64 (2 x 2 MFMA tiles), K in steps of 16 through LDS.
typedef float fa_f32x16 __attribute__((ext_vector_type(16)));
// GATHER (real A only): A is the movie itself -- row m = (patch frame f = m / PY, patch row y = m % PY) of the product is the window of
// frame f % nFrames at the frame's rounded global shift, (frame - dark) * gain like k_fa_gather, read where it lies: the patch copy
// (k_fa_gather: 4.3 GB written and read back per K3 movie) disappears.
struct FaGather { const float *dark, *gain; const int *offs; int nFrames, Y, X, PY; };
// a second batch level: blockIdx.z = z1 Z0 + z0, matrices at s z0 + s1 z1 (Z0 = 0: one level, blockIdx.z = z0)
struct FaBatch2 { int Z0; size_t sA1, sB1, sC1; };
template <bool ACPLX, bool GATHER = false>
__global__ void __launch_bounds__(256) k_fa_gemm_mfma(const float *__restrict__ A, size_t lda, size_t sA, const fa_cf *__restrict__ B, size_t ldb, size_t sB,
                                                      fa_cf *__restrict__ C, size_t ldc, size_t sC, int M, int Nc, int K, FaGather G = FaGather{}, FaBatch2 Z = FaBatch2{})
{
    constexpr int BM = 128, BN = 128, BK = 16, LD = BM + 4;
    __shared__ float As[BK][LD], Bs[BK][LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;          // n0: real column
    const int N2 = 2 * Nc, K2 = ACPLX ? 2 * K : K;
    const size_t z1 = Z.Z0 ? blockIdx.z / Z.Z0 : 0, z0 = blockIdx.z - z1 * Z.Z0;
    A += (sA * z0 + Z.sA1 * z1) * (ACPLX ? 2 : 1);
    const float *Bf = reinterpret_cast<const float *>(B + sB * z0 + Z.sB1 * z1);
    float *Cf = reinterpret_cast<float *>(C + sC * z0 + Z.sC1 * z1);
    fa_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int arow = t >> 1, akseg = (t & 1) * 8;                  // A tile: 128 rows x 16 k, eight consecutive k per thread
    const int bk = t >> 4, bcseg = (t & 15) * 8;                   // B tile: 16 k x 128 real columns, eight consecutive columns per thread
    size_t gsrc = 0, gfrm = 0;             // GATHER: where row m0 + arow starts in its frame, and that frame
    if (GATHER && m0 + arow < M) {
        const int m = m0 + arow, f = m / G.PY, y = m - f * G.PY;
        gsrc = (size_t)(G.offs[2 * f + 1] + y) * G.X + G.offs[2 * f];
        gfrm = (size_t)(f % G.nFrames) * G.Y * G.X;
    }
    // the tile of step k + 1 is fetched into registers before the products of step k are issued (its latency flies under them), and goes
    // to LDS when they are done
    auto fetch = [&](int k0, float (&av)[8], float (&bv)[8]) {
        const int m = m0 + arow;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kk = k0 + akseg + q;
            float v = 0.f;
            if (m < M && kk < K2) {
                if (GATHER) {
                    v = A[gfrm + gsrc + kk];
                    if (G.dark) v -= G.dark[gsrc + kk];
                    if (G.gain) v *= G.gain[gsrc + kk];
                } else if (!ACPLX) v = A[(size_t)m * lda + kk];
                else v = kk < K ? A[((size_t)m * lda + kk) * 2] : A[((size_t)m * lda + (kk - K)) * 2 + 1];
            }
            av[q] = v;
        }
        const int kk = k0 + bk;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = n0 + bcseg + q;
            float v = 0.f;
            if (kk < K2 && c < N2) {
                if (!ACPLX || kk < K) v = Bf[(size_t)kk * 2 * ldb + c];
                else {
                    const float *br = Bf + (size_t)(kk - K) * 2 * ldb;        // row of i B: (-Bi, Br)
                    v = (c & 1) ? br[c - 1] : -br[c + 1];
                }
            }
            bv[q] = v;
        }
    };
    float av[8], bv[8];
    fetch(0, av, bv);
    for (int k0 = 0; k0 < K2; k0 += BK) {
        __syncthreads();                           // the products of the previous step have read the tile
#pragma unroll
        for (int q = 0; q < 8; ++q) { As[akseg + q][arow] = av[q]; Bs[bk][bcseg + q] = bv[q]; }
        __syncthreads();
        if (k0 + BK < K2) fetch(k0 + BK, av, bv);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5), l31 = lane & 31;
            const float a0 = As[kr][wm * 64 + l31], a1 = As[kr][wm * 64 + 32 + l31];
            const float b0 = Bs[kr][wn * 64 + l31], b1 = Bs[kr][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // D[row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)][col = lane & 31]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + wn * 64 + j * 32 + (lane & 31);
            if (c >= N2) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < M) Cf[(size_t)m * 2 * ldc + c] = acc[i][j][e];
            }
        }
}

// S[t] = filter * sum of the single-frame spectra of the frames t - (avg-1)/2 .. t + avg/2
__global__ void __launch_bounds__(256) k_fa_patch_sum(const fa_cf *__restrict__ single, fa_cf *__restrict__ S, const float *__restrict__ filter, int N, int nFrames, size_t E, int avg)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (size_t)N * E) return;
    const int pt = (int)(g / E), t = pt % nFrames;                      // (patch of the batch, frame)
    const size_t e = g - (size_t)pt * E;
    const fa_cf *sp = single + (size_t)(pt - t) * E;
    float re = 0.f, im = 0.f;
    for (int f = max(0, t - ((avg - 1) / 2)); f <= min(nFrames - 1, t + (avg / 2)); ++f) { const fa_cf v = sp[(size_t)f * E + e]; re += v.x; im += v.y; }
    const float w = filter[e];
    S[g] = fa_cf{re * w, im * w};
}

// One block per frame pair (a, b), a < b in the order (0,1), (0,2) ...: the correlation map in rows y0 .. y0+wy-1, columns
// x0 .. x0+wx-1 (U, W: scratch of the block), its first maximum within maxDist of the centre and the 3 x 3 centre of mass.
// blockDim = the kept columns rounded up to whole waves (at most 256): a thread owns a column kx of the product spectrum and
// eight rows of the window at a time, so the product of a (ky, kx) is formed once per eight outputs and the twiddles
// twY[ky][yy] are the same for the whole wave. out[pair] = (posX, posY) in pixels of the correlation map.
__global__ void __launch_bounds__(256) k_fa_patch_corr(const fa_cf *__restrict__ S, int N, int CY, int CX, const fa_cf *__restrict__ twY, const fa_cf *__restrict__ twX,
                                                       int y0, int wy, int x0, int wx, int maxDist, fa_cf *__restrict__ Uall, float *__restrict__ Wall,
                                                       double *__restrict__ out)
{
    constexpr int RW = 8;
    __shared__ float sv[256];
    __shared__ int si[256];
    const int cxh = CX / 2 + 1, nt = blockDim.x;
    // pair index -> (a, b)
    int a = 0, rem = blockIdx.x;
    while (rem >= N - 1 - a) { rem -= N - 1 - a; ++a; }
    const int b = a + 1 + rem;
    const size_t pairIdx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;                // blockIdx.y: patch of the batch
    const fa_cf *Sp = S + (size_t)blockIdx.y * N * CY * cxh;
    const fa_cf *Sa = Sp + (size_t)a * CY * cxh, *Sb = Sp + (size_t)b * CY * cxh;
    fa_cf *U = Uall + pairIdx * wy * cxh;
    float *W = Wall + pairIdx * wy * wx;
    // along y: U[yy][kx] = sum_ky P[ky][kx] e^{2 pi i ky y / CY}
    for (int kx = threadIdx.x; kx < cxh; kx += nt)
        for (int g = 0; g < wy; g += RW) {
            fa_cf acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = fa_cf{0.f, 0.f};
#pragma unroll 4
            for (int ky = 0; ky < CY; ++ky) {
                const fa_cf p = Sa[(size_t)ky * cxh + kx], q = Sb[(size_t)ky * cxh + kx];
                const float sgn = ((kx + ky) & 1) ? -1.f : 1.f;                      // centres the correlation
                const float pr = (p.x * q.x + p.y * q.y) * sgn, pi = (p.y * q.x - p.x * q.y) * sgn;
                const fa_cf *w = twY + (size_t)ky * wy + g;
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (g + r < wy) { const fa_cf t = w[r]; acc[r].x += pr * t.x - pi * t.y; acc[r].y += pr * t.y + pi * t.x; }
            }
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (g + r < wy) U[(size_t)(g + r) * cxh + kx] = acc[r];
        }
    __syncthreads();
    // along x, real part: what a complex-to-real transform of the half spectrum returns
    for (int o = threadIdx.x; o < wy * wx; o += nt) {
        const int yy = o / wx, xx = o - yy * wx;
        const fa_cf *u = U + (size_t)yy * cxh;
        const fa_cf *w = twX + xx;
        float acc = 0.f;
        for (int kx = 0; kx < cxh; ++kx) {
            const fa_cf t = w[(size_t)kx * wx], v = u[kx];
            const float r = v.x * t.x - v.y * t.y;
            acc += (kx == 0 || 2 * kx == CX) ? r : 2.f * r;
        }
        W[o] = acc;
    }
    __syncthreads();
    const int xHalf = CX / 2, yHalf = CY / 2;
    float best = -3.402823466e+38f;
    int bestIdx = 0x7fffffff;
    for (int o = threadIdx.x; o < wy * wx; o += nt) {
        const int yy = o / wx, xx = o - yy * wx;
        const int ly = y0 + yy - yHalf, lx = x0 + xx - xHalf;
        if (ly * ly + lx * lx > maxDist * maxDist) continue;
        const float v = W[o];
        if (v > best) { best = v; bestIdx = o; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bestIdx;
    for (int o = nt + threadIdx.x; o < 256; o += nt) { sv[o] = -3.402823466e+38f; si[o] = 0x7fffffff; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        for (int t = threadIdx.x; t < o; t += nt) {
            const float v = sv[t + o];
            const int k = si[t + o];
            if (v > sv[t] || (v == sv[t] && k < si[t])) { sv[t] = v; si[t] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double posX = 0, posY = 0;
    if (si[0] != 0x7fffffff) {
        const int refY = y0 + si[0] / wx, refX = x0 + si[0] % wx;
        double refVal = (double)sv[0];
        refVal = (refVal == 0) ? 0 : 1.0 / refVal;
        double sw = 0, slx = 0, sly = 0;
        for (int y = max(0, refY - 1); y <= min(CY - 1, refY + 1); ++y)
            for (int x = max(0, refX - 1); x <= min(CX - 1, refX + 1); ++x) {
                const double rel = (double)W[(size_t)(y - y0) * wx + (x - x0)] * refVal;
                sw += rel; slx += x * rel; sly += y * rel;
            }
        sw = (sw == 0) ? 0 : 1.0 / sw;
        posX = slx * sw; posY = sly * sw;
    }
    out[2 * pairIdx] = posX; out[2 * pairIdx + 1] = posY;
}

// k_fa_patch_corr with the first pruned transform like k_fa_pairwin_a2 (all RW >= wy rows of the window per thread in one sweep over ky, the
// factors of a ky by scalar loads out of a table padded to RW entries per ky, packed multiply-adds) and U in LDS instead of global
// scratch for the second. Dynamic LDS: wy cxh complex values.
template <int RW>
__global__ void __launch_bounds__(256) k_fa_patch_corr2(const fa_cf *__restrict__ S, int N, int CY, int CX, const fa_v2 *__restrict__ twYp, const fa_cf *__restrict__ twX,
                                                        int y0, int wy, int x0, int wx, int maxDist, float *__restrict__ Wall, double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char fa_pc_smem[];
    fa_cf *U = reinterpret_cast<fa_cf *>(fa_pc_smem);
    __shared__ float sv[256];
    __shared__ int si[256];
    const int cxh = CX / 2 + 1, nt = blockDim.x;
    int a = 0, rem = blockIdx.x;
    while (rem >= N - 1 - a) { rem -= N - 1 - a; ++a; }
    const int b = a + 1 + rem;
    const size_t pairIdx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;                // blockIdx.y: patch of the batch
    const fa_cf *Sp = S + (size_t)blockIdx.y * N * CY * cxh;
    float *W = Wall + pairIdx * wy * wx;
    for (int kx0 = 0; kx0 < cxh; kx0 += nt) {
        const int kx = kx0 + threadIdx.x, kxc = min(kx, cxh - 1);
        const fa_cf *Sa = Sp + (size_t)a * CY * cxh + kxc, *Sb = Sp + (size_t)b * CY * cxh + kxc;
        fa_v2 acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = fa_v2{0.f, 0.f};
        float sgn = (kxc & 1) ? -1.f : 1.f;                                         // (-1)^(kx + ky) centres the correlation
        const fa_v2 *w = twYp;
        for (int ky = 0; ky < CY; ++ky, w += RW, sgn = -sgn) {
            const fa_cf p = Sa[(size_t)ky * cxh], q = Sb[(size_t)ky * cxh];
            const float pr = (p.x * q.x + p.y * q.y) * sgn, pi = (p.y * q.x - p.x * q.y) * sgn;
            const fa_v2 vr = fa_v2{pr, pr}, vi = fa_v2{-pi, pi};
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const fa_v2 t = w[r];
                acc[r] = __builtin_elementwise_fma(vr, t, acc[r]);
                acc[r] = __builtin_elementwise_fma(vi, fa_v2{t.y, t.x}, acc[r]);
            }
        }
        if (kx < cxh) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (r < wy) U[(size_t)r * cxh + kx] = fa_cf{acc[r].x, acc[r].y};
        }
    }
    __syncthreads();
    // along x, real part (what a complex-to-real transform of the half spectrum returns): a lane per column xx of the window and group of
    // up to 16 rows -- nt / wx groups side by side --, one sweep over kx in which the lane's factor tabX[kx][xx] is loaded once (the loads do not
    // depend on one another) and U[yy][kx] comes out of LDS for all the lane's rows. (A thread per output with kx innermost waited for a
    // global load 58 times per output: two thirds of this kernel's time.)
    {
        const int G = max(1, min(nt / wx, wy)), grp = threadIdx.x / wx, xx = threadIdx.x - grp * wx;
        for (int yb = 0; yb < wy; yb += 16 * G) {
            const int yq = yb + grp * 16;                                              // this lane's rows: yq .. yq + 15
            const bool act = grp < G && yq < wy;
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            if (act) {
#pragma unroll 2
                for (int kx = 0; kx < cxh; ++kx) {
                    fa_cf t = twX[(size_t)kx * wx + xx];
                    const float wgt = (kx == 0 || 2 * kx == CX) ? 1.f : 2.f;
                    t.x *= wgt; t.y *= -wgt;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const fa_cf v = U[(size_t)min(yq + i, wy - 1) * cxh + kx];
                        acc[i] = fmaf(v.x, t.x, acc[i]);
                        acc[i] = fmaf(v.y, t.y, acc[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (yq + i < wy) W[(yq + i) * wx + xx] = acc[i];
            }
        }
    }
    __syncthreads();
    const int xHalf = CX / 2, yHalf = CY / 2;
    float best = -3.402823466e+38f;
    int bestIdx = 0x7fffffff;
    for (int o = threadIdx.x; o < wy * wx; o += nt) {
        const int yy = o / wx, xx = o - yy * wx;
        const int ly = y0 + yy - yHalf, lx = x0 + xx - xHalf;
        if (ly * ly + lx * lx > maxDist * maxDist) continue;
        const float v = W[o];
        if (v > best) { best = v; bestIdx = o; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bestIdx;
    for (int o = nt + threadIdx.x; o < 256; o += nt) { sv[o] = -3.402823466e+38f; si[o] = 0x7fffffff; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        for (int t = threadIdx.x; t < o; t += nt) {
            const float v = sv[t + o];
            const int k = si[t + o];
            if (v > sv[t] || (v == sv[t] && k < si[t])) { sv[t] = v; si[t] = k; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double posX = 0, posY = 0;
    if (si[0] != 0x7fffffff) {
        const int refY = y0 + si[0] / wx, refX = x0 + si[0] % wx;
        double refVal = (double)sv[0];
        refVal = (refVal == 0) ? 0 : 1.0 / refVal;
        double sw = 0, slx = 0, sly = 0;
        for (int y = max(0, refY - 1); y <= min(CY - 1, refY + 1); ++y)
            for (int x = max(0, refX - 1); x <= min(CX - 1, refX + 1); ++x) {
                const double rel = (double)W[(size_t)(y - y0) * wx + (x - x0)] * refVal;
                sw += rel; slx += x * rel; sly += y * rel;
            }
        sw = (sw == 0) ? 0 : 1.0 / sw;
        posX = slx * sw; posY = sly * sw;
    }
    out[2 * pairIdx] = posX; out[2 * pairIdx + 1] = posY;
}

// ---- B-spline warp: applyBSplineTransform(3, ...) (cuda_gpu_geo_transformer.cpp:186-239) ---------------------------------------
// cubic B-spline prefilter of the (dark / gain corrected) frame as a convolution (xh_bspline.h: exactly the recursion with the
// half-sample mirror, 33 taps in fp32), both passes in one kernel like k_pm_prefilter_fir2d, for frames that are not square:
// a block owns XH_FIR_V rows x 256 columns, filters down the columns (thread <-> column, the 16 either side too, the correction
// applied as the samples are read) into an LDS tile and along the rows out of it. plain += the corrected samples (initialMic).
__global__ void __launch_bounds__(256) k_fa_prefilter(const float *__restrict__ in, const float *__restrict__ dark, const float *__restrict__ gain,
                                                      float *__restrict__ out, float *__restrict__ plain, int Y, int X, int tilesX, XhFir F)
{
    constexpr int TW = 256;
    __shared__ __align__(16) float tile[XH_FIR_V][TW + 2 * XH_FIR_K];
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * TW, y0 = ty * XH_FIR_V;
    for (int xx = threadIdx.x; xx < TW + 2 * XH_FIR_K; xx += 256) {
        const int pu = x0 + xx - XH_FIR_K;
        if (pu >= X + XH_FIR_K) break;                   // beyond the halo of the last, partial tile
        int p = pu;
        while (p < 0 || p >= X) p = p < 0 ? -1 - p : 2 * X - 1 - p;
        float w[XH_FIR_V + 2 * XH_FIR_K];
#pragma unroll
        for (int i = 0; i < XH_FIR_V + 2 * XH_FIR_K; ++i) {
            int q = y0 + i - XH_FIR_K;
            while (q < 0 || q >= Y) q = q < 0 ? -1 - q : 2 * Y - 1 - q;
            const size_t src = (size_t)q * X + p;
            float v = in[src];
            if (dark) v -= dark[src];
            if (gain) v *= gain[src];
            w[i] = v;
        }
        const bool own = plain && pu >= x0 && pu < x0 + TW && pu < X;
#pragma unroll
        for (int o = 0; o < XH_FIR_V; ++o) {
            float acc = F.h[0] * w[o + XH_FIR_K];
#pragma unroll
            for (int j = 1; j <= XH_FIR_K; ++j) acc += F.h[j] * (w[o + XH_FIR_K - j] + w[o + XH_FIR_K + j]);
            tile[o][xx] = acc;
            if (own && y0 + o < Y) plain[(size_t)(y0 + o) * X + pu] += w[o + XH_FIR_K];
        }
    }
    __syncthreads();
    const int r = threadIdx.x / (TW / 8), seg = threadIdx.x - r * (TW / 8);
    const int xo = x0 + seg * 8, y = y0 + r;
    if (y >= Y || xo >= X) return;
    float w[8 + 2 * XH_FIR_K];
    const float4 *t4 = reinterpret_cast<const float4 *>(&tile[r][seg * 8]);
#pragma unroll
    for (int i = 0; i < (8 + 2 * XH_FIR_K) / 4; ++i) {
        const float4 q = t4[i];
        w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
    }
    float *dst = out + (size_t)y * X + xo;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float acc = F.h[0] * w[o + XH_FIR_K];
#pragma unroll
        for (int j = 1; j <= XH_FIR_K; ++j) acc += F.h[j] * (w[o + XH_FIR_K - j] + w[o + XH_FIR_K + j]);
        if (xo + o < X) dst[o] = acc;
    }
}

// The same filter in its recursive form (xh_bspline.h, k_pm_prefilter_rec2d: three operations per output and axis instead of 33, a
// run of 32 outputs entered 14 samples early on either side), for frames: a block owns 32 rows x 224 columns, so that the columns it
// filters -- 14 halo columns either side -- are 252, one per thread; along the rows thread <-> (row, 32 columns) out of an LDS tile
// whose rows lie 324 floats apart (the sixteen lanes of a ds_read_b128 on sixteen rows = all banks); the results go back through the
// tile and leave as whole rows.
#define XH_FA_REC_TW 224
__global__ void __launch_bounds__(256) k_fa_prefilter_rec(const float *__restrict__ in, const float *__restrict__ dark, const float *__restrict__ gain,
                                                          float *__restrict__ out, float *__restrict__ plain, int Y, int X, int tilesX)
{
    constexpr int K = XH_REC_K, V = XH_REC_V, S = XH_REC_S, NW = V + 2 * K, TW = XH_FA_REC_TW;
    static_assert(TW + 2 * K <= 256 && TW % 32 == 0 && S >= TW + 2 * K, "one column per thread");
    __shared__ __align__(16) float tile[V * S];
    const float z = -0.26794919243112270647f, zend = z / (z - 1.f);
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * TW, y0 = ty * V;
    if (x0 + TW + K > X) {                               // a tile the frame does not fill: see k_pm_prefilter_rec2d
        for (int e = tid; e < V * S; e += 256) tile[e] = 0.f;
        __syncthreads();
    }
    auto reflX = [X](int a) { const int b = a < 0 ? -1 - a : (a >= X ? 2 * X - 1 - a : a); return min(max(b, 0), X - 1); };
    auto reflY = [Y](int a) { const int b = a < 0 ? -1 - a : (a >= Y ? 2 * Y - 1 - a : a); return min(max(b, 0), Y - 1); };
    if (tid < TW + 2 * K && x0 + tid - K < X + K) {
        const int pu = x0 + tid - K, p = reflX(pu);
        float w[NW];
        const bool inside = y0 >= K && y0 + V + K <= Y;   // (block-uniform) every row of the window lies inside the frame
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const size_t src = (size_t)(inside ? y0 - K + i : reflY(y0 + i - K)) * X + p;
            float v = in[src];
            if (dark) v -= dark[src];
            if (gain) v *= gain[src];
            w[i] = v;
        }
        if (plain && tid >= K && tid < K + TW && pu < X) {
#pragma unroll
            for (int o = 0; o < V; ++o)
                if (y0 + o < Y) plain[(size_t)(y0 + o) * X + pu] += w[o + K];
        }
#pragma unroll
        for (int i = 1; i < NW; ++i) w[i] = __builtin_fmaf(z, w[i - 1], w[i]);
        float a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            if (i < V + K) tile[(i - K) * S + tid] = 6.f * a;
        }
    }
    __syncthreads();
    const int r = tid & (V - 1), sg = tid >> 5;
    float w[NW];
    if (sg < TW / 32) {
        const float4 *t4 = reinterpret_cast<const float4 *>(&tile[r * S + 32 * sg]);
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const float4 q = t4[i];
            w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
        }
#pragma unroll
        for (int i = 1; i < NW; ++i) w[i] = __builtin_fmaf(z, w[i - 1], w[i]);
        float a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            w[i] = 6.f * a;
        }
    }
    __syncthreads();                                    // every window has been read
    if (sg < TW / 32) {
        float4 *t4 = reinterpret_cast<float4 *>(&tile[r * S + 32 * sg]);
#pragma unroll
        for (int i = 0; i < 8; ++i) t4[i] = make_float4(w[K + 4 * i], w[K + 4 * i + 1], w[K + 4 * i + 2], w[K + 4 * i + 3]);
    }
    __syncthreads();
    const bool vec = (X & 3) == 0;
    for (int e = tid; e < V * (TW / 4); e += 256) {
        const int rr = e / (TW / 4), c4 = e - rr * (TW / 4);
        const int y = y0 + rr, x = x0 + 4 * c4;
        if (y >= Y || x >= X) continue;
        const float4 q = *reinterpret_cast<const float4 *>(&tile[rr * S + 4 * c4]);
        float *dst = out + (size_t)y * X + x;
        if (vec) *reinterpret_cast<float4 *>(dst) = q;
        else { dst[0] = q.x; if (x + 1 < X) dst[1] = q.y; if (x + 2 < X) dst[2] = q.z; if (x + 3 < X) dst[3] = q.w; }
    }
}
// which form the frames' prefilter takes: the recursion (default) or the 33-tap convolution (XH_PREFILTER_FORM=0, A/B)
static void fa_prefilter_launch(hipStream_t stream, const float *in, const float *dark, const float *gain, float *out, float *plain, int Y, int X)
{
    if (xh_prefilter_form() == 1 && Y >= 32 && X >= 32) {
        const int tilesX = (X + XH_FA_REC_TW - 1) / XH_FA_REC_TW, tilesY = (Y + XH_REC_V - 1) / XH_REC_V;
        hipLaunchKernelGGL(k_fa_prefilter_rec, dim3((unsigned)(tilesX * tilesY)), dim3(256), 0, stream, in, dark, gain, out, plain, Y, X, tilesX);
        return;
    }
    const XhFir F = xh_fir_taps();
    const int tilesX = (X + 255) / 256, tilesY = (Y + XH_FIR_V - 1) / XH_FIR_V;
    hipLaunchKernelGGL(k_fa_prefilter, dim3((unsigned)(tilesX * tilesY)), dim3(256), 0, stream, in, dark, gain, out, plain, Y, X, tilesX, F);
}

__device__ __forceinline__ float d_fa_b3(float x)
{
    // bspline03, reconstruction_cuda/cuda_gpu_bilib.cu:16-25
    float a = fabsf(x);
    if (a < 1.f) return a * a * (a - 2.f) * 0.5f + (2.f / 3.f);
    if (a < 2.f) { a -= 2.f; return a * a * a * (-1.f / 6.f); }
    return 0.f;
}

#ifndef XH_FA_B3_TAPS
#define XH_FA_B3_TAPS 1             // d_fa_b3_taps without d_fa_b3's branches (0: d_fa_b3 per weight, A/B)
#endif
// The four weights d_fa_b3(t0 - i), i = 0 .. 3, of a position whose taps start one control point / pixel before it: t0 lies in [1, 2],
// so the arguments fall into (1, 2], (0, 1], (-1, 0], (-2, -1] and each weight's polynomial is known beforehand -- the outer two take
// the cubic tail, the inner two the central piece (at |x| = 1 the two pieces meet: 1/6 from either, to a rounding).  Same expressions
// as d_fa_b3, without its two compares and selects per weight (12 -> 4-5 vector instructions; twenty weights per pixel of the warp).
__device__ __forceinline__ void d_fa_b3_taps(float t0, float (&w)[4])
{
#if XH_FA_B3_TAPS
    float a0 = fabsf(t0) - 2.f, a3 = fabsf(t0 - 3.f) - 2.f;
    const float a1 = fabsf(t0 - 1.f), a2 = fabsf(t0 - 2.f);
    w[0] = a0 * a0 * a0 * (-1.f / 6.f);
    w[1] = a1 * a1 * (a1 - 2.f) * 0.5f + (2.f / 3.f);
    w[2] = a2 * a2 * (a2 - 2.f) * 0.5f + (2.f / 3.f);
    w[3] = a3 * a3 * a3 * (-1.f / 6.f);
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = d_fa_b3(t0 - i);
#endif
}

// applyLocalShiftGeometryKernelMorePixels<T, 3> (cuda_gpu_geo_transformer.cu:193-254): the shift of every pixel from the control
// points (getShiftMorePixels: terms of weight <= 1e-4 are dropped), the frame sampled at (x - shiftX, y - shiftY) with mirrored
// borders (interpolatedElementBSpline2D_Degree3MorePixelsEdge, cuda_gpu_multidim_array.cu:277-334)
__global__ void __launch_bounds__(256) k_fa_warp(const float *__restrict__ coef, const float *__restrict__ cX, const float *__restrict__ cY, int lX, int lY, int lT,
                                                 float hX, float hY, float tPos, int Y, int X, float *__restrict__ out, float *__restrict__ sum)
{
    // the control points of the frames around tPos (at most four layers of lX lY values) in LDS
    extern __shared__ float sc[];
    const int nl = lX * lY, t0 = max((int)tPos - 1, -1), t1 = min((int)tPos + 2, lT - 2), nlay = t1 - t0 + 1;
    float *scX = sc, *scY = sc + nlay * nl;
    for (int i = threadIdx.x; i < nlay * nl; i += 256) { scX[i] = cX[(t0 + 1) * nl + i]; scY[i] = cY[(t0 + 1) * nl + i]; }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= X || y >= Y) return;
    const float delta = 0.0001f;
    const float xPos = x / hX, yPos = y / hY;
    float sx = 0.f, sy = 0.f;
    // the spline weights of the (at most) four control points per axis, evaluated once (the reference's loops evaluate them inside
    // one another: same values, same products bY (bX bT), same cut)
    const int tB = (int)tPos - 1, xB = (int)xPos - 1, yB = (int)yPos - 1;
    const int nT = min((int)tPos + 2, lT - 2) - tB + 1, nX = min((int)xPos + 2, lX - 2) - xB + 1, nY = min((int)yPos + 2, lY - 2) - yB + 1;
    float bT[4], bX[4], bY[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bT[i] = d_fa_b3(tPos - (tB + i)); bX[i] = d_fa_b3(xPos - (xB + i)); bY[i] = d_fa_b3(yPos - (yB + i)); }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (a >= nT) break;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b >= nX) break;
            const float tX = bX[b] * bT[a];
            const int ob = (tB + a - t0) * nl + (xB + b + 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= nY) break;
                const float tmp = bY[c] * tX;
                if (tmp > delta) {
                    const int o = ob + (yB + c + 1) * lX;
                    sx += scX[o] * tmp; sy += scY[o] * tmp;
                }
            }
        }
    }
    const int xc = (int)ceilf(-sx), yc = (int)ceilf(-sy);
    const float xd = 2.f - (sx + xc), yd = 2.f - (sy + yc);
    const int l1 = x + xc - 2, m1 = y + yc - 2;
    float wx[4];
    int lx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int l = l1 + i;
        wx[i] = d_fa_b3(xd - i);
        while (l < 0 || l >= X) l = l < 0 ? -l - 1 : 2 * X - l - 1;
        lx[i] = l;
    }
    float columns = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m1 + i;
        while (m < 0 || m >= Y) m = m < 0 ? -m - 1 : 2 * Y - m - 1;
        const float *ref = coef + (size_t)m * X;
        float rows = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) rows += ref[lx[j]] * wx[j];
        columns += rows * d_fa_b3(yd - i);
    }
    const size_t o = (size_t)y * X + x;
    if (out) out[o] = columns;
    if (sum) sum[o] += columns;
}

// The same for lX, lY >= 4 (every pixel then has four control points per axis in x and y).  What differs is how the 64 terms are fed:
// the control points of the (at most four) layers around tPos lie in LDS as QUADS -- for every layer, control row and first column the
// four neighbours C[x0 .. x0 + 3] as one 16-byte record -- so a (layer, row) pair costs two ds_read_b128 (X and Y field; the lanes of a
// wave share the cell almost always: a broadcast) instead of eight ds_read_b32, and the 1e-4 cut multiplies a dropped term by zero
// instead of branching around it (sx + C * 0 = sx).  Same terms, same order (layer, column, row), same products bY (bX bT).
#define XH_FA_WARP_RG 8
#ifndef XH_FA_WARP_INSIDE
#define XH_FA_WARP_INSIDE 1         // the mirrored-border index loops only in the waves that touch a border (0: in every wave, A/B)
#endif
#ifndef XH_FA_WARP_WAVE
#define XH_FA_WARP_WAVE 1           // the 64 terms of the shift once per wave where its pixels share a control cell (0: per lane always, A/B)
#endif
__global__ void __launch_bounds__(256) k_fa_warp_quads(const float *__restrict__ coef, const float *__restrict__ cX, const float *__restrict__ cY, int lX, int lY, int lT,
                                                       float hX, float hY, float tPos, int Y, int X, float *__restrict__ out, float *__restrict__ sum)
{
    extern __shared__ float4 sq[];
    const int nl = lX * lY, qx = lX - 3, nq = qx * lY;
    const int tB = (int)tPos - 1, t1 = min((int)tPos + 2, lT - 2), nT = t1 - tB + 1;       // tB >= -1: layers tB + 1 .. t1 + 1 of the arrays
    // record i = (layer, control row, first column): (X0, Y0, X1, Y1), (X2, Y2, X3, Y3) -- the X and Y field of a control point side by
    // side, so that (sx, sy) += (CX, CY) * tmp is one packed fused multiply-add
    for (int i = threadIdx.x; i < nT * nq; i += 256) {
        const int a = i / nq, rem = i - a * nq, yy = rem / qx, x0 = rem - yy * qx;
        const float *px = cX + (size_t)(tB + 1 + a) * nl + yy * lX + x0, *py = cY + (size_t)(tB + 1 + a) * nl + yy * lX + x0;
        sq[2 * i] = make_float4(px[0], py[0], px[1], py[1]);
        sq[2 * i + 1] = make_float4(px[2], py[2], px[3], py[3]);
    }
    __syncthreads();
    // XH_FA_WARP_RG groups of four rows per block: the quads are staged once for 64 x 4 XH_FA_WARP_RG pixels (with one group a block
    // spent more time staging them and waiting at the barrier than on its 256 pixels)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    if (x >= X) return;
    for (int rg = 0; rg < XH_FA_WARP_RG; ++rg) {
    const int y = (blockIdx.y * XH_FA_WARP_RG + rg) * 4 + (threadIdx.x >> 6);
    if (y >= Y) break;
    const float delta = 0.0001f;
    const float xPos = x / hX, yPos = y / hY;
    const int xi = (int)xPos, yi = (int)yPos;                 // first control column / row of the pixel: xB + 1, yB + 1
    float bT[4], bX[4], bY[4];
    d_fa_b3_taps(tPos - tB, bT);
    d_fa_b3_taps(xPos - (xi - 1), bX);
    d_fa_b3_taps(yPos - (yi - 1), bY);
    fa_v2 sxy = fa_v2{0.f, 0.f};
    const unsigned long long lanesOn = __builtin_amdgcn_read_exec();
#if XH_FA_WARP_WAVE
    // A wave is 64 neighbouring pixels of one row: they share the frame's layers, the control row and -- unless a control column ends
    // inside them -- the control column, i.e. all 64 control points and the products bY bT; only bX differs from lane to lane, and
    // within a control cell each bX[b] runs monotonically from the first lane to the last.  So the 64 terms are formed ONCE per wave,
    // a lane a term: a term whose weight passes the 1e-4 cut at both ends of the wave passes it on every lane and goes into the sum
    // G[b] of its column (sixteen lanes each, added by shuffles); a term that fails at both ends is dropped; the few that change sides
    // inside the wave are handled lane by lane as before.  A pixel is then sx = sum_b bX[b] G[b] + its share of the mixed terms:
    // 4 packed multiply-adds instead of 64 compares and 64 multiply-adds.  (Same terms and the same cut on every lane; the sums are
    // taken in another order and the weight is bX (bY bT) instead of bY (bX bT): rounding, where the parity bound is 2e-4 of the peak.)
    const int xi0 = __builtin_amdgcn_readfirstlane(xi);
    const bool waveForm = lanesOn == ~0ull && __builtin_amdgcn_ballot_w64(xi != xi0) == 0ull;
    if (waveForm) {
        const int lane = threadIdx.x & 63, tb = lane >> 4, ta = (lane >> 2) & 3, tc = lane & 3;
        float bmin[4], bmax[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bX[b]), 0));
            const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bX[b]), 63));
            bmin[b] = fminf(f0, f1); bmax[b] = fmaxf(f0, f1);
        }
        const float bTa = ta == 0 ? bT[0] : ta == 1 ? bT[1] : ta == 2 ? bT[2] : bT[3];
        const float bYc = tc == 0 ? bY[0] : tc == 1 ? bY[1] : tc == 2 ? bY[2] : bY[3];
        const float lob = tb == 0 ? bmin[0] : tb == 1 ? bmin[1] : tb == 2 ? bmin[2] : bmin[3];
        const float hib = tb == 0 ? bmax[0] : tb == 1 ? bmax[1] : tb == 2 ? bmax[2] : bmax[3];
        const float wyt = bYc * bTa;
        const bool valid = ta < nT;
#ifdef XH_FA_WARP_DEBUG
        const bool kept = false, mixed = valid && wyt * hib > delta;
#else
        const bool kept = valid && wyt * lob > delta, mixed = valid && !kept && wyt * hib > delta;
#endif
        fa_v2 C = fa_v2{0.f, 0.f};
        if (valid) {
            const float *rec = reinterpret_cast<const float *>(sq + 2 * (ta * nq + (yi + tc) * qx + xi0));
            C = fa_v2{rec[2 * tb], rec[2 * tb + 1]};
        }
        float px = kept ? wyt * C.x : 0.f, py = kept ? wyt * C.y : 0.f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), 16 * b));
            const float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py), 16 * b));
            sxy.x = __builtin_fmaf(bX[b], gx, sxy.x);
            sxy.y = __builtin_fmaf(bX[b], gy, sxy.y);
        }
        unsigned long long mm = __builtin_amdgcn_ballot_w64(mixed);
        while (mm) {
            const int t = __builtin_ctzll(mm);
            mm &= mm - 1;
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyt), t));
            // (the term's control point comes out of LDS again, a broadcast: two v_readlane of C.x and C.y here were merged by ROCm 7.2's
            // clang into ONE read of C.x feeding both halves of the packed multiply-add below -- seen in the ISA, and in
            // tests/test_gpu_flexalign.py::test_warp_with_the_usual_control_grid_against_the_oracle as wrong shifts in y)
            const float *rt = reinterpret_cast<const float *>(sq + 2 * (((t >> 2) & 3) * nq + (yi + (t & 3)) * qx + xi0)) + 2 * (t >> 4);
            const float cx = rt[0], cy = rt[1];
            const int bt = t >> 4;
            const float tmp = (bt == 0 ? bX[0] : bt == 1 ? bX[1] : bt == 2 ? bX[2] : bX[3]) * w;
            if (tmp > delta) { sxy.x = __builtin_fmaf(cx, tmp, sxy.x); sxy.y = __builtin_fmaf(cy, tmp, sxy.y); }
        }
    } else
#endif
    // The 1e-4 cut as an execution mask: v_cmpx switches the lanes whose term is dropped off for the one packed multiply-add that follows
    // (compare + select + multiply-add were three vector instructions per term, 192 of a pixel's 540; this is two and a scalar move).
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (a >= nT) break;
        float4 Q0[4], Q1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int o = a * nq + (yi + c) * qx + xi; Q0[c] = sq[2 * o]; Q1[c] = sq[2 * o + 1]; }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float tX = bX[b] * bT[a];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                fa_v2 tt;
                tt.x = bY[c] * tX;                                              // the packed instruction reads the low half for both products (op_sel_hi 0)
                asm("" : "=v"(tt.y));                                          // (the high half is never read: no move to fill it)
                const fa_v2 C = b == 0 ? fa_v2{Q0[c].x, Q0[c].y} : b == 1 ? fa_v2{Q0[c].z, Q0[c].w} : b == 2 ? fa_v2{Q1[c].x, Q1[c].y} : fa_v2{Q1[c].z, Q1[c].w};
                // tmp > 1e-4 (0x38d1b717) ? (sx, sy) += (CX, CY) tmp : nothing -- the CUDA kernel's sx += C * tmp contracts to a fused multiply-add as well
                asm volatile("v_cmpx_lt_f32_e32 vcc, 0x38d1b717, %[t]\n\t"
                             "v_pk_fma_f32 %[acc], %[cc], %[tt], %[acc] op_sel_hi:[1,0,1]\n\t"
                             "s_mov_b64 exec, %[on]"
                             : [acc] "+v"(sxy)
                             : [t] "v"(tt.x), [cc] "v"(C), [tt] "v"(tt), [on] "s"(lanesOn)
                             : "vcc");
            }
        }
    }
    const float sx = sxy.x, sy = sxy.y;
    const int xc = (int)ceilf(-sx), yc = (int)ceilf(-sy);
    const float xd = 2.f - (sx + xc), yd = 2.f - (sy + yc);
    const int l1 = x + xc - 2, m1 = y + yc - 2;
    float wx[4], wy[4];
    int lx[4];
    d_fa_b3_taps(xd, wx);
    d_fa_b3_taps(yd, wy);
    // the mirrored borders concern the waves at the frame's edges: where the sixteen taps of every lane lie inside the frame (one ballot)
    // the indices are l1 + i, m1 + i as they stand
    const bool inside = XH_FA_WARP_INSIDE && __builtin_amdgcn_ballot_w64(l1 < 0 || l1 + 3 >= X || m1 < 0 || m1 + 3 >= Y) == 0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int l = l1 + i;
        if (!inside)
            while (l < 0 || l >= X) l = l < 0 ? -l - 1 : 2 * X - l - 1;
        lx[i] = l;
    }
    float columns = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m1 + i;
        if (!inside)
            while (m < 0 || m >= Y) m = m < 0 ? -m - 1 : 2 * Y - m - 1;
        const float *ref = coef + (size_t)m * X;
        float rows = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) rows += ref[lx[j]] * wx[j];
        columns += rows * wy[i];
    }
    const size_t o = (size_t)y * X + x;
    if (out) out[o] = columns;
    if (sum) sum[o] += columns;
    }
}

// ---- host: EquationSystemSolver::solve + computeAlignment ------------------------------------------------------------------
void mean_stddev(const std::vector<double> &v, double &mean, double &sd)
{
    const size_t n = v.size();
    double s = 0, s2 = 0;
    for (double x : v) { s += x; s2 += x * x; }
    mean = n ? s / n : 0;
    sd = 0;
    if (n > 1) sd = std::sqrt(std::fabs((s2 / n - mean * mean) * ((double)n / (n - 1))));
}

// EquationSystemSolver::solve (eq_system_solver.cpp:35-106) for an observation matrix A0 [rows][cols]: weighted least squares for
// both right-hand sides (normal equations, Gauss-Jordan with partial pivoting), residuals against the row-weighted matrix, rows
// beyond 3 sigma get weight 0, `iterations` rounds. The weights are 0 or 1 and the rows sparse (a run of ones in the alignment,
// 64 spline weights in the fit): the rows are kept in compressed form, the normal equations of ALL rows are formed once, and a
// round subtracts the rejected rows from them.
void fa_solve_system(const std::vector<double> &A0, int rows, int cols, const std::vector<double> &bx, const std::vector<double> &by, int iterations,
                     std::vector<double> &sx, std::vector<double> &sy)
{
    std::vector<int> start(rows + 1, 0), idx;
    std::vector<double> val;
    for (int r = 0; r < rows; ++r) {
        const double *a = &A0[(size_t)r * cols];
        for (int c = 0; c < cols; ++c)
            if (a[c] != 0) { idx.push_back(c); val.push_back(a[c]); }
        start[r + 1] = (int)idx.size();
    }
    const int W = cols + 2;
    auto addRow = [&](std::vector<double> &M, int r, double sign) {
        for (int a = start[r]; a < start[r + 1]; ++a) {
            const double rp = sign * val[a];
            double *Mp = &M[(size_t)idx[a] * W];
            for (int b = a; b < start[r + 1]; ++b) Mp[idx[b]] += rp * val[b];          // upper triangle
            Mp[cols] += rp * bx[r];
            Mp[cols + 1] += rp * by[r];
        }
    };
    std::vector<double> Mall((size_t)cols * W, 0.0), w(rows, 1.0);
    for (int r = 0; r < rows; ++r) addRow(Mall, r, 1.0);
    int it = 0;
    do {
        std::vector<double> M = Mall;
        for (int r = 0; r < rows; ++r)
            if (w[r] == 0.0) addRow(M, r, -1.0);
        for (int p = 0; p < cols; ++p)
            for (int q = 0; q < p; ++q) M[(size_t)p * W + q] = M[(size_t)q * W + p];
        for (int k = 0; k < cols; ++k) {
            int piv = k;
            for (int r = k + 1; r < cols; ++r) if (std::fabs(M[(size_t)r * W + k]) > std::fabs(M[(size_t)piv * W + k])) piv = r;
            if (piv != k) for (int c = 0; c < W; ++c) std::swap(M[(size_t)k * W + c], M[(size_t)piv * W + c]);
            const double d = M[(size_t)k * W + k];
            if (d == 0) continue;
            for (int r = 0; r < cols; ++r) {
                if (r == k) continue;
                const double f = M[(size_t)r * W + k] / d;
                if (f == 0) continue;
                for (int c = k; c < W; ++c) M[(size_t)r * W + c] -= f * M[(size_t)k * W + c];
            }
        }
        sx.assign(cols, 0.0); sy.assign(cols, 0.0);
        for (int k = 0; k < cols; ++k) {
            const double d = M[(size_t)k * W + k];
            if (d != 0) { sx[k] = M[(size_t)k * W + cols] / d; sy[k] = M[(size_t)k * W + cols + 1] / d; }
        }
        std::vector<double> ex(rows), ey(rows);
        for (int r = 0; r < rows; ++r) {
            double px = 0, py = 0;
            if (w[r] != 0.0)                                  // (rows already rejected carry a zeroed A: their residual is b itself)
                for (int a = start[r]; a < start[r + 1]; ++a) { px += val[a] * sx[idx[a]]; py += val[a] * sy[idx[a]]; }
            ex[r] = bx[r] - px; ey[r] = by[r] - py;
        }
        double mean, sdx, sdy;
        mean_stddev(ex, mean, sdx);
        mean_stddev(ey, mean, sdy);
        for (int r = 0; r < rows; ++r)
            if (std::fabs(ex[r]) > 3 * sdx || std::fabs(ey[r]) > 3 * sdy) w[r] = 0.0;
        ++it;
    } while (it < iterations);
}

// computeAlignment (movie_alignment_correlation_base.cpp:399-418): refIn < 0 looks for the reference frame (global alignment),
// otherwise the shifts are taken from frame refIn (patches of the local alignment keep the global reference frame)
void fa_solve(const std::vector<double> &bx, const std::vector<double> &by, int N, int iterations, int refIn, double *shiftX, double *shiftY, int *refFrame)
{
    const int rows = N * (N - 1) / 2, cols = N - 1;
    std::vector<double> A0((size_t)rows * cols, 0.0), sx, sy;
    int idx = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j, ++idx)
            for (int ij = i; ij < j; ++ij) A0[(size_t)idx * cols + ij] = 1;
    fa_solve_system(A0, rows, cols, bx, by, iterations, sx, sy);
    auto total = [&](int iref, int j, double &tx, double &ty) {
        tx = ty = 0;
        if (iref < j) for (int jj = j - 1; jj >= iref; --jj) { tx -= sx[jj]; ty -= sy[jj]; }
        else if (iref > j) for (int jj = j; jj <= iref - 1; ++jj) { tx += sx[jj]; ty += sy[jj]; }
    };
    int best = refIn;
    if (best < 0) {
        double worstEver = std::numeric_limits<double>::max();
        for (int iref = 0; iref < N; ++iref) {
            double worst = -1;
            for (int j = 0; j < N; ++j) {
                double tx, ty;
                total(iref, j, tx, ty);
                if (std::fabs(tx) > worst) worst = std::fabs(tx);         // X only: movie_alignment_correlation_base.cpp:258-261
            }
            if (worst < worstEver) { worstEver = worst; best = iref; }
        }
    }
    *refFrame = best;
    for (int i = 0; i < N; ++i) total(best, i, shiftX[i], shiftY[i]);
}

// createLPF + scaleLPF (movie_alignment_correlation_base.cpp:184-227): a 1-D Gaussian profile of nX samples, looked up by
// |w| nX with linear interpolation, for the half spectrum [nY][nX/2+1] of images sampled at Tsp A/px
std::vector<float> fa_make_lpf(double Tsp, float maxRes, int nX, int nY, double scale)
{
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));
    const int nxh = nX / 2 + 1;
    std::vector<double> prof(nX);
    const double iX = 1 / (double)nX, sigma = (Tsp * c) / maxRes;
    for (int x = 0; x < nX; ++x) { const double w = x * iX; prof[x] = std::exp(-0.5 * (w * w) / (sigma * sigma)); }
    std::vector<float> lpf((size_t)nY * nxh);
    for (int i = 0; i < nY; ++i)
        for (int j = 0; j < nxh; ++j) {
            const double wy = nY <= 1 ? 0.0 : (double)(i <= nY / 2 ? i : i - nY) / nY, wx = (double)(j <= nX / 2 ? j : j - nX) / nX;
            const double x = std::sqrt(wx * wx + wy * wy) * nX;
            const int x0 = (int)std::floor(x), x1 = x0 + 1;
            const double fx = x - x0;
            const double d0 = (x0 < 0 || x0 >= nX) ? 0.0 : prof[x0], d1 = (x1 < 0 || x1 >= nX) ? 0.0 : prof[x1];
            lpf[(size_t)i * nxh + j] = (float)(((1 - fx) * d0 + fx * d1) * scale);
        }
    return lpf;
}
int fa_upload(const void *src, XhBuf &b, size_t bytes, xh_ctx *ctx)
{
    XH_TRY(xh_buf_alloc(ctx, b, bytes));
    XH_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipStreamSynchronize(ctx->stream));          // the source is a host temporary
    return XH_OK;
}
}  // namespace

struct xh_fa {
    xh_ctx *ctx;
    int Y, X, nY, nX;
    float Ts, maxRes;
    double sizeFactor;
    xh_fft2d *rows, *cols, *small;        // rows: (Y+1)/2 packed rows of X points; cols: the nc kept columns of Y points; small: a pair map
    int nc;                               // columns of the frame transform the reduced frame keeps (nX/2 + 1)
    int lastFull = 0;                     // pairs of the last global alignment that went through the full transform
    int use_mfma = 1;                     // the pruned-DFT products of the local alignment on the matrix cores (0: the vector-ALU kernel)
    int use_window = 1;                   // pair correlations inside the search window only (0: every pair through the full transform)
    XhBuf work, spectra, lpf, pair, part, res, warpC;
    std::vector<float> warpCHost;         // the spline coefficients the device holds (uploaded once per set, not once per frame)
    int capFrames;
    // tables and scratch of the two alignment calls stay with the handle (movie after movie: no allocation, no table upload):
    // grow-only buffers by name, the tables re-made only when the parameters they depend on change
    std::map<std::string, XhBuf> cache;
    std::string gKey, lKey, cKey;
    // "pruned_columns": the column pass of the frame transform as two matrix products that compute the kept rows only (below)
    int pruned_cols = 1;
    int rows_kept = 1;                    // the row pass of 45 x 128-point rows by the kernel that writes the kept columns only (0: A/B)
    int pairwin_form = 1;                 // 1: k_fa_pairwin_a2 (packed multiply-adds, scalar-loaded factors); 0: k_fa_pairwin_a
    int cn1 = 0, cn2 = 0, cP = 0;         // Y = cn1 cn2; cP of the cn2 second-step frequencies are kept
    int cForm = 0;                        // 1: both steps by k_fa_small_dft, 2: by k_fa_gemm_mfma
    // "prefilter_ahead": the local alignment ends with the B-spline prefilter of every frame (which does not need the spline it is
    // about to fit on the host), so the device works while the host solves; xh_fa_apply_bspline then finds the coefficients ready
    int prefilter_ahead = 0;
    const float *aheadBase = nullptr, *aheadDark = nullptr, *aheadGain = nullptr;
    int aheadN = 0;
};

// rows per thread of k_fa_pairwin_a2 for a window of wy rows: the group size in {12, 16, 20} that pads wy least, the larger on a tie
static int fa_rw_sel(int wy)
{
    int best = 20, pad = (wy + 19) / 20 * 20;
    for (int rw : {16, 12}) { const int p = (wy + rw - 1) / rw * rw; if (p < pad) { pad = p; best = rw; } }
    return best;
}
static int fa_scratch(xh_fa *h, const char *name, size_t bytes, XhBuf **out)
{
    XhBuf &b = h->cache[name];
    XH_TRY(xh_buf_reserve(h->ctx, b, bytes));
    *out = &b;
    return XH_OK;
}
static int fa_table(xh_fa *h, const char *name, const void *src, size_t bytes, XhBuf **out)
{
    XH_TRY(fa_scratch(h, name, bytes, out));
    XH_HIP(hipMemcpyAsync((*out)->p, src, bytes, hipMemcpyHostToDevice, h->ctx->stream));
    XH_HIP(hipStreamSynchronize(h->ctx->stream));          // the source is a host temporary
    return XH_OK;
}

extern "C" {

int xh_fa_destroy(xh_fa *h)
{
    if (!h) return XH_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->rows) xh_fft2d_destroy(h->rows);
    if (h->cols) xh_fft2d_destroy(h->cols);
    if (h->small) xh_fft2d_destroy(h->small);
    xh_buf_free(h->work); xh_buf_free(h->spectra); xh_buf_free(h->lpf); xh_buf_free(h->pair); xh_buf_free(h->part); xh_buf_free(h->res); xh_buf_free(h->warpC);
    for (auto &kv : h->cache) xh_buf_free(kv.second);
    delete h;
    return XH_OK;
}

int xh_fa_create(xh_ctx *ctx, int32_t Y, int32_t X, float sampling_rate, float max_res_for_correlation, xh_fa **out)
{
    XH_CHECK(ctx && out && Y >= 8 && X >= 8 && sampling_rate > 0 && max_res_for_correlation > 0, XH_ERR_ARG, "xh_fa_create: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    // getC, getTsPrime, getScaleFactor (movie_alignment_correlation_base.cpp:286-314), float like the reference
    const float c = std::sqrt(-1.f / (2.f * std::log(0.5f)));
    const float tsPrime = max_res_for_correlation / (8.f * c);
    const float scale = sampling_rate / tsPrime;
    XH_CHECK(scale < 1, XH_ERR_ARG, "xh_fa_create: the correlation scale factor is bigger than one; for this sampling rate use a maximal resolution of %g or higher "
             "(checkSettings, movie_alignment_correlation_base.cpp:74-79)", (double)(sampling_rate * 8 * c));
    xh_fa *h = new xh_fa;
    h->ctx = ctx; h->Y = Y; h->X = X; h->Ts = sampling_rate; h->maxRes = max_res_for_correlation;
    h->sizeFactor = scale;
    h->nX = (int)(X * h->sizeFactor); h->nY = (int)(Y * h->sizeFactor);       // loadData, movie_alignment_correlation.cpp:101-102
    h->rows = h->cols = h->small = nullptr;
    h->capFrames = 0;
    h->nc = h->nX / 2 + 1;
    int rc = (h->nX >= 4 && h->nY >= 4) ? XH_OK : XH_ERR_ARG;
    if (rc != XH_OK) xh_set_error("xh_fa_create: reduced frames of %d x %d pixels", h->nY, h->nX);
    if (rc == XH_OK) rc = xh_fft2d_create(ctx, (Y + 1) / 2, X, &h->rows);
    if (rc == XH_OK) rc = xh_fft2d_create(ctx, Y, h->nc, &h->cols);
    if (rc == XH_OK) rc = xh_fft2d_create(ctx, h->nY, h->nX, &h->small);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->work, sizeof(fa_cf) * (size_t)Y * X);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->pair, sizeof(fa_cf) * (size_t)h->nY * h->nX);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->part, sizeof(double) * 2 * 256);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->res, sizeof(double) * 4);
    if (rc == XH_OK) {
        const double Tsp = (double)(sampling_rate / (float)h->sizeFactor);       // getPixelResolution (float)
        const std::vector<float> lpf = fa_make_lpf(Tsp, max_res_for_correlation, h->nX, h->nY, 1.0);
        rc = xh_buf_alloc(ctx, h->lpf, sizeof(float) * lpf.size());
        if (rc == XH_OK && hipMemcpy(h->lpf.p, lpf.data(), sizeof(float) * lpf.size(), hipMemcpyHostToDevice) != hipSuccess) rc = XH_ERR_HIP;
    }
    if (rc != XH_OK) { xh_fa_destroy(h); return rc; }
    *out = h;
    return XH_OK;
}

int xh_fa_set_option(xh_fa *h, const char *name, double value)
{
    XH_CHECK(h && name, XH_ERR_ARG, "xh_fa_set_option: bad argument");
    if (!strcmp(name, "window")) h->use_window = value != 0;
    else if (!strcmp(name, "mfma")) h->use_mfma = value != 0;
    else if (!strcmp(name, "pairwin_form")) h->pairwin_form = (int)value;
    else if (!strcmp(name, "rows_kept")) h->rows_kept = value != 0;
    else if (!strcmp(name, "pruned_columns")) { h->pruned_cols = (int)value; h->cKey.clear(); }
    else if (!strcmp(name, "prefilter_ahead")) { h->prefilter_ahead = value != 0; h->aheadBase = nullptr; }
    else { xh_set_error("xh_fa_set_option: unknown option %s", name); return XH_ERR_ARG; }
    return XH_OK;
}

int xh_fa_last_full_pairs(const xh_fa *h) { return h ? h->lastFull : -1; }

int xh_fa_info(const xh_fa *h, int32_t *newY, int32_t *newX, double *size_factor)
{
    XH_CHECK(h, XH_ERR_ARG, "xh_fa_info: null handle");
    if (newY) *newY = h->nY;
    if (newX) *newX = h->nX;
    if (size_factor) *size_factor = h->sizeFactor;
    return XH_OK;
}

int xh_fa_global_alignment(xh_fa *h, const float *d_frames, int32_t N, const float *d_dark, const float *d_gain, float max_shift_px,
                           double *h_bX, double *h_bY, double *h_shiftX, double *h_shiftY, int32_t *h_ref)
{
    XH_CHECK(h && d_frames && N >= 2 && h_shiftX && h_shiftY && h_ref, XH_ERR_ARG, "xh_fa_global_alignment: bad argument");
    h->aheadBase = nullptr;                  // a new movie: coefficients prefiltered ahead for the previous one are void
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int nY = h->nY, nX = h->nX;
    const int ms = (int)(max_shift_px * h->sizeFactor);                          // computeShifts, movie_alignment_correlation.cpp:141
    XH_CHECK(ms >= 0 && ms < nY / 2 && ms < nX / 2, XH_ERR_ARG, "xh_fa_global_alignment: --maxShift %g px reaches beyond the reduced frame (%d x %d)", (double)max_shift_px, nY, nX);
    const size_t big = (size_t)h->Y * h->X, small = (size_t)nY * nX;
    XH_TRY(xh_buf_reserve(ctx, h->spectra, sizeof(fa_cf) * small * (size_t)N));
    fa_cf *S = (fa_cf *)h->spectra.p, *wk = (fa_cf *)h->work.p, *P = (fa_cf *)h->pair.p;
    const float inorm = (float)(1.0 / ((double)h->Y * (double)h->X));
    // frame transform: two rows per complex row along x, the nc kept columns apart again and down y on their own
    // ((Y+1)/2 X + Y nc complex values: both fit the Y X of `work` because nc <= X / 2)
    const int Yh = (h->Y + 1) / 2, nc = h->nc;
    XH_CHECK((size_t)Yh * h->X + (size_t)h->Y * nc <= big, XH_ERR_ARG, "xh_fa_global_alignment: frames of %d x %d are too small", h->Y, h->X);
    fa_cf *Cc = wk + (size_t)Yh * h->X;
    // Column pass. The reduced frame keeps the nY lowest of the Y frequencies along y (scaleToSizeFourier), so the columns are not
    // transformed in full: Y = n1 n2, y = b + n2 a, k = k1 + n1 k2,
    //     T[k1][b]  = sum_a  W_n1^(a k1) C[b + n2 a]                             every k1, every b
    //     F[k]      = sum_b  W_Y^(b k1) W_n2^(b k2) T[k1][b]                     the k2 whose k is kept only
    // both as complex matrix products on the matrix cores (k_fa_gemm_mfma), the kept k2 being the first Plow and the last Phigh, so
    // that the rows of the result are a spectrum of P n1 rows in the usual order and k_fa_reduce reads it like the full one.
    // n1: the largest divisor of Y that fits one 128-row tile. (4092 = 124 x 33 keeps 8 of 33: 0.36 G complex multiply-adds per K3
    // frame against two Bluestein line passes + twiddle + transpose over all 4092 rows.)
    bool colsPruned = false;
    int cn1 = 0, cn2 = 0, cP = 0;
    XhBuf *pW1 = nullptr, *pW2 = nullptr, *pT = nullptr, *pR = nullptr, *pC = nullptr;
    int FB = 1;
    if (h->pruned_cols) {
        char key[64];
        snprintf(key, sizeof(key), "%d %d", h->Y, nY);
        if (h->cKey != key) {
            const int ihalf = std::min(nY / 2 + 1, h->Y / 2 + 1);             // d_fa_crop: rows 0 .. ihalf-1 and Y-(ihalf-2) .. Y-1
            auto kept = [&](int n1, int &Plow, int &Phigh) {
                const int n2 = h->Y / n1;
                Plow = (ihalf + n1 - 1) / n1; Phigh = (std::max(ihalf - 2, 0) + n1 - 1) / n1;
                if (Plow + Phigh >= n2) { Plow = n2; Phigh = 0; }
                return Plow + Phigh;
            };
            // the two steps on the vector ALUs (k_fa_small_dft): n1 <= 32 outputs in the first, P <= 32 in the second, least n1 + P
            int n1 = 0, form = 0, lo = 0, hi = 0;
            if (h->pruned_cols == 1)
                for (int d = 2; d <= 32; ++d) {
                    if (h->Y % d) continue;
                    const int P = kept(d, lo, hi);
                    if (P > 32) continue;
                    int l2, h2;
                    if (!n1 || d + P < n1 + kept(n1, l2, h2)) { n1 = d; form = 1; }
                }
            // or as matrix products (k_fa_gemm_mfma): n1 the largest divisor of Y that fits one 128-row tile
            if (!n1)
                for (int d = 128; d >= 8; --d) if (h->Y % d == 0) { n1 = d; form = 2; break; }
            h->cn1 = 0;
            if (n1) {
                const int n2 = h->Y / n1;
                int Plow, Phigh;
                const int P = kept(n1, Plow, Phigh);
                const long double twoPi = 6.28318530717958647692528676655900577L;
                // form 2: W1 [k1][a], W2 [k1][q][b]; form 1: W1 [a][k1 padded to a multiple of 4], W2 [k1][b][q padded]
                const int J1 = form == 1 ? (n1 + 3) / 4 * 4 : n1, J2 = form == 1 ? (P + 3) / 4 * 4 : P;
                std::vector<fa_cf> W1((size_t)n1 * J1, fa_cf{0.f, 0.f}), W2((size_t)n1 * J2 * n2, fa_cf{0.f, 0.f});
                for (int k1 = 0; k1 < n1; ++k1)
                    for (int a = 0; a < n1; ++a) {
                        const long double ang = -twoPi * (long double)(((long long)k1 * a) % n1) / n1;
                        W1[form == 1 ? (size_t)a * J1 + k1 : (size_t)k1 * n1 + a] = fa_cf{(float)cosl(ang), (float)sinl(ang)};
                    }
                for (int k1 = 0; k1 < n1; ++k1)
                    for (int q = 0; q < P; ++q) {
                        const int k2 = q < Plow ? q : n2 - Phigh + (q - Plow);
                        const long long k = k1 + (long long)n1 * k2;
                        for (int b = 0; b < n2; ++b) {
                            const long double ang = -twoPi * (long double)((k * b) % h->Y) / h->Y;      // W_Y^(b k1) W_n2^(b k2) = W_Y^(b k)
                            W2[form == 1 ? ((size_t)k1 * n2 + b) * J2 + q : ((size_t)k1 * P + q) * n2 + b] = fa_cf{(float)cosl(ang), (float)sinl(ang)};
                        }
                    }
                XhBuf *t1b = nullptr, *t2b = nullptr;
                XH_TRY(fa_table(h, "c_W1", W1.data(), sizeof(fa_cf) * W1.size(), &t1b));
                XH_TRY(fa_table(h, "c_W2", W2.data(), sizeof(fa_cf) * W2.size(), &t2b));
                h->cn1 = n1; h->cn2 = n2; h->cP = P; h->cForm = form;
            }
            h->cKey = key;
        }
        if (h->cn1) {
            cn1 = h->cn1; cn2 = h->cn2; cP = h->cP;
            XH_TRY(fa_scratch(h, "c_W1", 0, &pW1));
            XH_TRY(fa_scratch(h, "c_W2", 0, &pW2));
            // FB frames per pair of products (one frame's 363 + 1364 workgroups of a few steps each leave the device waiting)
            FB = std::min(N, 10);
            XH_TRY(fa_scratch(h, "c_C", sizeof(fa_cf) * (size_t)FB * h->Y * nc, &pC));
            XH_TRY(fa_scratch(h, "c_T", sizeof(fa_cf) * (size_t)FB * h->Y * nc, &pT));
            XH_TRY(fa_scratch(h, "c_R", sizeof(fa_cf) * (size_t)FB * cP * cn1 * nc, &pR));
            colsPruned = true;
        }
    }
    for (int n = 0; n < N; ++n) {
        const size_t tz = (size_t)Yh * h->X, tc = (size_t)Yh * nc;
        // rows: from the real frame straight into the first line pass, the result left where the third step puts it (a K3 frame's
        // 5760 = 45 x 128: two trips through memory instead of six); any other length: pack, transform, read in natural order
        fa_cf *Cn = colsPruned ? (fa_cf *)pC->p + (size_t)(n % FB) * h->Y * nc : Cc;
        int kept = 0;
        // (a K3 frame's rows: one kernel that writes the kept columns only)
        if (h->rows_kept) XH_TRY(xh_fft2d_rows_of_real_pairs_kept(h->rows, d_frames + (size_t)n * big, d_dark, d_gain, h->Y, nc, (float *)Cn, &kept));
        if (!kept) {
            int t1 = 0, t2 = 0;
            XH_TRY(xh_fft2d_rows_of_real_pairs(h->rows, d_frames + (size_t)n * big, d_dark, d_gain, h->Y, (float *)wk, &t1, &t2));
            if (t1 == 0) {
                hipLaunchKernelGGL(k_fa_load2, dim3((unsigned)((tz + 255) / 256)), dim3(256), 0, ctx->stream, d_frames + (size_t)n * big, d_dark, d_gain, wk, h->Y, h->X);
                XH_LAUNCH_CHECK();
                XH_TRY(xh_fft2d_exec_axis(h->rows, (float *)wk, 0, 0));
            }
            hipLaunchKernelGGL(k_fa_unpack, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)wk, Cn, h->Y, h->X, nc, t1, t2);
            XH_LAUNCH_CHECK();
        }
        if (colsPruned) {
            if ((n + 1) % FB != 0 && n + 1 != N) continue;
            const int nf = n % FB + 1, f0 = n - (nf - 1);                     // frames f0 .. n lie in c_C
            const size_t fr = (size_t)h->Y * nc;
            if (h->cForm == 1) {
                const dim3 g1((nc + 63) / 64, cn2, nf), g2((nc + 63) / 64, cn1, nf);
                const int J1 = (cn1 + 3) / 4 * 4, J2 = (cP + 3) / 4 * 4;
                // T[f][k1 n2 + b][kx] = sum_a W1[a][k1] C[f][b + n2 a][kx]: batch (b, f), lines n2 nc apart in and out
#define XH_FA_SD1(JP_) hipLaunchKernelGGL((k_fa_small_dft<JP_>), g1, dim3(64), 0, ctx->stream, (const fa_v2 *)pC->p, (size_t)cn2 * nc, (size_t)nc, fr, (fa_v2 *)pT->p, \
                                          (size_t)cn2 * nc, (size_t)nc, fr, (const fa_v2 *)pW1->p, (size_t)0, cn1, cn1, nc)
                // R[f][q n1 + k1][kx] = sum_b W2[k1][b][q] T[f][k1 n2 + b][kx]: batch (k1, f)
#define XH_FA_SD2(JP_) hipLaunchKernelGGL((k_fa_small_dft<JP_>), g2, dim3(64), 0, ctx->stream, (const fa_v2 *)pT->p, (size_t)nc, (size_t)cn2 * nc, fr, (fa_v2 *)pR->p, \
                                          (size_t)cn1 * nc, (size_t)nc, (size_t)cP * cn1 * nc, (const fa_v2 *)pW2->p, (size_t)cn2 * J2, cn2, cP, nc)
#define XH_FA_SDSW(M_, J_) switch ((J_) / 4) { case 1: M_(4); break; case 2: M_(8); break; case 3: M_(12); break; case 4: M_(16); break; case 5: M_(20); break; \
                                               case 6: M_(24); break; case 7: M_(28); break; default: M_(32); break; }
                XH_FA_SDSW(XH_FA_SD1, J1)
                XH_FA_SDSW(XH_FA_SD2, J2)
#undef XH_FA_SD1
#undef XH_FA_SD2
#undef XH_FA_SDSW
            } else {
                // T[f][k1 n2 + b][kx] = sum_a W[k1][a] C[f][b + n2 a][kx], one product per (f, b)
                hipLaunchKernelGGL((k_fa_gemm_mfma<true>), dim3((2 * nc + 127) / 128, (cn1 + 127) / 128, cn2 * nf), dim3(256), 0, ctx->stream, (const float *)pW1->p, (size_t)cn1,
                                   (size_t)0, (const fa_cf *)pC->p, (size_t)cn2 * nc, (size_t)nc, (fa_cf *)pT->p, (size_t)cn2 * nc, (size_t)nc, cn1, nc, cn1, FaGather{},
                                   FaBatch2{cn2, (size_t)0, fr, fr});
                // R[f][q n1 + k1][kx] = sum_b A2[k1][q][b] T[f][k1 n2 + b][kx], one product per (f, k1)
                hipLaunchKernelGGL((k_fa_gemm_mfma<true>), dim3((2 * nc + 127) / 128, (cP + 127) / 128, cn1 * nf), dim3(256), 0, ctx->stream, (const float *)pW2->p, (size_t)cn2,
                                   (size_t)cP * cn2, (const fa_cf *)pT->p, (size_t)nc, (size_t)cn2 * nc, (fa_cf *)pR->p, (size_t)cn1 * nc, (size_t)nc, cP, nc, cn2, FaGather{},
                                   FaBatch2{cn1, (size_t)0, fr, (size_t)cP * cn1 * nc});
            }
            hipLaunchKernelGGL(k_fa_reduce, dim3((unsigned)((small + 255) / 256), nf), dim3(256), 0, ctx->stream, (const fa_cf *)pR->p, cP * cn1, nc, S + (size_t)f0 * small, nY,
                               nX, (const float *)h->lpf.p, inorm);
            XH_LAUNCH_CHECK();
            continue;
        }
        XH_TRY(xh_fft2d_exec_axis(h->cols, (float *)Cc, 0, 1));
        hipLaunchKernelGGL(k_fa_reduce, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)Cc, h->Y, nc, S + (size_t)n * small, nY, nX,
                           (const float *)h->lpf.p, inorm);
        XH_LAUNCH_CHECK();
    }
    const int rows = N * (N - 1) / 2;
    std::vector<double> bx(rows), by(rows);
    XhBuf *pResAll = nullptr;
    XH_TRY(fa_scratch(h, "g_res", sizeof(double) * 3 * (size_t)rows, &pResAll));
    XhBuf &resAll = *pResAll;
    const double dSize = (double)small;
    const int nparts = 256;
    int rc = XH_OK;
    // the full path of one pair: FFT1 conj(FFT2) dSize through an un-normalised inverse (correlation_matrix); ours divides by dSize
    auto pairFull = [&](int i, int j, int idx) {
        hipLaunchKernelGGL(k_fa_pair, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)(S + (size_t)i * small),
                           (const fa_cf *)(S + (size_t)j * small), P, small, (float)(dSize * dSize));
        int r2 = xh_fft2d_exec(h->small, (float *)P, 1);
        if (r2 != XH_OK) return r2;
        hipLaunchKernelGGL(k_fa_stats, dim3(nparts), dim3(256), 0, ctx->stream, (const fa_cf *)P, small, (double *)h->part.p);
        hipLaunchKernelGGL(k_fa_bestshift, dim3(1), dim3(256), 0, ctx->stream, (const fa_cf *)P, nY, nX, ms, (const double *)h->part.p, nparts,
                           (double *)resAll.p + 3 * (size_t)idx);
        if (hipGetLastError() != hipSuccess) { xh_set_error("xh_fa_global_alignment: kernel launch failed"); return (int)XH_ERR_HIP; }
        return (int)XH_OK;
    };
    std::vector<double> res(3 * (size_t)rows);
    const int G = 8;                                  // rows / columns beyond the search disc for the square bestShift grows
    const int hy = ms + G, hx = ms + G;
    const bool windowed = h->use_window && hy < nY / 2 - 1 && hx < nX / 2 - 1;
    if (windowed) {
        const int wy = 2 * hy + 1, wx = 2 * hx + 1, nxh = nX / 2 + 1;
        XhBuf *pTwY = nullptr, *pTwX = nullptr, *pU = nullptr, *pW = nullptr, *pStat = nullptr, *pOut = nullptr;
        char key[96];
        snprintf(key, sizeof(key), "%d %d %d %d", nY, nX, hy, hx);
        if (h->gKey != key) {
            std::vector<fa_cf> twY((size_t)nY * wy), twX((size_t)nxh * wx);
            const double twoPi = 6.283185307179586476925286766559;
            for (int ky = 0; ky < nY; ++ky)
                for (int yy = 0; yy < wy; ++yy) {
                    const long long m = (((long long)ky * (yy - hy)) % nY + nY) % nY;
                    twY[(size_t)ky * wy + yy] = fa_cf{(float)std::cos(twoPi * m / nY), (float)std::sin(twoPi * m / nY)};
                }
            for (int kx = 0; kx < nxh; ++kx)
                for (int xx = 0; xx < wx; ++xx) {
                    const long long m = (((long long)kx * (xx - hx)) % nX + nX) % nX;
                    twX[(size_t)kx * wx + xx] = fa_cf{(float)std::cos(twoPi * m / nX), (float)std::sin(twoPi * m / nX)};
                }
            rc = fa_table(h, "g_twY", twY.data(), sizeof(fa_cf) * twY.size(), &pTwY);
            {
                // the same factors in rows of wyp = whole groups of rwSel(wy) entries, zeros beyond wy (k_fa_pairwin_a2)
                const int rw = fa_rw_sel(wy), wyp = (wy + rw - 1) / rw * rw;
                std::vector<fa_cf> twYp((size_t)nY * wyp, fa_cf{0.f, 0.f});
                for (int ky = 0; ky < nY; ++ky)
                    for (int yy = 0; yy < wy; ++yy) twYp[(size_t)ky * wyp + yy] = twY[(size_t)ky * wy + yy];
                XhBuf *pP = nullptr;
                if (rc == XH_OK) rc = fa_table(h, "g_twYp", twYp.data(), sizeof(fa_cf) * twYp.size(), &pP);
            }
            if (rc == XH_OK) rc = fa_table(h, "g_twX", twX.data(), sizeof(fa_cf) * twX.size(), &pTwX);
            if (rc == XH_OK) h->gKey = key;
        }
        if (rc == XH_OK) rc = fa_scratch(h, "g_twY", 0, &pTwY);
        if (rc == XH_OK) rc = fa_scratch(h, "g_twX", 0, &pTwX);
        if (rc == XH_OK) rc = fa_scratch(h, "g_U", sizeof(fa_cf) * (size_t)rows * wy * nxh, &pU);
        if (rc == XH_OK) rc = fa_scratch(h, "g_W", sizeof(float) * (size_t)rows * wy * wx, &pW);
        if (rc == XH_OK) rc = fa_scratch(h, "g_stat", sizeof(double) * 2 * (size_t)rows, &pStat);
        if (rc == XH_OK) rc = fa_scratch(h, "g_out", sizeof(double) * 4 * (size_t)rows, &pOut);
        auto freeWin = [&]() {};
        if (rc != XH_OK) return rc;
        XhBuf &bTwY = *pTwY, &bTwX = *pTwX, &bU = *pU, &bW = *pW, &bStat = *pStat, &bOut = *pOut;
        if (hipMemsetAsync(bStat.p, 0, sizeof(double) * 2 * (size_t)rows, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        std::vector<double> out4(4 * (size_t)rows);
        XhBuf *pTwYp = nullptr;
        if (rc == XH_OK) rc = fa_scratch(h, "g_twYp", 0, &pTwYp);
        if (rc == XH_OK) {
            const int rw = fa_rw_sel(wy), nz = (wy + rw - 1) / rw, wyp = nz * rw;
            if (h->pairwin_form == 0) {
                constexpr int RW = 16;
                hipLaunchKernelGGL((k_fa_pairwin_a<RW>), dim3((nxh + 255) / 256, rows, (wy + RW - 1) / RW), dim3(256), 0, ctx->stream, (const fa_cf *)S, N, nY, nX,
                                   (const fa_cf *)bTwY.p, wy, (fa_cf *)bU.p, (double *)bStat.p);
            } else if (rw == 20)
                hipLaunchKernelGGL((k_fa_pairwin_a2<20>), dim3((nxh + 63) / 64, rows, nz), dim3(64), 0, ctx->stream, (const fa_cf *)S, N, nY, nX, (const fa_v2 *)pTwYp->p, wy, wyp,
                                   (fa_cf *)bU.p, (double *)bStat.p);
            else if (rw == 16)
                hipLaunchKernelGGL((k_fa_pairwin_a2<16>), dim3((nxh + 63) / 64, rows, nz), dim3(64), 0, ctx->stream, (const fa_cf *)S, N, nY, nX, (const fa_v2 *)pTwYp->p, wy, wyp,
                                   (fa_cf *)bU.p, (double *)bStat.p);
            else
                hipLaunchKernelGGL((k_fa_pairwin_a2<12>), dim3((nxh + 63) / 64, rows, nz), dim3(64), 0, ctx->stream, (const fa_cf *)S, N, nY, nX, (const fa_v2 *)pTwYp->p, wy, wyp,
                                   (fa_cf *)bU.p, (double *)bStat.p);
            // rows of U staged per chunk of kx: all of the window's when 256 / wx groups of 16 cover them, at most 96 (25 KB); windows wider
            // than 256 columns (or the A/B form) keep the thread-per-output loop
            const int ldsRows = (wx <= 256 && h->pairwin_form != 0) ? std::min(96, (std::min(256 / wx, (wy + 15) / 16)) * 16) : 0;
            hipLaunchKernelGGL(k_fa_pairwin_b, dim3(rows), dim3(256), sizeof(fa_cf) * (size_t)ldsRows * (XH_FA_PWB_KC + 1), ctx->stream, (const fa_cf *)bU.p, (const fa_cf *)bTwX.p,
                               (const double *)bStat.p, nY, nX, hy, hx, ms, (float *)bW.p, (double *)bOut.p, ldsRows);
            if (hipGetLastError() != hipSuccess) rc = XH_ERR_HIP;
        }
        if (rc == XH_OK && hipMemcpyAsync(out4.data(), bOut.p, sizeof(double) * out4.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        freeWin();
        // pairs whose maximum is so wide that the square grown around it leaves the window: through the full transform
        std::vector<int> redo;
        int idx = 0;
        for (int i = 0; i < N - 1 && rc == XH_OK; ++i)
            for (int j = i + 1; j < N && rc == XH_OK; ++j, ++idx) {
                if (out4[4 * (size_t)idx + 3] != 0) { redo.push_back(idx); rc = pairFull(i, j, idx); }
                else for (int c = 0; c < 3; ++c) res[3 * (size_t)idx + c] = out4[4 * (size_t)idx + c];
            }
        h->lastFull = (int)redo.size();
        if (rc == XH_OK && !redo.empty()) {
            std::vector<double> full(3 * (size_t)rows);
            if (hipMemcpyAsync(full.data(), resAll.p, sizeof(double) * full.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
            if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
            for (int q : redo) for (int c = 0; c < 3; ++c) res[3 * (size_t)q + c] = full[3 * (size_t)q + c];
        }
    } else {
        int idx = 0;
        for (int i = 0; i < N - 1 && rc == XH_OK; ++i)
            for (int j = i + 1; j < N && rc == XH_OK; ++j, ++idx) rc = pairFull(i, j, idx);
        h->lastFull = rows;
        if (rc == XH_OK && hipMemcpyAsync(res.data(), resAll.p, sizeof(double) * res.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
        if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    }
    if (rc != XH_OK) { if (rc == XH_ERR_HIP) xh_set_error("xh_fa_global_alignment: device error"); return rc; }
    for (int r = 0; r < rows; ++r) { bx[r] = res[3 * r] / h->sizeFactor; by[r] = res[3 * r + 1] / h->sizeFactor; }       // scale to the movie's pixels
    if (h_bX) std::memcpy(h_bX, bx.data(), sizeof(double) * rows);
    if (h_bY) std::memcpy(h_bY, by.data(), sizeof(double) * rows);
    int ref = 0;
    fa_solve(bx, by, N, 2, -1, h_shiftX, h_shiftY, &ref);          // solverIterations = 2 (movie_alignment_correlation_base.h:332)
    *h_ref = ref;
    return XH_OK;
}

}  // extern "C"

namespace {
double fa_b3(double x)
{
    x = std::fabs(x);
    if (x < 1) return (x * x * (x - 2) * 3 + 4) * (1.0 / 6.0);
    if (x < 2) { x -= 2; return x * x * x * (-1.0 / 6.0); }
    return 0;
}

// getMovieBorders + getPatchesLocation (movie_alignment_correlation_gpu.cpp:204-222, 139-164): top-left corner [p][2] and centre
// [p][2] of every patch, p = py patchesX + px
void fa_patch_layout(int N, int Y, int X, const double *gx, const double *gy, int patchesX, int patchesY, int PX, int PY, std::vector<double> &tl,
                     double *centers)
{
    double minX = 1e300, maxX = -1e300, minY = 1e300, maxY = -1e300;
    for (int i = 0; i < N; ++i) {
        minX = std::min(std::floor(gx[i]), minX); maxX = std::max(std::ceil(gx[i]), maxX);
        minY = std::min(std::floor(gy[i]), minY); maxY = std::max(std::ceil(gy[i]), maxY);
    }
    const double bordX = std::fabs(maxX - minX), bordY = std::fabs(maxY - minY);
    const double windowX = X - 2 * bordX, windowY = Y - 2 * bordY;
    const double corrX = std::ceil(((patchesX * PX) - windowX) / (double)(patchesX - 1)), corrY = std::ceil(((patchesY * PY) - windowY) / (double)(patchesY - 1));
    const double stepX = PX - corrX, stepY = PY - corrY;
    tl.resize((size_t)patchesX * patchesY * 2);
    for (int py = 0; py < patchesY; ++py)
        for (int px = 0; px < patchesX; ++px) {
            const size_t p = (size_t)py * patchesX + px;
            const double tlx = bordX + px * stepX, tly = bordY + py * stepY;
            tl[2 * p] = tlx; tl[2 * p + 1] = tly;
            centers[2 * p] = (tlx + (tlx + PX - 1)) / 2; centers[2 * p + 1] = (tly + (tly + PY - 1)) / 2;        // Rectangle::getCenter
        }
}

// BSplineHelper::computeBSplineCoeffs (bspline_helper.cpp:34-87): the spline of lX x lY x lT control points whose value at
// (patch centre, frame) is minus the patch's shift, least squares with one round of outlier rejection
void fa_fit_bspline(int N, int Y, int X, int nP, const double *centers, const double *patchShifts, int lX, int lY, int lT, double *coeffsX, double *coeffsY)
{
    const int R = nP * N, Cc = lX * lY * lT;
    std::vector<double> A((size_t)R * Cc, 0.0), bX(R), bY(R), cX, cY;
    const double hX = (lX == 3) ? X : (X / (double)(lX - 3)), hY = (lY == 3) ? Y : (Y / (double)(lY - 3)), hT = (lT == 3) ? N : (N / (double)(lT - 3));
    for (int i = 0; i < nP; ++i)
        for (int t = 0; t < N; ++t) {
            const int row = t * nP + i;
            const int tcx = (int)centers[(size_t)i * 2], tcy = (int)centers[(size_t)i * 2 + 1];
            for (int ct = -1; ct < lT - 1; ++ct) {
                const double tT = fa_b3((t / hT) - ct);
                if (tT == 0) continue;
                for (int cy = -1; cy < lY - 1; ++cy) {
                    const double tY = fa_b3((tcy / hY) - cy);
                    if (tY == 0) continue;
                    for (int cx = -1; cx < lX - 1; ++cx)
                        A[(size_t)row * Cc + ((ct + 1) * lX * lY) + ((cy + 1) * lX) + (cx + 1)] = tT * tY * fa_b3((tcx / hX) - cx);
                }
            }
            bX[row] = -patchShifts[((size_t)i * N + t) * 2];
            bY[row] = -patchShifts[((size_t)i * N + t) * 2 + 1];
        }
    fa_solve_system(A, R, Cc, bX, bY, 2, cX, cY);
    for (int k = 0; k < Cc; ++k) { coeffsX[k] = cX[k]; coeffsY[k] = cY[k]; }
}

}  // namespace

extern "C" {

int xh_fa_local_alignment(xh_fa *h, const float *d_frames, int32_t N, const float *d_dark, const float *d_gain, const double *h_gShiftX,
                          const double *h_gShiftY, int32_t ref_frame, float max_shift_px, int32_t patchesX, int32_t patchesY, int32_t patchSizeX,
                          int32_t patchSizeY, int32_t patchesAvg, int32_t lX, int32_t lY, int32_t lT, double *h_patchShifts, double *h_centers,
                          double *h_coeffsX, double *h_coeffsY, int32_t *h_dims)
{
    XH_CHECK(h && d_frames && N >= 2 && h_gShiftX && h_gShiftY && ref_frame >= 0 && ref_frame < N && h_patchShifts && h_centers, XH_ERR_ARG,
             "xh_fa_local_alignment: bad argument");
    XH_CHECK(patchesX >= 2 && patchesY >= 2 && patchesAvg >= 1 && lX >= 3 && lY >= 3 && lT >= 3, XH_ERR_ARG,
             "xh_fa_local_alignment: at least 2 x 2 patches, 1 frame per patch and 3 control points per axis");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int Y = h->Y, X = h->X;
    const int PX = patchSizeX & ~1, PY = patchSizeY & ~1;
    XH_CHECK(PX >= 8 && PY >= 8 && PX <= X && PY <= Y, XH_ERR_ARG, "xh_fa_local_alignment: patches of %d x %d pixels in frames of %d x %d (Movie is too small for local alignment.)",
             PX, PY, X, Y);
    // getCorrelationHint (movie_alignment_correlation_gpu.cpp:124-137): the smallest even size that keeps the requested scale
    const float reqScale = (float)h->sizeFactor;
    auto nearestEven = [](int v, float minScale) { int size = 2; while ((size / (float)v) < minScale) size += 2; return size; };
    const int CX = nearestEven(PX, reqScale), CY = nearestEven(PY, reqScale), cxh = CX / 2 + 1;
    if (h_dims) { h_dims[0] = PX; h_dims[1] = PY; h_dims[2] = CX; h_dims[3] = CY; }
    const float actualScale = (float)CX / (float)PX;
    const int maxDist = (int)(max_shift_px * actualScale);
    XH_CHECK(maxDist >= 0, XH_ERR_ARG, "xh_fa_local_alignment: negative --maxShift");
    const bool timing = getenv("XH_FA_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tA = now();
    const int nP = patchesX * patchesY, rows = N * (N - 1) / 2;
    std::vector<double> tl;
    fa_patch_layout(N, Y, X, h_gShiftX, h_gShiftY, patchesX, patchesY, PX, PY, tl, h_centers);
    // the window of every frame of every patch lies inside the frame (getPatchData reads without a check)
    std::vector<int> offs((size_t)nP * N * 2);
    for (int p = 0; p < nP; ++p)
        for (int f = 0; f < N; ++f) {
            const int sx = (int)tl[2 * p] + (int)std::round(h_gShiftX[f]), sy = (int)tl[2 * p + 1] + (int)std::round(h_gShiftY[f]);
            XH_CHECK(sx >= 0 && sy >= 0 && sx + PX <= X && sy + PY <= Y, XH_ERR_ARG,
                     "xh_fa_local_alignment: patch %d of frame %d reaches outside the frame (x %d..%d, y %d..%d): fewer or smaller patches", p, f, sx, sx + PX - 1, sy,
                     sy + PY - 1);
            offs[((size_t)p * N + f) * 2] = sx; offs[((size_t)p * N + f) * 2 + 1] = sy;
        }
    // tables: pruned forward transforms, low-pass of the correlation size with the 1 / (PX PY) of the transform, inverse twiddles --
    // made and uploaded when the sizes they depend on change (the first movie), kept with the handle afterwards
    const int yHalf = CY / 2, xHalf = CX / 2;
    const int y0 = std::max(0, yHalf - maxDist - 1), y1 = std::min(CY - 1, yHalf + maxDist + 1), wy = y1 - y0 + 1;
    const int x0 = std::max(0, xHalf - maxDist - 1), x1 = std::min(CX - 1, xHalf + maxDist + 1), wx = x1 - x0 + 1;
    const size_t E = (size_t)CY * cxh;
    XhBuf *pOffs = nullptr, *pWx = nullptr, *pWy = nullptr, *pTabY = nullptr, *pTabX = nullptr, *pFilter = nullptr, *pPatch = nullptr, *pT = nullptr, *pSingle = nullptr, *pS = nullptr,
          *pU = nullptr, *pW = nullptr, *pRes = nullptr;
    int rc = XH_OK;
    char key[160];
    snprintf(key, sizeof(key), "%d %d %d %d %d %d %d %d %.9g %.9g", PX, PY, CX, CY, y0, wy, x0, wx, (double)(h->Ts / actualScale), (double)h->maxRes);
    if (h->lKey != key) {
        std::vector<fa_cf> Wx((size_t)PX * cxh), Wy((size_t)CY * PY);
        const double twoPi = 6.283185307179586476925286766559;
        for (int x = 0; x < PX; ++x)
            for (int k = 0; k < cxh; ++k) { const double a = -twoPi * (double)(((long long)x * k) % PX) / PX; Wx[(size_t)x * cxh + k] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
        for (int iy = 0; iy < CY; ++iy) {
            const int origY = (iy <= CY / 2) ? iy : (PY - (CY - iy));          // rows 0 .. C/2 from the top, the others from the bottom (scaleFFT2DKernel)
            for (int y = 0; y < PY; ++y) { const double a = -twoPi * (double)(((long long)origY * y) % PY) / PY; Wy[(size_t)iy * PY + y] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
        }
        const std::vector<float> filter = fa_make_lpf((double)(h->Ts / actualScale), h->maxRes, CX, CY, 1.0 / ((double)PX * PY));
        std::vector<fa_cf> tabY((size_t)CY * wy), tabX((size_t)cxh * wx);           // e^{2 pi i ky y / CY}, e^{2 pi i kx x / CX} of the window
        for (int ky = 0; ky < CY; ++ky)
            for (int yy = 0; yy < wy; ++yy) { const double a = twoPi * (double)(((long long)ky * (y0 + yy)) % CY) / CY; tabY[(size_t)ky * wy + yy] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
        for (int kx = 0; kx < cxh; ++kx)
            for (int xx = 0; xx < wx; ++xx) { const double a = twoPi * (double)(((long long)kx * (x0 + xx)) % CX) / CX; tabX[(size_t)kx * wx + xx] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
        rc = fa_table(h, "l_Wx", Wx.data(), sizeof(fa_cf) * Wx.size(), &pWx);
        if (rc == XH_OK && wy <= 32) {
            // the window's row factors in groups of rwp = wy rounded up to a multiple of four entries per ky, zeros beyond wy (k_fa_patch_corr2)
            const int rwp = (wy + 3) / 4 * 4;
            std::vector<fa_cf> tabYp((size_t)CY * rwp, fa_cf{0.f, 0.f});
            for (int ky = 0; ky < CY; ++ky)
                for (int yy = 0; yy < wy; ++yy) tabYp[(size_t)ky * rwp + yy] = tabY[(size_t)ky * wy + yy];
            XhBuf *pP = nullptr;
            rc = fa_table(h, "l_tabYp", tabYp.data(), sizeof(fa_cf) * tabYp.size(), &pP);
        }
        if (rc == XH_OK) rc = fa_table(h, "l_Wy", Wy.data(), sizeof(fa_cf) * Wy.size(), &pWy);
        if (rc == XH_OK) rc = fa_table(h, "l_tabY", tabY.data(), sizeof(fa_cf) * tabY.size(), &pTabY);
        if (rc == XH_OK) rc = fa_table(h, "l_tabX", tabX.data(), sizeof(fa_cf) * tabX.size(), &pTabX);
        if (rc == XH_OK) rc = fa_table(h, "l_filter", filter.data(), sizeof(float) * filter.size(), &pFilter);
        if (rc == XH_OK) h->lKey = key;
    }
    const double tB = now();
    if (rc == XH_OK) rc = fa_scratch(h, "l_Wx", 0, &pWx);
    if (rc == XH_OK) rc = fa_scratch(h, "l_Wy", 0, &pWy);
    if (rc == XH_OK) rc = fa_scratch(h, "l_tabY", 0, &pTabY);
    if (rc == XH_OK) rc = fa_scratch(h, "l_tabX", 0, &pTabX);
    if (rc == XH_OK) rc = fa_scratch(h, "l_filter", 0, &pFilter);
    if (rc == XH_OK) rc = fa_table(h, "l_offs", offs.data(), sizeof(int) * offs.size(), &pOffs);          // the patch corners move with the global shifts
    // PB patches at a time: the pair kernel has one wave per frame pair, the second product one workgroup per patch frame -- a single patch (780
    // waves, 40 workgroups for 40 frames) leaves most of the device idle; 36 (three launches for the 108 patches of a K3 movie) against
    // 16: local alignment 18.5 -> 17.5 ms per movie
    static const int pbEnv = xh_debug_env("XH_FA_PB") ? atoi(xh_debug_env("XH_FA_PB")) : 0;        // A/B runs
    const int PB = std::min(nP, pbEnv > 0 ? pbEnv : 36);
    static const bool copyPatchesEnv = xh_debug_env("XH_FA_COPY_PATCHES") != nullptr;
    if (rc == XH_OK) rc = fa_scratch(h, "l_patch", (h->use_mfma && !copyPatchesEnv) ? 16 : sizeof(float) * (size_t)PB * N * PY * PX, &pPatch);      // the fused product reads the frames
    if (rc == XH_OK) rc = fa_scratch(h, "l_T", sizeof(fa_cf) * (size_t)PB * N * PY * cxh, &pT);
    if (rc == XH_OK) rc = fa_scratch(h, "l_single", sizeof(fa_cf) * (size_t)PB * N * E, &pSingle);
    if (rc == XH_OK) rc = fa_scratch(h, "l_S", sizeof(fa_cf) * (size_t)PB * N * E, &pS);
    if (rc == XH_OK) rc = fa_scratch(h, "l_U", sizeof(fa_cf) * (size_t)PB * rows * wy * cxh, &pU);
    if (rc == XH_OK) rc = fa_scratch(h, "l_W", sizeof(float) * (size_t)PB * rows * wy * wx, &pW);
    if (rc == XH_OK) rc = fa_scratch(h, "l_res", sizeof(double) * 2 * (size_t)rows * nP, &pRes);
    if (rc != XH_OK) return rc;
    XhBuf &bOffs = *pOffs, &bWx = *pWx, &bWy = *pWy, &bTabY = *pTabY, &bTabX = *pTabX, &bFilter = *pFilter, &bPatch = *pPatch, &bT = *pT, &bSingle = *pSingle, &bS = *pS, &bU = *pU,
          &bW = *pW, &bRes = *pRes;
    auto freeAll = [&]() {};
    const double tC = now();
    for (int p0 = 0; p0 < nP && rc == XH_OK; p0 += PB) {
        const int pb = std::min(PB, nP - p0), nf = pb * N;
        const size_t tot = (size_t)nf * PY * PX;
        static const bool copyPatches = xh_debug_env("XH_FA_COPY_PATCHES") != nullptr;        // A/B runs
        const bool fused = h->use_mfma && !copyPatches;
        if (!fused)
            hipLaunchKernelGGL(k_fa_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_frames, d_dark, d_gain, (const int *)bOffs.p + (size_t)p0 * N * 2,
                               (float *)bPatch.p, nf, N, Y, X, PY, PX);
        // along x, all frames of all patches of the batch at once: [pb N PY][PX] x [PX][cxh]; the matrix-core product reads the patches
        // out of the frames
        if (fused) {
            FaGather G{d_dark, d_gain, (const int *)bOffs.p + (size_t)p0 * N * 2, N, Y, X, PY};
            hipLaunchKernelGGL((k_fa_gemm_mfma<false, true>), dim3((2 * cxh + 127) / 128, (unsigned)(((size_t)nf * PY + 127) / 128), 1), dim3(256), 0, ctx->stream, d_frames,
                               (size_t)PX, (size_t)0, (const fa_cf *)bWx.p, (size_t)cxh, (size_t)0, (fa_cf *)bT.p, (size_t)cxh, (size_t)0, nf * PY, cxh, PX, G);
        } else if (h->use_mfma)
            hipLaunchKernelGGL((k_fa_gemm_mfma<false>), dim3((2 * cxh + 127) / 128, (unsigned)(((size_t)nf * PY + 127) / 128), 1), dim3(256), 0, ctx->stream, (const float *)bPatch.p,
                               (size_t)PX, (size_t)0, (const fa_cf *)bWx.p, (size_t)cxh, (size_t)0, (fa_cf *)bT.p, (size_t)cxh, (size_t)0, nf * PY, cxh, PX, FaGather{});
        else
        hipLaunchKernelGGL((k_fa_gemm<false>), dim3((cxh + 31) / 32, (unsigned)(((size_t)nf * PY + 63) / 64), 1), dim3(256), 0, ctx->stream, (const float *)bPatch.p, (size_t)PX,
                           (size_t)0, (const fa_cf *)bWx.p, (size_t)cxh, (size_t)0, (fa_cf *)bT.p, (size_t)cxh, (size_t)0, nf * PY, cxh, PX);
        // along y, frame by frame: [CY][PY] x [PY][cxh]
        if (h->use_mfma)
            hipLaunchKernelGGL((k_fa_gemm_mfma<true>), dim3((2 * cxh + 127) / 128, (CY + 127) / 128, nf), dim3(256), 0, ctx->stream, (const float *)bWy.p, (size_t)PY, (size_t)0,
                               (const fa_cf *)bT.p, (size_t)cxh, (size_t)PY * cxh, (fa_cf *)bSingle.p, (size_t)cxh, E, CY, cxh, PY, FaGather{});
        else
        hipLaunchKernelGGL((k_fa_gemm<true>), dim3((cxh + 31) / 32, (CY + 63) / 64, nf), dim3(256), 0, ctx->stream, (const float *)bWy.p, (size_t)PY, (size_t)0,
                           (const fa_cf *)bT.p, (size_t)cxh, (size_t)PY * cxh, (fa_cf *)bSingle.p, (size_t)cxh, E, CY, cxh, PY);
        hipLaunchKernelGGL(k_fa_patch_sum, dim3((unsigned)(((size_t)nf * E + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)bSingle.p, (fa_cf *)bS.p,
                           (const float *)bFilter.p, nf, N, E, patchesAvg);
        const size_t ldsU = sizeof(fa_cf) * (size_t)wy * cxh;
        if (h->pairwin_form != 0 && wy <= 32 && wx <= 32 && ldsU <= 48 * 1024) {
            XhBuf *pP = nullptr;
            rc = fa_scratch(h, "l_tabYp", 0, &pP);
            if (rc != XH_OK) break;
            const dim3 g(rows, pb), bdim(std::min(256, 64 * ((cxh + 63) / 64)));
            double *o = (double *)bRes.p + 2 * (size_t)rows * p0;
#define XH_FA_PC2(RW_)                                                                                                                                                  \
    hipLaunchKernelGGL((k_fa_patch_corr2<RW_>), g, bdim, ldsU, ctx->stream, (const fa_cf *)bS.p, N, CY, CX, (const fa_v2 *)pP->p, (const fa_cf *)bTabX.p, y0, wy, x0, wx, \
                       maxDist, (float *)bW.p, o)
            switch ((wy + 3) / 4) {
            case 1: XH_FA_PC2(4); break;
            case 2: XH_FA_PC2(8); break;
            case 3: XH_FA_PC2(12); break;
            case 4: XH_FA_PC2(16); break;
            case 5: XH_FA_PC2(20); break;
            case 6: XH_FA_PC2(24); break;
            case 7: XH_FA_PC2(28); break;
            default: XH_FA_PC2(32); break;
            }
#undef XH_FA_PC2
        } else
        hipLaunchKernelGGL(k_fa_patch_corr, dim3(rows, pb), dim3(std::min(256, 64 * ((cxh + 63) / 64))), 0, ctx->stream, (const fa_cf *)bS.p, N, CY, CX, (const fa_cf *)bTabY.p,
                           (const fa_cf *)bTabX.p, y0, wy, x0, wx, maxDist, (fa_cf *)bU.p, (float *)bW.p, (double *)bRes.p + 2 * (size_t)rows * p0);
        if (hipGetLastError() != hipSuccess) { xh_set_error("xh_fa_local_alignment: kernel launch failed"); rc = XH_ERR_HIP; }
    }
    std::vector<double> res(2 * (size_t)rows * nP);
    if (rc == XH_OK && hipMemcpyAsync(res.data(), bRes.p, sizeof(double) * res.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    const double tD = now();
    freeAll();
    h->aheadBase = nullptr;
    if (rc == XH_OK && h->prefilter_ahead) {
        // the warp's prefilter of every frame, behind the results' copy: runs while the host solves below
        // (N Y X floats, 3.8 GB for a K3 movie: when the device cannot spare them the warp prefilters frame by frame as it always could)
        XhBuf *pCoef = nullptr;
        if (fa_scratch(h, "w_coefAll", sizeof(float) * (size_t)N * Y * X, &pCoef) != XH_OK) { (void)hipGetLastError(); pCoef = nullptr; }
        if (pCoef) {
            for (int n = 0; n < N; ++n)
                fa_prefilter_launch(ctx->stream, d_frames + (size_t)n * Y * X, d_dark, d_gain, (float *)pCoef->p + (size_t)n * Y * X, (float *)nullptr, Y, X);
            if (hipGetLastError() != hipSuccess) rc = XH_ERR_HIP;
            else { h->aheadBase = d_frames; h->aheadDark = d_dark; h->aheadGain = d_gain; h->aheadN = N; }
        }
    }
    const double tE = now();
    if (rc != XH_OK) { if (rc == XH_ERR_HIP) xh_set_error("xh_fa_local_alignment: device error"); return rc; }
    // computeAlignment (:776-797) per patch: deduct the centre, scale to the movie's pixels, solve, add the rounded global shift
    auto solvePatches = [&](int pBegin, int pEnd) {
        std::vector<double> bx(rows), by(rows), lsx(N), lsy(N);
        for (int p = pBegin; p < pEnd; ++p) {
            for (int r = 0; r < rows; ++r) {
                bx[r] = (res[2 * ((size_t)rows * p + r)] - CX / 2.0) * ((double)PX / CX);
                by[r] = (res[2 * ((size_t)rows * p + r) + 1] - CY / 2.0) * ((double)PY / CY);
            }
            int ref = ref_frame;
            fa_solve(bx, by, N, 2, ref_frame, lsx.data(), lsy.data(), &ref);
            for (int t = 0; t < N; ++t) {
                h_patchShifts[((size_t)p * N + t) * 2] = std::round(h_gShiftX[t]) + lsx[t];
                h_patchShifts[((size_t)p * N + t) * 2 + 1] = std::round(h_gShiftY[t]) + lsy[t];
            }
        }
    };
    {
        // the patches are independent: a few host threads
        const int nthr = std::max(1, std::min({nP, 16, (int)std::thread::hardware_concurrency()}));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t) pool.emplace_back(solvePatches, (int)((long long)nP * t / nthr), (int)((long long)nP * (t + 1) / nthr));
        for (auto &th : pool) th.join();
    }
    const double tF = now();
    if (h_coeffsX && h_coeffsY) fa_fit_bspline(N, Y, X, nP, h_centers, h_patchShifts, lX, lY, lT, h_coeffsX, h_coeffsY);
    if (timing)
        fprintf(stderr, "xh_fa_local_alignment: tables %.1f ms, buffers %.1f ms, device %.1f ms, free %.1f ms, patch solves %.1f ms, spline fit %.1f ms\n", 1e3 * (tB - tA),
                1e3 * (tC - tB), 1e3 * (tD - tC), 1e3 * (tE - tD), 1e3 * (tF - tE), 1e3 * (now() - tF));
    return XH_OK;
}

// CUDAFlexAlignCorrelate<T>::run (reconstruction_cuda/cuda_flexalign_correlate.cpp:95-140) on its own: N real frames [N][Y][X]
// (even sizes) -> the position of the correlation maximum of every pair i < j in pixels of the map (centre = (X/2, Y/2)), the
// stage of the local alignment between the patch spectra and the solver, without crop and filter. The reference tests this
// stage alone (test_cuda_flexalign_correlate.cpp), so it is callable alone here too.
int xh_fa_correlate(xh_ctx *ctx, const float *d_frames, int32_t N, int32_t Y, int32_t X, float max_dist, double *h_pos)
{
    XH_CHECK(ctx && d_frames && h_pos && N >= 2 && Y >= 4 && X >= 4 && (Y & 1) == 0 && (X & 1) == 0 && max_dist >= 0, XH_ERR_ARG,
             "xh_fa_correlate: bad argument (even frame sizes, at least two frames)");
    XH_HIP(hipSetDevice(ctx->device));
    const int xh = X / 2 + 1, rows = N * (N - 1) / 2, maxDist = (int)max_dist;
    const double twoPi = 6.283185307179586476925286766559;
    std::vector<fa_cf> Wx((size_t)X * xh), Wy((size_t)Y * Y);
    for (int x = 0; x < X; ++x)
        for (int k = 0; k < xh; ++k) { const double a = -twoPi * (double)(((long long)x * k) % X) / X; Wx[(size_t)x * xh + k] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
    for (int k = 0; k < Y; ++k)
        for (int y = 0; y < Y; ++y) { const double a = -twoPi * (double)(((long long)k * y) % Y) / Y; Wy[(size_t)k * Y + y] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
    const int yHalf = Y / 2, xHalf = X / 2;
    const int y0 = std::max(0, yHalf - maxDist - 1), y1 = std::min(Y - 1, yHalf + maxDist + 1), wy = y1 - y0 + 1;
    const int x0 = std::max(0, xHalf - maxDist - 1), x1 = std::min(X - 1, xHalf + maxDist + 1), wx = x1 - x0 + 1;
    std::vector<fa_cf> tabY((size_t)Y * wy), tabX((size_t)xh * wx);
    for (int ky = 0; ky < Y; ++ky)
        for (int yy = 0; yy < wy; ++yy) { const double a = twoPi * (double)(((long long)ky * (y0 + yy)) % Y) / Y; tabY[(size_t)ky * wy + yy] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
    for (int kx = 0; kx < xh; ++kx)
        for (int xx = 0; xx < wx; ++xx) { const double a = twoPi * (double)(((long long)kx * (x0 + xx)) % X) / X; tabX[(size_t)kx * wx + xx] = fa_cf{(float)std::cos(a), (float)std::sin(a)}; }
    XhBuf bWx, bWy, bTabY, bTabX, bT, bS, bU, bW, bRes;
    auto freeAll = [&]() { XhBuf *all[] = {&bWx, &bWy, &bTabY, &bTabX, &bT, &bS, &bU, &bW, &bRes}; for (XhBuf *q : all) xh_buf_free(*q); };
    const size_t E = (size_t)Y * xh;
    int rc = fa_upload(Wx.data(), bWx, sizeof(fa_cf) * Wx.size(), ctx);
    if (rc == XH_OK) rc = fa_upload(Wy.data(), bWy, sizeof(fa_cf) * Wy.size(), ctx);
    if (rc == XH_OK) rc = fa_upload(tabY.data(), bTabY, sizeof(fa_cf) * tabY.size(), ctx);
    if (rc == XH_OK) rc = fa_upload(tabX.data(), bTabX, sizeof(fa_cf) * tabX.size(), ctx);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bT, sizeof(fa_cf) * (size_t)N * E);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bS, sizeof(fa_cf) * (size_t)N * E);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bU, sizeof(fa_cf) * (size_t)rows * wy * xh);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bW, sizeof(float) * (size_t)rows * wy * wx);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, bRes, sizeof(double) * 2 * (size_t)rows);
    if (rc == XH_OK) {
        hipLaunchKernelGGL((k_fa_gemm<false>), dim3((xh + 31) / 32, (unsigned)(((size_t)N * Y + 63) / 64), 1), dim3(256), 0, ctx->stream, d_frames, (size_t)X, (size_t)0,
                           (const fa_cf *)bWx.p, (size_t)xh, (size_t)0, (fa_cf *)bT.p, (size_t)xh, (size_t)0, N * Y, xh, X);
        hipLaunchKernelGGL((k_fa_gemm<true>), dim3((xh + 31) / 32, (Y + 63) / 64, N), dim3(256), 0, ctx->stream, (const float *)bWy.p, (size_t)Y, (size_t)0, (const fa_cf *)bT.p,
                           (size_t)xh, E, (fa_cf *)bS.p, (size_t)xh, E, Y, xh, Y);
        hipLaunchKernelGGL(k_fa_patch_corr, dim3(rows, 1), dim3(std::min(256, 64 * ((xh + 63) / 64))), 0, ctx->stream, (const fa_cf *)bS.p, N, Y, X, (const fa_cf *)bTabY.p,
                           (const fa_cf *)bTabX.p, y0, wy, x0, wx, maxDist, (fa_cf *)bU.p, (float *)bW.p, (double *)bRes.p);
        if (hipGetLastError() != hipSuccess) rc = XH_ERR_HIP;
    }
    if (rc == XH_OK && hipMemcpyAsync(h_pos, bRes.p, sizeof(double) * 2 * (size_t)rows, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = XH_ERR_HIP;
    freeAll();
    if (rc == XH_ERR_HIP) xh_set_error("xh_fa_correlate: device error");
    return rc;
}

// localFromGlobal (movie_alignment_correlation_gpu.cpp:432-456): the B-spline of a movie aligned globally only -- every patch
// carries the global shift of its frame
int xh_fa_local_from_global(xh_fa *h, int32_t N, const double *h_gShiftX, const double *h_gShiftY, int32_t patchesX, int32_t patchesY, int32_t patchSizeX,
                            int32_t patchSizeY, int32_t lX, int32_t lY, int32_t lT, double *h_centers, double *h_coeffsX, double *h_coeffsY)
{
    XH_CHECK(h && N >= 1 && h_gShiftX && h_gShiftY && h_centers && h_coeffsX && h_coeffsY && patchesX >= 2 && patchesY >= 2 && lX >= 3 && lY >= 3 && lT >= 3, XH_ERR_ARG,
             "xh_fa_local_from_global: bad argument");
    const int PX = patchSizeX & ~1, PY = patchSizeY & ~1, nP = patchesX * patchesY;
    std::vector<double> tl, shifts((size_t)nP * N * 2);
    fa_patch_layout(N, h->Y, h->X, h_gShiftX, h_gShiftY, patchesX, patchesY, PX, PY, tl, h_centers);
    for (int p = 0; p < nP; ++p)
        for (int t = 0; t < N; ++t) { shifts[((size_t)p * N + t) * 2] = h_gShiftX[t]; shifts[((size_t)p * N + t) * 2 + 1] = h_gShiftY[t]; }
    fa_fit_bspline(N, h->Y, h->X, nP, h_centers, shifts.data(), lX, lY, lT, h_coeffsX, h_coeffsY);
    return XH_OK;
}

int xh_fa_apply_bspline(xh_fa *h, const float *d_frame, const float *d_dark, const float *d_gain, const double *h_coeffsX, const double *h_coeffsY, int32_t lX,
                        int32_t lY, int32_t lT, int32_t N, int32_t n, float *d_out, float *d_sum, float *d_initial_sum)
{
    XH_CHECK(h && d_frame && h_coeffsX && h_coeffsY && lX >= 3 && lY >= 3 && lT >= 3 && N >= 1 && n >= 0 && n < N, XH_ERR_ARG, "xh_fa_apply_bspline: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int Y = h->Y, X = h->X, Cc = lX * lY * lT;
    std::vector<float> c(2 * (size_t)Cc);                                   // the reference's coefficients are Matrix1D<float>
    for (int k = 0; k < Cc; ++k) { c[k] = (float)h_coeffsX[k]; c[Cc + k] = (float)h_coeffsY[k]; }
    if (c != h->warpCHost) {
        // a new set of coefficients: wait for the frames still being warped with the old one, then upload (the forty frames of a
        // movie share one set: one upload and one synchronisation per movie)
        XH_HIP(hipStreamSynchronize(ctx->stream));
        XH_TRY(xh_buf_reserve(ctx, h->warpC, sizeof(float) * c.size()));
        h->warpCHost = c;
        XH_HIP(hipMemcpyAsync(h->warpC.p, h->warpCHost.data(), sizeof(float) * c.size(), hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(hipStreamSynchronize(ctx->stream));
    }
    float *coef = (float *)h->work.p;
    const bool ahead = h->aheadBase && !d_initial_sum && N == h->aheadN && d_frame == h->aheadBase + (size_t)n * Y * X && d_dark == h->aheadDark && d_gain == h->aheadGain &&
                       h->cache.count("w_coefAll");
    if (ahead) coef = (float *)h->cache["w_coefAll"].p + (size_t)n * Y * X;
    else {
        fa_prefilter_launch(ctx->stream, d_frame, d_dark, d_gain, coef, d_initial_sum, Y, X);
    }
    if (d_out || d_sum) {
        // hX, hY, tPos in float on the host like applyBSplineTransform (cuda_gpu_geo_transformer.cpp:206-210)
        const float hX = (lX == 3) ? (float)X : (X / (float)(lX - 3)), hY = (lY == 3) ? (float)Y : (Y / (float)(lY - 3)), hT = (lT == 3) ? (float)N : (N / (float)(lT - 3));
        const float tPos = n / hT;
        static const bool plain = xh_debug_env("XH_FA_WARP_PLAIN") != nullptr;        // A/B runs
        // dynamic LDS of the two forms: four layers of quads (16 bytes per control row and first column, both fields) or of plain control
        // points; a control grid whose quads do not fit 64 KB takes the plain kernel, one that fits neither is refused
        const size_t ldsQuads = sizeof(float4) * 2 * 4 * (size_t)(lX - 3) * lY, ldsPlain = sizeof(float) * 2 * 4 * (size_t)lX * lY;
        XH_CHECK(ldsPlain <= 64 * 1024, XH_ERR_UNSUPPORTED, "xh_fa_apply_bspline: %d x %d control points per layer exceed the 64 KB of LDS the warp kernel stages them in", lX, lY);
        if (lX >= 4 && lY >= 4 && !plain && ldsQuads <= 64 * 1024)
            hipLaunchKernelGGL(k_fa_warp_quads, dim3((X + 63) / 64, (Y + 4 * XH_FA_WARP_RG - 1) / (4 * XH_FA_WARP_RG)), dim3(256), ldsQuads, ctx->stream, (const float *)coef,
                               (const float *)h->warpC.p, (const float *)h->warpC.p + Cc, lX, lY, lT, hX, hY, tPos, Y, X, d_out, d_sum);
        else
        hipLaunchKernelGGL(k_fa_warp, dim3((X + 63) / 64, (Y + 3) / 4), dim3(256), ldsPlain, ctx->stream, (const float *)coef, (const float *)h->warpC.p,
                           (const float *)h->warpC.p + Cc, lX, lY, lT, hX, hY, tPos, Y, X, d_out, d_sum);
    }
    XH_LAUNCH_CHECK();
    return XH_OK;
}

// applyShiftsComputeAverage's loop (movie_alignment_correlation_gpu.cpp:479-560) in one call: frames n0 .. n1 of d_frames [N][Y][X] warped by the
// spline of their frame index and added into d_sum / d_initial_sum (either may be null); d_out_stack (null or [n1 - n0 + 1][Y][X])
// receives the aligned frames.  Saves a host round trip per frame.
int xh_fa_apply_bspline_frames(xh_fa *h, const float *d_frames, int32_t N, int32_t n0, int32_t n1, const float *d_dark, const float *d_gain, const double *h_coeffsX,
                               const double *h_coeffsY, int32_t lX, int32_t lY, int32_t lT, float *d_out_stack, float *d_sum, float *d_initial_sum)
{
    XH_CHECK(h && d_frames && N >= 1 && n0 >= 0 && n1 >= n0 && n1 < N, XH_ERR_ARG, "xh_fa_apply_bspline_frames: bad argument");
    const size_t per = (size_t)h->Y * h->X;
    for (int n = n0; n <= n1; ++n)
        XH_TRY(xh_fa_apply_bspline(h, d_frames + (size_t)n * per, d_dark, d_gain, h_coeffsX, h_coeffsY, lX, lY, lT, N, n, d_out_stack ? d_out_stack + (size_t)(n - n0) * per : nullptr,
                                   d_sum, d_initial_sum));
    return XH_OK;
}

// ---- dose weighting of movie frames: ProgMovieFilterDose (reconstruction/movie_filter_dose.cpp:85-170, 283) -----------------------
}  // extern "C"

namespace {
// applyDoseFilterToImage on the full spectrum of a real frame (the filter depends on the squared frequencies only): critical
// dose of the spatial frequency (summovie's curve), the frame contributes where its final dose is nearer the optimal dose than
// its initial dose, attenuated by exp(-dose / (2 critical dose))
__global__ void __launch_bounds__(256) k_dose_apply(fa_cf *__restrict__ F, int Y, int X, double pixel_size, double vscale, double dose_start, double dose_finish)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)Y * X) return;
    const int i = (int)(t / X), j = (int)(t - (size_t)i * X);
    const double y = (i <= Y / 2 ? (double)i : (double)(i - Y)) * (1.0 / Y), x = (j <= X / 2 ? (double)j : (double)(j - X)) * (1.0 / X);
    double crit;
    if (i == 0 && j == 0) crit = 1.79769313486231570815e+308 * 0.001;
    else crit = ((0.24499 * pow(sqrt(x * x + y * y) / pixel_size, -1.6649)) + 2.8141) * vscale;
    const double opt = 2.51284 * crit;
    fa_cf v = F[t];
    if (fabs(dose_finish - opt) < fabs(dose_start - opt)) {
        const double f = exp((-0.5 * dose_finish) / crit);
        v = fa_cf{(float)(v.x * f), (float)(v.y * f)};
    } else v = fa_cf{0.f, 0.f};
    F[t] = v;
}

__global__ void __launch_bounds__(256) k_dose_store(const fa_cf *__restrict__ F, float *__restrict__ out, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < tot) out[t] = F[t].x;
}
}  // namespace

extern "C" {

int xh_movie_dose_filter(xh_ctx *ctx, xh_fft2d *plan, float *d_frame, int32_t Y, int32_t X, double pixel_size, double acc_voltage, double dose_start,
                         double dose_finish)
{
    XH_CHECK(ctx && plan && d_frame && Y >= 2 && X >= 2 && pixel_size > 0, XH_ERR_ARG, "xh_movie_dose_filter: bad argument");
    double vscale;
    if (acc_voltage < 301 && acc_voltage > 299.) vscale = 1.0;
    else if (acc_voltage < 201.0 && acc_voltage > 199.0) vscale = 0.8;
    else { xh_set_error("xh_movie_dose_filter: Bad acceleration voltage (must be 200 or 300 kV"); return XH_ERR_ARG; }     // initVoltage, :112-124
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)Y * X;
    void *wp = nullptr;
    XH_TRY(xh_fft2d_user_scratch(plan, sizeof(fa_cf) * tot, &wp));        // lives with the plan: no allocation per frame
    fa_cf *F = (fa_cf *)wp;
    const unsigned grid = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_fa_load, dim3(grid), dim3(256), 0, ctx->stream, (const float *)d_frame, (const float *)nullptr, (const float *)nullptr, F, tot);
    int rc = xh_fft2d_exec(plan, (float *)F, 0);
    if (rc == XH_OK) {
        hipLaunchKernelGGL(k_dose_apply, dim3(grid), dim3(256), 0, ctx->stream, F, Y, X, pixel_size, vscale, dose_start, dose_finish);
        rc = xh_fft2d_exec(plan, (float *)F, 1);
    }
    if (rc == XH_OK) {
        hipLaunchKernelGGL(k_dose_store, dim3(grid), dim3(256), 0, ctx->stream, (const fa_cf *)F, d_frame, tot);
        if (hipGetLastError() != hipSuccess) { xh_set_error("xh_movie_dose_filter: device error"); rc = XH_ERR_HIP; }        // (no synchronisation: stream order)
    }
    return rc;
}

}  // extern "C"

namespace {
// Binning of a frame as the CUDA program's loader does it (CUDAFlexAlignScale::runScaleIFT, cuda_flexalign_scale.cpp:101-121 +
// scaleFFT2DKernel, cuda_scaleFFT_kernels.cu:44-79): the half spectrum of the raw frame is cropped to the binned size -- columns
// 0 .. Xb/2, rows 0 .. Yb/2 from the top and the last Yb - Yb/2 - 1 from the bottom, times 1 / (X Y) -- and transformed back.  The
// inverse real transform of a half spectrum is the real part of the complex inverse of its Hermitian completion, which is what
// this kernel writes: G[ky][kx] = H[ky][kx] for kx <= Xb/2, conj(H[(Yb - ky) % Yb][Xb - kx]) beyond.
__global__ void __launch_bounds__(256) k_bin_crop(const fa_cf *__restrict__ F, fa_cf *__restrict__ G, int Y, int X, int Yb, int Xb, float norm)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)Yb * Xb) return;
    const int ky = (int)(t / Xb), kx = (int)(t - (size_t)ky * Xb);
    const int xh = Xb / 2, yhalf = Yb / 2;
    const bool direct = kx <= xh;
    const int idy = direct ? ky : (Yb - ky) % Yb, idx = direct ? kx : Xb - kx;
    const int origY = (idy <= yhalf) ? idy : (Y - (Yb - idy));
    fa_cf v = F[(size_t)origY * X + idx];
    v.x *= norm; v.y *= norm;
    if (!direct) v.y = -v.y;
    G[t] = v;
}
}  // namespace

// ---- frames as the detector stores them: MRC modes 0 (int8), 1 (int16), 6 (uint16), 2 (float32, a copy) and, beyond MRC, 100 (uint8)
// -> float32, the cast of Image<float>::read (xmippCore castPage2T) done after the host copy instead of before it: a K3 frame crosses
// the link as 23.6 MB of counts instead of 94 MB of floats.  A thread converts 16 bytes of input.
namespace {
__global__ void __launch_bounds__(256) k_movie_crop(const float *__restrict__ src, float *__restrict__ dst, int Y, int X, int cy, int cx, size_t tot)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tot) return;
    const size_t per = (size_t)cy * cx, n = t / per, r = t - n * per;
    const int y = (int)(r / cx), x = (int)(r - (size_t)y * cx);
    dst[t] = src[(n * Y + y) * (size_t)X + x];
}

// four counts per thread: a wave reads 256 (or 512) contiguous bytes and writes 1 KB as one run of float4 (sixteen counts per thread
// left every store instruction with 16 bytes in each 64-byte piece: 2.1 TB/s where this form streams)
template <typename TIN>
__global__ void __launch_bounds__(256) k_frame_to_float(const TIN *__restrict__ in, float *__restrict__ out, size_t n)
{
    constexpr int V = 4;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 >= n) return;
    if (i0 + V <= n && ((uintptr_t)in & (V * sizeof(TIN) - 1)) == 0 && ((uintptr_t)out & 15) == 0) {
        TIN v[V];
        if (sizeof(TIN) == 1) *reinterpret_cast<unsigned *>(v) = *reinterpret_cast<const unsigned *>(in + i0);
        else *reinterpret_cast<uint2 *>(v) = *reinterpret_cast<const uint2 *>(in + i0);
        *reinterpret_cast<float4 *>(out + i0) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else
        for (size_t i = i0; i < n && i < i0 + V; ++i) out[i] = (float)in[i];
}
}  // namespace

extern "C" {

int xh_movie_crop_frames(xh_ctx *ctx, const float *d_src, int32_t N, int32_t Y, int32_t X, int32_t cropY, int32_t cropX, float *d_dst)
{
    XH_CHECK(ctx && d_src && d_dst && N >= 1 && cropY >= 1 && cropX >= 1 && cropY <= Y && cropX <= X, XH_ERR_ARG, "xh_movie_crop_frames: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)N * cropY * cropX;
    hipLaunchKernelGGL(k_movie_crop, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_src, d_dst, Y, X, cropY, cropX, tot);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

int xh_movie_frame_to_float(xh_ctx *ctx, const void *d_raw, int32_t mode, int64_t n, float *d_out)
{
    XH_CHECK(ctx && d_raw && d_out && n >= 0, XH_ERR_ARG, "xh_movie_frame_to_float: bad argument");
    XH_CHECK(mode == 0 || mode == 1 || mode == 2 || mode == 6 || mode == 100, XH_ERR_UNSUPPORTED,
             "xh_movie_frame_to_float: mode %d (0 int8, 1 int16, 2 float32, 6 uint16, 100 uint8)", (int)mode);
    XH_HIP(hipSetDevice(ctx->device));
    if (n == 0) return XH_OK;
    const size_t N = (size_t)n;
#define XH_F2F(T_) hipLaunchKernelGGL((k_frame_to_float<T_>), dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, ctx->stream, \
                                      (const T_ *)d_raw, d_out, N)
    if (mode == 0) XH_F2F(signed char);
    else if (mode == 1) XH_F2F(short);
    else if (mode == 6) XH_F2F(unsigned short);
    else if (mode == 100) XH_F2F(unsigned char);
    else XH_HIP(hipMemcpyAsync(d_out, d_raw, N * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
#undef XH_F2F
    XH_LAUNCH_CHECK();
    return XH_OK;
}

int xh_movie_bin_frame(xh_ctx *ctx, xh_fft2d *planRaw, xh_fft2d *planBinned, const float *d_frame, const float *d_dark, const float *d_gain, int32_t Y, int32_t X,
                       float *d_out, int32_t Yb, int32_t Xb)
{
    XH_CHECK(ctx && planRaw && planBinned && d_frame && d_out && Y >= 2 && X >= 2 && Yb >= 2 && Xb >= 2 && Yb <= Y && Xb <= X, XH_ERR_ARG, "xh_movie_bin_frame: bad argument");
    XH_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)Y * X, totb = (size_t)Yb * Xb;
    struct { void *p; } wa{nullptr}, wb{nullptr};
    XH_TRY(xh_fft2d_user_scratch(planRaw, sizeof(fa_cf) * tot, &wa.p));
    int rc = xh_fft2d_user_scratch(planBinned, sizeof(fa_cf) * totb, &wb.p);
    if (rc == XH_OK) {
        hipLaunchKernelGGL(k_fa_load, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_frame, d_dark, d_gain, (fa_cf *)wa.p, tot);
        rc = xh_fft2d_exec(planRaw, (float *)wa.p, 0);
    }
    if (rc == XH_OK) {
        hipLaunchKernelGGL(k_bin_crop, dim3((unsigned)((totb + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)wa.p, (fa_cf *)wb.p, Y, X, Yb, Xb,
                           ((float)Xb * (float)Yb) / ((float)X * (float)Y));        // 1 / (X Y) of the reference; xh_fft2d's inverse divides by Xb Yb, cuFFT's does not
        rc = xh_fft2d_exec(planBinned, (float *)wb.p, 1);
    }
    if (rc == XH_OK) {
        hipLaunchKernelGGL(k_dose_store, dim3((unsigned)((totb + 255) / 256)), dim3(256), 0, ctx->stream, (const fa_cf *)wb.p, d_out, totb);
        if (hipGetLastError() != hipSuccess) { xh_set_error("xh_movie_bin_frame: device error"); rc = XH_ERR_HIP; }
    }
    return rc;
}

}  // extern "C"
