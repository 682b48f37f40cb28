// xh_bspline.h -- cubic B-spline prefilter / interpolation device code shared by the
// projection-matching and gridding translation units (gfx950).
// xmippCore produceSplineCoefficients + interpolatedElementBSpline2D(.,.,3); in-tree evidence:
// reconstruction_cuda/cuda_gpu_iirconvolve.cu:28-41, cuda_gpu_bilib.cu:16-25,
// cuda_gpu_multidim_array.cu:78-157.
#ifndef XH_BSPLINE_H
#define XH_BSPLINE_H
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <algorithm>

// cubic B-spline prefilter, pole sqrt(3)-2, half-sample mirror boundary (xmippCore
// produceSplineCoefficients; in-tree GPU twin reconstruction_cuda/cuda_gpu_iirconvolve.cu:28-41)
template <typename T>
__device__ __forceinline__ void d_prefilter_line(T *c, int n, int stride)
{
    if (n == 1) return;
    const T z = (T)(-0.26794919243112270647);   // sqrt(3) - 2
    const T lambda = (T)6.0;
    for (int i = 0; i < n; ++i) c[i * stride] *= lambda;
    T sum = c[0];
    T zk = z;
    const int H = 64;                          // |z|^64 ~ 2e-37: below fp64 resolution
    int k = 1;
    for (; k <= n && k <= H; ++k) { sum += zk * c[(k - 1) * stride]; zk *= z; }
    if (n < H) {
        for (k = n + 1; k <= 2 * n - 1; ++k) { sum += zk * c[(2 * n - k) * stride]; zk *= z; }
        sum /= ((T)1 - zk);
    }
    c[0] = sum;
    for (int i = 1; i < n; ++i) c[i * stride] += z * c[(i - 1) * stride];
    c[(n - 1) * stride] = (z / (z - (T)1)) * c[(n - 1) * stride];
    for (int i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
}

// rows pass through an LDS tile: block = 64 threads handles 32 rows of one image
template <typename T, typename TIN>
__global__ void __launch_bounds__(64)
k_pm_prefilter_rows(const TIN *__restrict__ imgs, const int *__restrict__ gather, T *__restrict__ coefs, int D,
                    int TR, const int *__restrict__ count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    const int tilesPerImg = (D + TR - 1) / TR;
    const int slot = blockIdx.x / tilesPerImg;
    if (count && slot >= *count) return;
    const int row0 = (blockIdx.x - slot * tilesPerImg) * TR;
    const int src = gather ? gather[slot] : slot;
    const int nrow = min(TR, D - row0);
    const int ld = D + 1;
    const TIN *in = imgs + (size_t)src * D * D + (size_t)row0 * D;
    for (int i = threadIdx.x; i < nrow * D; i += 64) {
        const int r = i / D, c = i - r * D;
        s[r * ld + c] = (T)in[i];
    }
    __syncthreads();
    if ((int)threadIdx.x < nrow) d_prefilter_line(s + threadIdx.x * ld, D, 1);
    __syncthreads();
    T *out = coefs + (size_t)slot * D * D + (size_t)row0 * D;
    for (int i = threadIdx.x; i < nrow * D; i += 64) {
        const int r = i / D, c = i - r * D;
        out[i] = s[r * ld + c];
    }
}

template <typename T>
__global__ void k_pm_prefilter_cols(T *__restrict__ coefs, int D, int nslots, const int *__restrict__ count)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = gid / D, x = gid - slot * D;
    if (slot >= nslots) return;
    if (count && slot >= *count) return;
    d_prefilter_line(coefs + (size_t)slot * D * D + x, D, D);
}

// ---- cubic B-spline prefilter in fp32 as a convolution ------------------------------------------------
// The recursive filter (pole z = sqrt(3)-2, half-sample-symmetric boundary; k_pm_prefilter_rows/cols)
// is, exactly, the convolution of the mirror-extended samples with h[j] = sqrt(3) z^|j|. The recursion is
// a 2*D-step dependent chain per line (latency-bound: 1.5 us per 256-px image); z^17 < 2e-10 is below
// fp32 resolution, so the fp32 users (coarse matching pass, image shifts before gridding) use the 33-tap
// form, every output independent, eight outputs per thread from one 40-sample window. The fp64 paths
// (reference library, re-scoring) use the 65-tap form below; the gallery projector keeps the recursion.
#define XH_FIR_K 16
#ifndef XH_FIR_V
#define XH_FIR_V 8
#endif
struct XhFir { float h[XH_FIR_K + 1]; };
template <bool COLS>
__global__ void __launch_bounds__(256)
k_pm_prefilter_fir(const float *__restrict__ in, float *__restrict__ out, int D, XhFir F)
{
    // thread <-> XH_FIR_V consecutive outputs along the filtered axis; neighbouring threads are neighbours
    // along x for the column pass (coalesced rows) and along the line for the row pass. blockIdx.y: image
    // (32-bit index arithmetic throughout: a 64-bit division costs more than the filter itself)
    const unsigned segs = (D + XH_FIR_V - 1) / XH_FIR_V;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)D * segs) return;
    int x0, y0;
    if (COLS) { const unsigned q = t / (unsigned)D; x0 = t - q * D; y0 = q * XH_FIR_V; }
    else { const unsigned q = t / segs; x0 = (t - q * segs) * XH_FIR_V; y0 = q; }
    const size_t base = (size_t)blockIdx.y * D * D;
    const float *src = in + base;
    float w[XH_FIR_V + 2 * XH_FIR_K];
    const bool vec = !COLS && (D & 3) == 0;
    if (vec && x0 >= XH_FIR_K && x0 + XH_FIR_V + XH_FIR_K <= D) {
        // interior of a row: the window is 16-byte aligned (x0 and XH_FIR_K are multiples of 4), ten float4 loads
        const float4 *v = reinterpret_cast<const float4 *>(src + (size_t)y0 * D + x0 - XH_FIR_K);
#pragma unroll
        for (int i = 0; i < (XH_FIR_V + 2 * XH_FIR_K) / 4; ++i) {
            const float4 q = v[i];
            w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
        }
    } else
#pragma unroll
    for (int i = 0; i < XH_FIR_V + 2 * XH_FIR_K; ++i) {
        int p = (COLS ? y0 : x0) + i - XH_FIR_K;
        // half-sample-symmetric extension: -1-i <-> i, D+i <-> D-1-i (repeated for tiny images)
        while (p < 0 || p >= D) p = p < 0 ? -1 - p : 2 * D - 1 - p;
        w[i] = COLS ? src[(size_t)p * D + x0] : src[(size_t)y0 * D + p];
    }
    float res[XH_FIR_V];
#pragma unroll
    for (int o = 0; o < XH_FIR_V; ++o) {
        float acc = F.h[0] * w[o + XH_FIR_K];
#pragma unroll
        for (int j = 1; j <= XH_FIR_K; ++j) acc += F.h[j] * (w[o + XH_FIR_K - j] + w[o + XH_FIR_K + j]);
        res[o] = acc;
    }
    if (vec && x0 + XH_FIR_V <= D) {
        float4 *d = reinterpret_cast<float4 *>(out + base + (size_t)y0 * D + x0);
#pragma unroll
        for (int i = 0; i < XH_FIR_V / 4; ++i) d[i] = make_float4(res[4 * i], res[4 * i + 1], res[4 * i + 2], res[4 * i + 3]);
        return;
    }
#pragma unroll
    for (int o = 0; o < XH_FIR_V; ++o) {
        const int q = (COLS ? y0 : x0) + o;
        if (q >= D) break;
        if (COLS) out[base + (size_t)q * D + x0] = res[o];
        else out[base + (size_t)y0 * D + q] = res[o];
    }
}

// The same convolution in fp64 for the re-scoring path and the reference bank: z^33 = 1.3e-19 is below double resolution,
// so 65 taps reproduce the recursion to rounding (the recursion's own 2*D-step chain carries that much error), with every
// output independent. in: float or double images [src][D][D] (gather: source image of slot, nullable; count: number of
// live slots on the device, nullable); out: [slot][D][D] double. One thread per XH_FIR64_V outputs.
#define XH_FIR64_K 32
#define XH_FIR64_V 8
struct XhFir64 { double h[XH_FIR64_K + 1]; };
template <bool COLS, typename TIN>
__global__ void __launch_bounds__(256)
k_pm_prefilter_fir64(const TIN *__restrict__ in, double *__restrict__ out, int D, size_t nvec, XhFir64 F,
                     const int *__restrict__ gather, const int *__restrict__ count)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec) return;
    const int segs = (D + XH_FIR64_V - 1) / XH_FIR64_V;
    int x0, y0;
    size_t slot;
    if (COLS) { x0 = t % D; y0 = (int)((t / D) % segs) * XH_FIR64_V; slot = t / ((size_t)D * segs); }
    else { x0 = (int)(t % segs) * XH_FIR64_V; y0 = (int)((t / segs) % D); slot = t / ((size_t)segs * D); }
    if (count && (int)slot >= *count) return;
    const TIN *src = in + (size_t)(gather ? gather[slot] : (int)slot) * D * D;
    double w[XH_FIR64_V + 2 * XH_FIR64_K];
#pragma unroll
    for (int i = 0; i < XH_FIR64_V + 2 * XH_FIR64_K; ++i) {
        int p = (COLS ? y0 : x0) + i - XH_FIR64_K;
        // half-sample-symmetric extension: -1-i <-> i, D+i <-> D-1-i (repeated for tiny images)
        while (p < 0 || p >= D) p = p < 0 ? -1 - p : 2 * D - 1 - p;
        w[i] = (double)(COLS ? src[(size_t)p * D + x0] : src[(size_t)y0 * D + p]);
    }
#pragma unroll
    for (int o = 0; o < XH_FIR64_V; ++o) {
        const int q = (COLS ? y0 : x0) + o;
        if (q >= D) break;
        double acc = F.h[0] * w[o + XH_FIR64_K];
#pragma unroll
        for (int j = 1; j <= XH_FIR64_K; ++j) acc += F.h[j] * (w[o + XH_FIR64_K - j] + w[o + XH_FIR64_K + j]);
        if (COLS) out[slot * D * D + (size_t)q * D + x0] = acc;
        else out[slot * D * D + (size_t)y0 * D + q] = acc;
    }
}

// Both fp64 passes in one kernel (the re-scored particles' prefilter was two kernels with a D x D double intermediate per slot): a
// block owns 8 rows x 128 columns of a slot's image, filters down the columns (thread <-> column, the 32 either side too) into an
// LDS tile and along the rows out of it; same taps, same mirror extension.  blockIdx.y: slot (gather / count as above).
#define XH_FIR64_TW 128
template <typename TIN>
__global__ void __launch_bounds__(256)
k_pm_prefilter_fir64_2d(const TIN *__restrict__ in, double *__restrict__ out, int D, int tilesX, XhFir64 F, const int *__restrict__ gather,
                        const int *__restrict__ count)
{
    constexpr int K = XH_FIR64_K, V = XH_FIR64_V, TW = XH_FIR64_TW;
    __shared__ double tile[V][TW + 2 * K];
    const int slot = blockIdx.y;
    if (count && slot >= *count) return;
    const TIN *src = in + (size_t)(gather ? gather[slot] : slot) * D * D;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * TW, y0 = ty * V;
    const int xx = threadIdx.x;
    if (xx < TW + 2 * K && x0 + xx - K < D + K) {
        int p = x0 + xx - K;
        while (p < 0 || p >= D) p = p < 0 ? -1 - p : 2 * D - 1 - p;
        double w[V + 2 * K];
#pragma unroll
        for (int i = 0; i < V + 2 * K; ++i) {
            int q = y0 + i - K;
            while (q < 0 || q >= D) q = q < 0 ? -1 - q : 2 * D - 1 - q;
            w[i] = (double)src[(size_t)q * D + p];
        }
#pragma unroll
        for (int o = 0; o < V; ++o) {
            double acc = F.h[0] * w[o + K];
#pragma unroll
            for (int j = 1; j <= K; ++j) acc += F.h[j] * (w[o + K - j] + w[o + K + j]);
            tile[o][xx] = acc;
        }
    }
    __syncthreads();
    // rows: thread <-> 4 consecutive outputs of one tile row
    const int r = threadIdx.x / (TW / 4), seg = threadIdx.x - r * (TW / 4);
    const int xo = x0 + seg * 4, y = y0 + r;
    if (y >= D || xo >= D) return;
    double w[4 + 2 * K];
#pragma unroll
    for (int i = 0; i < 4 + 2 * K; ++i) w[i] = tile[r][seg * 4 + i];
    double *dst = out + (size_t)slot * D * D + (size_t)y * D + xo;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        double acc = F.h[0] * w[o + K];
#pragma unroll
        for (int j = 1; j <= K; ++j) acc += F.h[j] * (w[o + K - j] + w[o + K + j]);
        if (xo + o < D) dst[o] = acc;
    }
}

static inline XhFir64 xh_fir64_taps()
{
    XhFir64 F;
    const long double z = sqrtl(3.0L) - 2.0L;
    for (int j = 0; j <= XH_FIR64_K; ++j) F.h[j] = (double)(sqrtl(3.0L) * powl(z, j));
    return F;
}

static inline XhFir xh_fir_taps()
{
    XhFir F;
    const double z = sqrt(3.0) - 2.0;
    for (int j = 0; j <= XH_FIR_K; ++j) F.h[j] = (float)(sqrt(3.0) * pow(z, j));
    return F;
}

// Both passes in one kernel: a block owns XH_FIR2D_V rows x XH_FIR_TW columns of an image. Its threads first filter down the
// columns (thread <-> column, 40-row window, coalesced; the 16 columns either side of the tile too, mirrored at the image
// border) into an LDS tile, then along the rows out of LDS (thread <-> 8 outputs of one row). The image is read once and
// written once; the two-kernel form above wrote and re-read the row-filtered intermediate.
#define XH_FIR_TW 256
#ifndef XH_FIR2D_V
#define XH_FIR2D_V 32        // tile rows of the two-pass kernel: 64 rows read per 32 written (8: 40 per 8; prep32 4.5 -> 4.2 ms per 4096 particles)
#endif
template <int TW>
__global__ void __launch_bounds__(256)
k_pm_prefilter_fir2d(const float *__restrict__ in, float *__restrict__ out, int D, int tilesX, XhFir F)
{
    __shared__ __align__(16) float tile[XH_FIR2D_V][XH_FIR_TW + 2 * XH_FIR_K];
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * XH_FIR_TW, y0 = ty * XH_FIR2D_V;
    const size_t base = (size_t)blockIdx.y * D * D;
    const float *src = in + base;
    for (int xx = threadIdx.x; xx < XH_FIR_TW + 2 * XH_FIR_K; xx += 256) {
        int p = x0 + xx - XH_FIR_K;
        if (p >= D + XH_FIR_K) break;                   // beyond the halo of the last, partial tile
        // half-sample-symmetric extension: -1-i <-> i, D+i <-> D-1-i (repeated for tiny images)
        while (p < 0 || p >= D) p = p < 0 ? -1 - p : 2 * D - 1 - p;
        float w[XH_FIR2D_V + 2 * XH_FIR_K];
#pragma unroll
        for (int i = 0; i < XH_FIR2D_V + 2 * XH_FIR_K; ++i) {
            int q = y0 + i - XH_FIR_K;
            while (q < 0 || q >= D) q = q < 0 ? -1 - q : 2 * D - 1 - q;
            w[i] = src[(size_t)q * D + p];
        }
#pragma unroll
        for (int o = 0; o < XH_FIR2D_V; ++o) {
            float acc = F.h[0] * w[o + XH_FIR_K];
#pragma unroll
            for (int j = 1; j <= XH_FIR_K; ++j) acc += F.h[j] * (w[o + XH_FIR_K - j] + w[o + XH_FIR_K + j]);
            tile[o][xx] = acc;
        }
    }
    __syncthreads();
    // rows: 256 threads cover 256 / (TW / 8) = 8 tile rows at a time
    constexpr int RPP = 256 / (XH_FIR_TW / 8);
    const int seg = threadIdx.x % (XH_FIR_TW / 8);
    const int xo = x0 + seg * 8;
    if (xo >= D) return;
#pragma unroll
    for (int r = threadIdx.x / (XH_FIR_TW / 8); r < XH_FIR2D_V; r += RPP) {
        const int y = y0 + r;
        if (y >= D) break;
        float w[8 + 2 * XH_FIR_K];
        const float4 *t4 = reinterpret_cast<const float4 *>(&tile[r][seg * 8]);
#pragma unroll
        for (int i = 0; i < (8 + 2 * XH_FIR_K) / 4; ++i) {
            const float4 q = t4[i];
            w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
        }
        float res[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float acc = F.h[0] * w[o + XH_FIR_K];
#pragma unroll
            for (int j = 1; j <= XH_FIR_K; ++j) acc += F.h[j] * (w[o + XH_FIR_K - j] + w[o + XH_FIR_K + j]);
            res[o] = acc;
        }
        float *dst = out + base + (size_t)y * D + xo;
        if ((D & 3) == 0 && xo + 8 <= D) {
            reinterpret_cast<float4 *>(dst)[0] = make_float4(res[0], res[1], res[2], res[3]);
            reinterpret_cast<float4 *>(dst)[1] = make_float4(res[4], res[5], res[6], res[7]);
        } else
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (xo + o < D) dst[o] = res[o];
    }
}

// ---- the same filter in its recursive form, tile by tile ---------------------------------------------------------------
// The 33-tap convolution costs 33 operations per output and axis (k_pm_prefilter_fir2d: 66 vector instructions per pixel, its
// row pass reading LDS with 32-byte lane strides). The recursion costs three -- c+[k] = s[k] + z c+[k-1] forwards,
// c[k] = z (c[k+1] - c+[k]) backwards -- and a thread may enter it anywhere: whatever it starts from is forgotten as z^k
// (0.268^14 = 1e-8, below fp32 resolution), so XH_REC_K = 14 warm-up samples either side of a run of 32 outputs give
// (32 + 28) + 2 (32 + 14) + 32 = 184 operations per 32 outputs, 5.8 per pixel and axis, on the same mirror-extended samples
// as the convolution (its own truncation is z^17).  A block owns 32 rows x 256 columns: down the columns from global memory
// (thread <-> column, coalesced) into an LDS tile, along the rows out of it (thread <-> (row, 32 columns); rows 324 floats
// apart: the sixteen lanes of a ds_read_b128 sit on sixteen different rows = all 64 banks), the results back into the tile
// and out as whole rows (one kilobyte per wave and store instruction).
#define XH_REC_K 14
#define XH_REC_V 32
#define XH_REC_S 324
template <int TW>
__global__ void __launch_bounds__(256)
k_pm_prefilter_rec2d(const float *__restrict__ in, float *__restrict__ out, int D, int tilesX)
{
    constexpr int K = XH_REC_K, V = XH_REC_V, S = XH_REC_S, NW = V + 2 * K;
    static_assert(S >= TW + 2 * K && S % 4 == 0 && S % 64 == 4 && TW == 256, "tile layout");
    __shared__ __align__(16) float tile[V * S];
    const float z = -0.26794919243112270647f, zend = z / (z - 1.f);
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * TW, y0 = ty * V;
    const size_t base = (size_t)blockIdx.y * D * D;
    const float *src = in + base;
    // the tile spans the image's rows: the halo columns are mirror images of columns the block filters anyway
    const bool full = tilesX == 1 && D >= 2 * K;
    auto refl = [D](int a) { const int b = a < 0 ? -1 - a : (a >= D ? 2 * D - 1 - a : a); return min(max(b, 0), D - 1); };
    // a tile the image does not fill: the row pass's windows reach columns no thread filters; whatever the LDS held there (NaN
    // bit patterns included) would enter the backward recursion, so they start as zeros
    if (x0 + TW > D) {
        for (int e = tid; e < V * S; e += 256) tile[e] = 0.f;
        __syncthreads();
    }
    // (full: thread <-> image column, one round; otherwise the halo columns either side are filtered too, a second round for 28 lanes)
    const int xbeg = full ? K : 0, xend = full ? K + min(D, TW) : TW + 2 * K;
    for (int xx = xbeg + tid; xx < xend; xx += 256) {
        int p = x0 + xx - K;
        if (p >= D + K) break;                          // beyond the halo of the last, partial tile
        // half-sample-symmetric extension: -1-i <-> i, D+i <-> D-1-i. One reflection and a clamp: the launcher keeps D >= 32, so a
        // row that would need a second reflection lies 32 or more rows beyond the last output of the tile (z^32 = 5e-19)
        p = refl(p);
        float w[NW];
        if (y0 >= K && y0 + V + K <= D) {                  // (block-uniform) every row of the window lies inside the image
            const float *c0 = src + (unsigned)((y0 - K) * D + p);
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = c0[(unsigned)(i * D)];
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = src[(unsigned)(refl(y0 + i - K) * D + p)];
        }
#pragma unroll
        for (int i = 1; i < NW; ++i) w[i] = __builtin_fmaf(z, w[i - 1], w[i]);
        float a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            if (i < V + K) tile[(i - K) * S + xx] = 6.f * a;
        }
    }
    __syncthreads();
    if (full) {
        for (int e = tid; e < V * 2 * K; e += 256) {
            const int r = e / (2 * K), i = e - r * 2 * K;
            if (i < K) tile[r * S + K - 1 - i] = tile[r * S + K + i];
            else tile[r * S + K + D + (i - K)] = tile[r * S + K + D - 1 - (i - K)];
        }
        __syncthreads();
    }
    // rows: thread <-> (row r, columns 32 sg .. 32 sg + 31 of the tile); its window starts K columns earlier = tile column 32 sg
    const int r = tid & (V - 1), sg = tid >> 5;
    float w[NW];
    {
        const float4 *t4 = reinterpret_cast<const float4 *>(&tile[r * S + 32 * sg]);
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const float4 q = t4[i];
            w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
        }
    }
#pragma unroll
    for (int i = 1; i < NW; ++i) w[i] = __builtin_fmaf(z, w[i - 1], w[i]);
    {
        float a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            w[i] = 6.f * a;
        }
    }
    __syncthreads();                                    // every window has been read
    {
        float4 *t4 = reinterpret_cast<float4 *>(&tile[r * S + 32 * sg]);
#pragma unroll
        for (int i = 0; i < 8; ++i) t4[i] = make_float4(w[K + 4 * i], w[K + 4 * i + 1], w[K + 4 * i + 2], w[K + 4 * i + 3]);
    }
    __syncthreads();
    const bool vec = (D & 3) == 0;
    for (int e = tid; e < V * (TW / 4); e += 256) {
        const int rr = e / (TW / 4), c4 = e - rr * (TW / 4);
        const int y = y0 + rr, x = x0 + 4 * c4;
        if (y >= D || x >= D) continue;
        const float4 q = *reinterpret_cast<const float4 *>(&tile[rr * S + 4 * c4]);
        float *dst = out + base + (size_t)y * D + x;
        if (vec) *reinterpret_cast<float4 *>(dst) = q;
        else { dst[0] = q.x; if (x + 1 < D) dst[1] = q.y; if (x + 2 < D) dst[2] = q.z; if (x + 3 < D) dst[3] = q.w; }
    }
}

// The recursion in double for the re-scored particles (k_pm_prefilter_fir64_2d: 65 taps per output and axis).  z^28 = 1e-16: a run
// entered 28 samples early has forgotten its start to double resolution, which is also what the reference's own recursion carries
// as rounding over a 256-sample line.  16 rows x 256 columns per block, thread <-> column, then thread <-> (row, 16 columns); LDS rows
// 322 doubles apart (the sixteen lanes of a ds_read_b128 on sixteen rows = all banks).  in: float or double images [src][D][D]
// (gather / count as for the convolution form), out [slot][D][D] double; blockIdx.y: slot.
#define XH_REC64_K 28
#define XH_REC64_V 16
#define XH_REC64_S 322
template <typename TIN>
__global__ void __launch_bounds__(256)
k_pm_prefilter_rec64_2d(const TIN *__restrict__ in, double *__restrict__ out, int D, int tilesX, const int *__restrict__ gather,
                        const int *__restrict__ count)
{
    constexpr int K = XH_REC64_K, V = XH_REC64_V, S = XH_REC64_S, NW = V + 2 * K, TW = 256;
    static_assert(S >= TW + 2 * K && S % 2 == 0 && S % 32 == 2, "tile layout");
    __shared__ __align__(16) double tile[V * S];
    const int slot = blockIdx.y;
    if (count && slot >= *count) return;
    const double z = -0.26794919243112270647, zend = z / (z - 1.0);
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x0 = tx * TW, y0 = ty * V;
    const TIN *src = in + (size_t)(gather ? gather[slot] : slot) * D * D;
    const bool full = tilesX == 1 && D >= 2 * K;
    auto refl = [D](int a) { const int b = a < 0 ? -1 - a : (a >= D ? 2 * D - 1 - a : a); return min(max(b, 0), D - 1); };
    if (x0 + TW > D) {
        for (int e = tid; e < V * S; e += 256) tile[e] = 0.0;
        __syncthreads();
    }
    const int xbeg = full ? K : 0, xend = full ? K + min(D, TW) : TW + 2 * K;
    for (int xx = xbeg + tid; xx < xend; xx += 256) {
        int p = x0 + xx - K;
        if (p >= D + K) break;
        p = refl(p);
        double w[NW];
        if (y0 >= K && y0 + V + K <= D) {
            const TIN *c0 = src + (unsigned)((y0 - K) * D + p);
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = (double)c0[(unsigned)(i * D)];
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = (double)src[(unsigned)(refl(y0 + i - K) * D + p)];
        }
#pragma unroll
        for (int i = 1; i < NW; ++i) w[i] = fma(z, w[i - 1], w[i]);
        double a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            if (i < V + K) tile[(i - K) * S + xx] = 6.0 * a;
        }
    }
    __syncthreads();
    if (full) {
        for (int e = tid; e < V * 2 * K; e += 256) {
            const int r = e / (2 * K), i = e - r * 2 * K;
            if (i < K) tile[r * S + K - 1 - i] = tile[r * S + K + i];
            else tile[r * S + K + D + (i - K)] = tile[r * S + K + D - 1 - (i - K)];
        }
        __syncthreads();
    }
    const int r = tid & (V - 1), sg = tid >> 4;
    double w[NW];
    {
        const double2 *t2 = reinterpret_cast<const double2 *>(&tile[r * S + 16 * sg]);
#pragma unroll
        for (int i = 0; i < NW / 2; ++i) { const double2 q = t2[i]; w[2 * i] = q.x; w[2 * i + 1] = q.y; }
    }
#pragma unroll
    for (int i = 1; i < NW; ++i) w[i] = fma(z, w[i - 1], w[i]);
    {
        double a = zend * w[NW - 1];
#pragma unroll
        for (int i = NW - 2; i >= K; --i) {
            a = z * (a - w[i]);
            w[i] = 6.0 * a;
        }
    }
    __syncthreads();
    {
        double2 *t2 = reinterpret_cast<double2 *>(&tile[r * S + 16 * sg]);
#pragma unroll
        for (int i = 0; i < 8; ++i) t2[i] = make_double2(w[K + 2 * i], w[K + 2 * i + 1]);
    }
    __syncthreads();
    double *dstImg = out + (size_t)slot * D * D;
    const bool vec = (D & 1) == 0;
    for (int e = tid; e < V * (TW / 2); e += 256) {
        const int rr = e / (TW / 2), c2 = e - rr * (TW / 2);
        const int y = y0 + rr, x = x0 + 2 * c2;
        if (y >= D || x >= D) continue;
        const double2 q = *reinterpret_cast<const double2 *>(&tile[rr * S + 2 * c2]);
        double *dst = dstImg + (size_t)y * D + x;
        if (vec) *reinterpret_cast<double2 *>(dst) = q;
        else { dst[0] = q.x; if (x + 1 < D) dst[1] = q.y; }
    }
}

// which form xh_prefilter_fir_launch runs: 1 the recursion tile by tile (k_pm_prefilter_rec2d), 0 the 33-tap convolution
// (k_pm_prefilter_fir2d); XH_PREFILTER_FORM in the environment picks for A/B runs
static inline int xh_prefilter_form()
{
    static const int form = [] { const char *e = xh_debug_env("XH_PREFILTER_FORM"); return e ? atoi(e) : 1; }();
    return form;
}

// n images [n][D][D]: in -> out (must not alias)
static inline void xh_prefilter_fir_launch(hipStream_t stream, const float *in, float *out, int D, size_t n)
{
    if (xh_prefilter_form() == 1) {
        const int tilesX = (D + XH_FIR_TW - 1) / XH_FIR_TW, tilesY = (D + XH_REC_V - 1) / XH_REC_V;
        const size_t img = (size_t)D * D;
        for (size_t i0 = 0; i0 < n; i0 += 65535) {
            const unsigned m = (unsigned)std::min<size_t>(65535, n - i0);
            hipLaunchKernelGGL((k_pm_prefilter_rec2d<XH_FIR_TW>), dim3(tilesX * tilesY, m), dim3(256), 0, stream, in + i0 * img, out + i0 * img, D, tilesX);
        }
        return;
    }
    const XhFir F = xh_fir_taps();
    const int tilesX = (D + XH_FIR_TW - 1) / XH_FIR_TW, tilesY = (D + XH_FIR2D_V - 1) / XH_FIR2D_V;
    const size_t img = (size_t)D * D;
    for (size_t i0 = 0; i0 < n; i0 += 65535) {
        const unsigned m = (unsigned)std::min<size_t>(65535, n - i0);
        hipLaunchKernelGGL((k_pm_prefilter_fir2d<XH_FIR_TW>), dim3(tilesX * tilesY, m), dim3(256), 0, stream, in + i0 * img, out + i0 * img, D, tilesX, F);
    }
}

template <typename T> __device__ __forceinline__ T d_bspline03(T x)
{
    // reconstruction_cuda/cuda_gpu_bilib.cu:16-25
    T a = fabs(x);
    if (a < (T)1) return a * a * (a - (T)2) * (T)0.5 + (T)(2.0 / 3.0);
    if (a < (T)2) { a -= (T)2; return a * a * a * (T)(-1.0 / 6.0); }
    return (T)0;
}

// The four weights d_bspline03(x - (l1 + t)), t = 0..3, of a footprint that starts at l1 = ceil(x - 2): the arguments lie in
// (1, 2], (0, 1], (-1, 0], (-2, -1], so the piece of the spline each of them takes is known and no lane has to branch (the
// generic form costs eight divergent branches per sample). Same expressions, same order as d_bspline03; the selects keep the
// values at the breakpoints (integer x) bit-identical to it.
template <typename T> __device__ __forceinline__ void d_bspline03_w4(T x, int l1, T w[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const T a = fabs(x - (T)(l1 + t));
        const T b = a - (T)2;
        const T outer = b * b * b * (T)(-1.0 / 6.0);
        if (t == 0 || t == 3) w[t] = a < (T)2 ? outer : (T)0;
        else w[t] = a < (T)1 ? a * a * (a - (T)2) * (T)0.5 + (T)(2.0 / 3.0) : outer;
    }
}

// The same four weights for the fp32 coarse passes (S1 polar sampling, S6 rotation), as polynomials of t = x - (l1 + 1) in (0, 1]:
//   w0 = (1 - t)^3 / 6,  w1 = 2/3 - t^2 (2 - t) / 2,  w2 = 2/3 - (1 - t)^2 (1 + t) / 2,  w3 = t^3 / 6
// 14 operations instead of ~32, no selects; they differ from d_bspline03_w4's by float rounding (the coarse passes carry their own
// error margins: tau_rel for the rotational search, s6_eps for the shifts; the fp64 chains keep the expressions above).
__device__ __forceinline__ void d_bspline03_w4_lean(float x, int l1, float w[4])
{
    const float t = x - (float)(l1 + 1), o = 1.f - t;
    const float t2 = t * t, o2 = o * o;
    w[0] = o2 * o * (1.f / 6.f);
    w[3] = t2 * t * (1.f / 6.f);
    w[1] = __builtin_fmaf(t2 * (t - 2.f), 0.5f, 2.f / 3.f);
    w[2] = __builtin_fmaf(o2 * (o - 2.f), 0.5f, 2.f / 3.f);
}
template <typename T> __device__ __forceinline__ void d_bspline03_w4_coarse(T x, int l1, T w[4])
{
    if constexpr (sizeof(T) == 4) d_bspline03_w4_lean(x, l1, w);
    else d_bspline03_w4<T>(x, l1, w);
}

// interpolatedElementBSpline2D degree 3 at logical (x,y); reconstruction_cuda/cuda_gpu_multidim_array.cu:78-157
template <typename T>
__device__ __forceinline__ T d_interp(const T *__restrict__ coef, int D, T x, T y)
{
    const int start = -(D / 2);
    x -= (T)start;
    y -= (T)start;
    const int l1 = (int)ceil(x - (T)2), m1 = (int)ceil(y - (T)2);
    int el[4];
    T wx[4], wy[4];
    d_bspline03_w4<T>(x, l1, wx);
    d_bspline03_w4<T>(y, m1, wy);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int l = l1 + t;
        el[t] = l < 0 ? -l - 1 : (l >= D ? 2 * D - l - 1 : l);
    }
    T columns = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = m1 + t;
        const int em = m < 0 ? -m - 1 : (m >= D ? 2 * D - m - 1 : m);
        const T *ref = coef + (size_t)em * D;
        T rows = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) rows += ref[el[u]] * wx[u];
        columns += rows * wy[t];
    }
    return columns;
}

template <typename T> __device__ __forceinline__ T d_realwrap(T x, T x0, T xF)
{
    if (x >= x0 && x <= xF) return x;
    if (x < x0) return x - (int)((x - x0) / (xF - x0) - 1) * (xF - x0);
    return x - (int)((x - xF) / (xF - x0) + 1) * (xF - x0);
}


#endif
