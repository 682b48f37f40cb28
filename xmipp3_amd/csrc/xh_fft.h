// xh_fft.h -- LDS-resident power-of-two FFT building blocks (gfx950).
//
// Transforms live entirely in LDS; a group of `nth` threads (a whole workgroup, every
// thread calling with the same trip counts so that __syncthreads is uniform) owns `nlines`
// independent lines of length n = 1<<logn stored contiguously at s + line*n.
// Twiddles W[j] = exp(-2*pi*i*j/nW), j < nW/2, come from a table computed on the host in
// double precision (accuracy of the fp32 path matters: the coarse orientation pass is
// only trusted within a measured error margin). nW >= n, both powers of two.
//
//  fft_dif: natural order in  -> bit-reversed order out   (Gentleman-Sande)
//  fft_dit: bit-reversed in   -> natural order out        (Cooley-Tukey)
// Pairing DIF (forward) with DIT (inverse) lets a convolution skip the permutation.
#ifndef XH_FFT_H
#define XH_FFT_H
#include <hip/hip_runtime.h>

template <typename T> struct xh_c2 { T x, y; };
typedef xh_c2<float> xh_cf;
typedef xh_c2<double> xh_cd;

template <typename T> __device__ __forceinline__ xh_c2<T> xh_cmul(xh_c2<T> a, xh_c2<T> b)
{
    xh_c2<T> r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
template <typename T> __device__ __forceinline__ xh_c2<T> xh_cmulc(xh_c2<T> a, xh_c2<T> b)
{  // a * conj(b)
    xh_c2<T> r;
    r.x = a.x * b.x + a.y * b.y;
    r.y = a.y * b.x - a.x * b.y;
    return r;
}

// INV=false: forward kernel exp(-i..); INV=true: conjugate twiddles (un-normalised inverse)
template <typename T, bool INV>
__device__ __forceinline__ void xh_fft_dif(xh_c2<T> *s, int logn, int nlines, const xh_c2<T> *W,
                                           int lognW, int tid, int nth)
{
    const int n = 1 << logn;
    const int halfn = n >> 1;
    const int total = nlines * halfn;
    for (int stage = 0; stage < logn; ++stage) {
        const int lh = logn - 1 - stage;  // log2(half span)
        const int half = 1 << lh;
        for (int b = tid; b < total; b += nth) {
            const int line = b / halfn, bb = b - line * halfn;
            const int grp = bb >> lh, j = bb & (half - 1);
            xh_c2<T> *p = s + (size_t)line * n + (grp << (lh + 1)) + j;
            xh_c2<T> u = p[0], v = p[half];
            xh_c2<T> w = W[j << (stage + (lognW - logn))];
            if (INV) w.y = -w.y;
            xh_c2<T> d;
            d.x = u.x - v.x;
            d.y = u.y - v.y;
            u.x += v.x;
            u.y += v.y;
            p[0] = u;
            p[half] = xh_cmul(d, w);
        }
        __syncthreads();
    }
}

template <typename T, bool INV>
__device__ __forceinline__ void xh_fft_dit(xh_c2<T> *s, int logn, int nlines, const xh_c2<T> *W,
                                           int lognW, int tid, int nth)
{
    const int n = 1 << logn;
    const int halfn = n >> 1;
    const int total = nlines * halfn;
    for (int stage = 0; stage < logn; ++stage) {
        const int lh = stage;
        const int half = 1 << lh;
        for (int b = tid; b < total; b += nth) {
            const int line = b / halfn, bb = b - line * halfn;
            const int grp = bb >> lh, j = bb & (half - 1);
            xh_c2<T> *p = s + (size_t)line * n + (grp << (lh + 1)) + j;
            xh_c2<T> w = W[j << ((logn - 1 - stage) + (lognW - logn))];
            if (INV) w.y = -w.y;
            xh_c2<T> u = p[0], v = xh_cmul(p[half], w);
            xh_c2<T> a, d;
            a.x = u.x + v.x;
            a.y = u.y + v.y;
            d.x = u.x - v.x;
            d.y = u.y - v.y;
            p[0] = a;
            p[half] = d;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int xh_bitrev(int i, int logn) { return (int)(__brev((unsigned)i) >> (32 - logn)); }

#endif
