// xh_fftreg.h -- register-blocked two-pass line FFTs, n = R1*R2 (R in {8,16,32}), fp32 or fp64 (gfx950).
//
// A thread keeps a whole radix-R butterfly in registers; LDS is only the exchange medium between
// the two passes (one round trip per transform instead of log2(n) for the radix-2 kernels of
// xh_fft.h). Forward is decimation in frequency, x[n1*R2+n2] -> X[k1 + R1*k2]; the inverse is its
// mirror image, so neither needs a reordering pass. A workgroup of 256 threads owns LN = 256/max(R1,R2)
// lines; line l lives at s + l*LS, element (a, b) of the R1 x R2 index grid at a*S1 + b (S1 = R2+1:
// padded against bank conflicts); the twiddles W_n^j, j < n, sit behind the lines in LDS.
#ifndef XH_FFTREG_H
#define XH_FFTREG_H
#include "xh_fft.h"

template <typename T, bool INV> __device__ __forceinline__ xh_c2<T> t_mulw(xh_c2<T> a, T wr, T wi)
{
    xh_c2<T> r;
    if (!INV) { r.x = a.x * wr - a.y * wi; r.y = a.x * wi + a.y * wr; }
    else { r.x = a.x * wr + a.y * wi; r.y = a.y * wr - a.x * wi; }
    return r;
}
template <typename T, bool INV> __device__ __forceinline__ xh_c2<T> t_mulmi(xh_c2<T> a)
{
    return INV ? xh_c2<T>{-a.y, a.x} : xh_c2<T>{a.y, -a.x};
}
template <typename T> __device__ __forceinline__ xh_c2<T> t_add(xh_c2<T> a, xh_c2<T> b) { return xh_c2<T>{a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ xh_c2<T> t_sub(xh_c2<T> a, xh_c2<T> b) { return xh_c2<T>{a.x - b.x, a.y - b.y}; }
template <typename T, bool INV> __device__ __forceinline__ void t_fft4(xh_c2<T> &x0, xh_c2<T> &x1, xh_c2<T> &x2, xh_c2<T> &x3)
{
    const xh_c2<T> a = t_add(x0, x2), b = t_sub(x0, x2), c = t_add(x1, x3), d = t_mulmi<T, INV>(t_sub(x1, x3));
    x0 = t_add(a, c); x1 = t_add(b, d); x2 = t_sub(a, c); x3 = t_sub(b, d);
}
template <typename T, bool INV> __device__ __forceinline__ void t_fft8(xh_c2<T> *v)
{
    xh_c2<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    t_fft4<T, INV>(e0, e1, e2, e3);
    t_fft4<T, INV>(o0, o1, o2, o3);
    const T h = (T)0.70710678118654752440;
    o1 = t_mulw<T, INV>(o1, h, -h);
    o2 = t_mulmi<T, INV>(o2);
    o3 = t_mulw<T, INV>(o3, -h, -h);
    v[0] = t_add(e0, o0); v[4] = t_sub(e0, o0);
    v[1] = t_add(e1, o1); v[5] = t_sub(e1, o1);
    v[2] = t_add(e2, o2); v[6] = t_sub(e2, o2);
    v[3] = t_add(e3, o3); v[7] = t_sub(e3, o3);
}
template <typename T, bool INV> __device__ __forceinline__ void t_fft16(xh_c2<T> *v)
{
    xh_c2<T> e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    t_fft8<T, INV>(e);
    t_fft8<T, INV>(o);
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, h = (T)0.70710678118654752440;
    o[1] = t_mulw<T, INV>(o[1], c1, -s1);
    o[2] = t_mulw<T, INV>(o[2], h, -h);
    o[3] = t_mulw<T, INV>(o[3], s1, -c1);
    o[4] = t_mulmi<T, INV>(o[4]);
    o[5] = t_mulw<T, INV>(o[5], -s1, -c1);
    o[6] = t_mulw<T, INV>(o[6], -h, -h);
    o[7] = t_mulw<T, INV>(o[7], -c1, -s1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = t_add(e[i], o[i]); v[i + 8] = t_sub(e[i], o[i]); }
}
template <typename T, bool INV> __device__ __forceinline__ void t_fft32(xh_c2<T> *v)
{
    xh_c2<T> e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    t_fft16<T, INV>(e);
    t_fft16<T, INV>(o);
    // W_32^i = cos(pi i/16) - i sin(pi i/16)
    const T c[16] = {(T)1.0, (T)0.98078528040323044913, (T)0.92387953251128675613, (T)0.83146961230254523708,
                     (T)0.70710678118654752440, (T)0.55557023301960222474, (T)0.38268343236508977173, (T)0.19509032201612826785,
                     (T)0.0, (T)-0.19509032201612826785, (T)-0.38268343236508977173, (T)-0.55557023301960222474,
                     (T)-0.70710678118654752440, (T)-0.83146961230254523708, (T)-0.92387953251128675613, (T)-0.98078528040323044913};
    const T sn[16] = {(T)0.0, (T)0.19509032201612826785, (T)0.38268343236508977173, (T)0.55557023301960222474,
                      (T)0.70710678118654752440, (T)0.83146961230254523708, (T)0.92387953251128675613, (T)0.98078528040323044913,
                      (T)1.0, (T)0.98078528040323044913, (T)0.92387953251128675613, (T)0.83146961230254523708,
                      (T)0.70710678118654752440, (T)0.55557023301960222474, (T)0.38268343236508977173, (T)0.19509032201612826785};
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        if (i == 8) o[i] = t_mulmi<T, INV>(o[i]);
        else o[i] = t_mulw<T, INV>(o[i], c[i], -sn[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = t_add(e[i], o[i]); v[i + 16] = t_sub(e[i], o[i]); }
}
template <typename T, int R, bool INV> __device__ __forceinline__ void t_fftR(xh_c2<T> *v)
{
    if (R == 32) t_fft32<T, INV>(v);
    else if (R == 16) t_fft16<T, INV>(v);
    else t_fft8<T, INV>(v);
}

// geometry of one workgroup: LN lines of D = R1*R2 points, 256 threads; line l lives at s + l*LS,
// element (a, b) of the R1 x R2 index grid at a*S1 + b (S1 = R2+1: padded against bank conflicts)
template <int R1, int R2, typename T = double> struct TrGeom {
    static constexpr int D = R1 * R2;
    static constexpr int RM = R1 > R2 ? R1 : R2;
    static constexpr int LN = 256 / RM;
    static constexpr int S1 = R2 + 1;
    // one element more than the grid needs: R1 * S1 elements are a multiple of 32 banks for every (R1, R2) in use, and the
    // column kernels map neighbouring lanes to neighbouring LINES (same element of 8-32 lines: all on one bank without this)
    static constexpr int LS = R1 * S1 + 1;
    static constexpr size_t smem = sizeof(xh_c2<T>) * ((size_t)LN * LS + D);
};
// forward pass 1 on registers v[n1] (thread = (line, n2)): radix R1, twiddle W^(k1*n2), to LDS
template <int R1, int R2, typename T> __device__ __forceinline__ void tr_fwd1(xh_c2<T> *v, xh_c2<T> *sl, const xh_c2<T> *sW, int n2)
{
    t_fftR<T, R1, false>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) { const xh_c2<T> w = sW[k1 * n2]; v[k1] = t_mulw<T, false>(v[k1], w.x, w.y); }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) sl[k1 * TrGeom<R1, R2, T>::S1 + n2] = v[k1];
}
// forward pass 2 (thread = (line, k1)): from LDS, radix R2 -> v[k2] = X[k1 + R1*k2]
template <int R1, int R2, typename T> __device__ __forceinline__ void tr_fwd2(xh_c2<T> *v, const xh_c2<T> *sl, int k1)
{
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = sl[k1 * TrGeom<R1, R2, T>::S1 + n2];
    t_fftR<T, R2, false>(v);
}
// inverse pass over k2 (thread = (line, k1)) from registers v[k2], conj twiddle, to LDS
template <int R1, int R2, typename T> __device__ __forceinline__ void tr_inv2(xh_c2<T> *v, xh_c2<T> *sl, const xh_c2<T> *sW, int k1)
{
    t_fftR<T, R2, true>(v);
#pragma unroll
    for (int n2 = 1; n2 < R2; ++n2) { const xh_c2<T> w = sW[k1 * n2]; v[n2] = t_mulw<T, true>(v[n2], w.x, w.y); }
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) sl[k1 * TrGeom<R1, R2, T>::S1 + n2] = v[n2];
}
// inverse pass over k1 (thread = (line, n2)) from LDS -> v[n1] = x[n1*R2 + n2] (un-normalised)
template <int R1, int R2, typename T> __device__ __forceinline__ void tr_inv1(xh_c2<T> *v, const xh_c2<T> *sl, int n2)
{
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) v[k1] = sl[k1 * TrGeom<R1, R2, T>::S1 + n2];
    t_fftR<T, R1, true>(v);
}


#endif
