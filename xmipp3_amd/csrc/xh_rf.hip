// xh_rf.hip -- Kaiser-Bessel Fourier gridding on MI355X (gfx950).
//
// Device side of ProgRecFourierAccel (reference: reconstruction/reconstruct_fourier_accel.cpp,
// "RFA").  Built with -ffp-contract=off: the float geometry that decides which voxels a
// projection touches (RFA:440-522,710-763) must round like the reference's scalar code.
//
// Data layout in HBM
//   images      [n][D][D]            float   (caller)
//   spectra     [n][mv][mv/2]        float2  centred half-plane, DC at row mv/2 (RFA:271-298)
//   ctf, mod    [n][mv][mv/2]        float
//   temp        [ V (mv+1)^3 float2 | W (mv+1)^3 float ]  [z][y][x]   (RFA:974-979)
//   cropped     [ V (mv+1)^2 (mv/2+1) float2 | W ... ]    after mirrorAndCrop (RFA:853-887)
//
// Kernels and their bounds (DESIGN.md has the byte counts):
//   k_rf_rows / k_rf_cols   batched 2-D r2c FFT via LDS (pad, CenterFFT and crop fused)   HBM
//   k_rf_ctf                CTF / modulator planes                                         VALU (fp64 sincos)
//   k_rf_insert             gridding: slab traversal + blob gather + 3 float atomics/voxel HBM atomics  <-- dominant
//   k_rf_mirror, k_rf_hermitian, k_rf_weights, k_rf_expand, xh_k_fft_lines (xh_plan.h), k_rf_c2r_window   O(volume) once
#include "xh_common.h"
#include "xh_rf_cell.h"
#include "xh_fft.h"
#include "xh_fftreg.h"
#include "xh_plan.h"
#include "xh_bspline.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <dlfcn.h>
#include <rccl/rccl.h>

#define XH_BLOB_TABLE 10000
#define XG_PAD 6                // zero cells around a packed projection record (xh_rf_grid.h)

namespace {
const double kPI = 3.14159265358979323846;

// ------------------------------------------------------------------ host: Kaiser-Bessel
// data/blobs.cpp:37-92,144-172 with the Numerical-Recipes modified Bessel functions that
// xmippCore provides (in-tree float copies: reconstruction_cuda/cuda_gpu_reconstruct_fourier.cpp:85-148).
double h_bessi0(double x)
{
    double ax = std::fabs(x);
    if (ax < 3.75) {
        double y = (x / 3.75) * (x / 3.75);
        return 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
    }
    double y = 3.75 / ax;
    return (std::exp(ax) / std::sqrt(ax)) *
           (0.39894228 + y * (0.1328592e-1 + y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 + y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 + y * 0.392377e-2))))))));
}
double h_bessi1(double x)
{
    double ax = std::fabs(x), ans;
    if (ax < 3.75) {
        double y = (x / 3.75) * (x / 3.75);
        ans = ax * (0.5 + y * (0.87890594 + y * (0.51498869 + y * (0.15084934 + y * (0.2658733e-1 + y * (0.301532e-2 + y * 0.32411e-3))))));
    } else {
        double y = 3.75 / ax;
        ans = 0.2282967e-1 + y * (-0.2895312e-1 + y * (0.1787654e-1 - y * 0.420059e-2));
        ans = 0.39894228 + y * (-0.3988024e-1 + y * (-0.362018e-2 + y * (0.163801e-2 + y * (-0.1031555e-1 + y * ans))));
        ans *= (std::exp(ax) / std::sqrt(ax));
    }
    return x < 0.0 ? -ans : ans;
}
double h_bessin(int n, double x)
{
    if (n == 0) return h_bessi0(x);
    if (n == 1) return h_bessi1(x);
    if (x == 0) return 0;
    return h_bessin(n - 2, x) - ((2 * (n - 1)) / x) * h_bessin(n - 1, x);
}
double h_kaiser_value(double r, double a, double alpha, int m)
{
    double rda = r / a;
    if (rda > 1.0) return 0.0;
    double rdas = rda * rda;
    double arg = alpha * std::sqrt(1.0 - rdas);
    if (m == 0) return h_bessi0(arg) / h_bessi0(alpha);
    double w = std::sqrt(1.0 - rdas);
    double wp = w;
    for (int i = 1; i < m; ++i) wp *= w;
    if (alpha != 0.0) wp *= h_bessin(m, arg) / h_bessin(m, alpha);
    return wp;
}
double h_bessi_half(int twice_order, double x)  // I_{1/2}, I_{3/2}, I_{5/2}, I_{7/2}
{
    if (x == 0) return 0;
    const double c = std::sqrt(2 / (kPI * x));
    double i05 = c * std::sinh(x);
    double i15 = c * (std::cosh(x) - std::sinh(x) / x);
    if (twice_order == 1) return i05;
    if (twice_order == 3) return i15;
    double i25 = i05 - (3 / x) * i15;
    if (twice_order == 5) return i25;
    return i15 - (5 / x) * i25;
}
double h_bessj_half(int twice_order, double x)  // J_{3/2}, J_{7/2}
{
    if (x == 0) return 0;
    const double c = std::sqrt(2 / (kPI * x));
    if (twice_order == 3) return c * (std::sin(x) / x - std::cos(x));
    return c * ((15 / (x * x * x) - 6 / x) * std::sin(x) - (15 / (x * x) - 1) * std::cos(x));
}
double h_kaiser_fourier(double w, double a, double alpha, int m)
{
    const double t = 2. * kPI * a * w;
    const double sigma = std::sqrt(std::fabs(alpha * alpha - t * t));
    if (m == 2) {
        const double num = std::pow(2. * kPI, 1.5) * std::pow(a, 3.) * std::pow(alpha, 2.);
        const double den = h_bessi0(alpha) * std::pow(sigma, 3.5);
        return num * (t > alpha ? h_bessj_half(7, sigma) : h_bessi_half(7, sigma)) / den;
    }
    const double num = std::pow(2. * kPI, 1.5) * std::pow(a, 3);
    const double den = h_bessi0(alpha) * std::pow(sigma, 1.5);
    return num * (t > alpha ? h_bessj_half(3, sigma) : h_bessi_half(3, sigma)) / den;
}

// xmippCore Euler_angles2matrix (closed form in applications/tests/function_tests/test_geometry_main.cpp:46-65)
__host__ __device__ inline void h_euler(double rot, double tilt, double psi, double *A)
{
    const double a = rot * kPI / 180., b = tilt * kPI / 180., g = psi * kPI / 180.;
    const double ca = std::cos(a), cb = std::cos(b), cg = std::cos(g);
    const double sa = std::sin(a), sb = std::sin(b), sg = std::sin(g);
    const double cc = cb * ca, cs = cb * sa, sc = sb * ca, ss = sb * sa;
    A[0] = cg * cc - sg * sa; A[1] = cg * cs + sg * ca; A[2] = -cg * sb;
    A[3] = -sg * cc - cg * sa; A[4] = -sg * cs + cg * ca; A[5] = sg * sb;
    A[6] = sc; A[7] = ss; A[8] = cb;
}
__host__ __device__ inline void h_inv3(const double *A, double *B)
{
    const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    B[0] = (e * i - f * h) * id; B[1] = (c * h - b * i) * id; B[2] = (b * f - c * e) * id;
    B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
    B[6] = (d * h - e * g) * id; B[7] = (b * g - a * h) * id; B[8] = (a * e - b * d) * id;
}

// One projection x symmetry placement; cf. RecFourierProjectionTraverseSpace
// (reconstruction/reconstruct_fourier_projection_traverse_space.h:37-59)
struct XhSpace {
    float tInv[9];
    float u[3], v[3];
    float p0[3], p4[3];
    int minY, maxY, minZ, maxZ;
    float weight;
    int img;
};

struct f3 { float x, y, z; };
__host__ __device__ inline void h_mul(const float t[9], f3 &p)
{
    float a = t[0] * p.x + t[1] * p.y + t[2] * p.z;
    float b = t[3] * p.x + t[4] * p.y + t[5] * p.z;
    float c = t[6] * p.x + t[7] * p.y + t[8] * p.z;
    p.x = a; p.y = b; p.z = c;
}

// RFA:430-442,492-522,258-269,724-741 in float, same operation order
__host__ __device__ inline void h_make_space(XhSpace &S, const double *A_SL, const double *A_SLInv, int mv, double blobRadius,
                                             bool useFast, float weight, int img)
{
    float tr[9];
    for (int i = 0; i < 9; ++i) { tr[i] = A_SL[i]; S.tInv[i] = A_SLInv[i]; }
    const int imgSizeX = mv / 2, imgSizeY = mv;
    f3 cub[8];
    const float sizeX = imgSizeX, sizeY = imgSizeY, blobSize = useFast ? 0.f : blobRadius;
    const float halfY = sizeY / 2.0f;
    cub[0].x = cub[3].x = cub[4].x = cub[7].x = 0.f - blobSize;
    cub[1].x = cub[2].x = cub[5].x = cub[6].x = sizeX + blobSize;
    cub[0].y = cub[1].y = cub[4].y = cub[5].y = -(halfY + blobSize);
    cub[2].y = cub[3].y = cub[6].y = cub[7].y = halfY + blobSize;
    cub[0].z = cub[1].z = cub[2].z = cub[3].z = 0.f + blobSize;
    cub[4].z = cub[5].z = cub[6].z = cub[7].z = 0.f - blobSize;
    const f3 origin = {mv / 2.f, mv / 2.f, mv / 2.f};
    for (int i = 0; i < 8; ++i) h_mul(tr, cub[i]);
    for (int i = 0; i < 8; ++i) { cub[i].x += origin.x; cub[i].y += origin.y; cub[i].z += origin.z; }
    f3 lo = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    // the reference seeds the upper corner with numeric_limits<float>::min() (RFA:503-504); kept
    f3 hi = {std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min()};
    for (int i = 0; i < 8; ++i) {
        if (lo.x > cub[i].x) lo.x = cub[i].x;
        if (lo.y > cub[i].y) lo.y = cub[i].y;
        if (lo.z > cub[i].z) lo.z = cub[i].z;
        if (hi.x < cub[i].x) hi.x = cub[i].x;
        if (hi.y < cub[i].y) hi.y = cub[i].y;
        if (hi.z < cub[i].z) hi.z = cub[i].z;
    }
    const float mx = mv;
    if (lo.y < 0) lo.y = 0;
    if (lo.z < 0) lo.z = 0;
    if (hi.y > mx) hi.y = mx;
    if (hi.z > mx) hi.z = mx;
    S.u[0] = cub[1].x - cub[0].x; S.u[1] = cub[1].y - cub[0].y; S.u[2] = cub[1].z - cub[0].z;
    S.v[0] = cub[3].x - cub[0].x; S.v[1] = cub[3].y - cub[0].y; S.v[2] = cub[3].z - cub[0].z;
    S.p0[0] = cub[0].x; S.p0[1] = cub[0].y; S.p0[2] = cub[0].z;
    S.p4[0] = cub[4].x; S.p4[1] = cub[4].y; S.p4[2] = cub[4].z;
    S.minZ = (int)std::floor(lo.z);
    S.minY = (int)std::floor(lo.y);
    S.maxZ = (int)std::ceil(hi.z);
    S.maxY = (int)std::ceil(hi.y);
    S.weight = weight;
    S.img = img;
}
// one traverse space from a symmetry matrix and the projection's inverse orientation (RFA:939-966)
__host__ __device__ inline void h_place(XhSpace &S, const double *R, const double *Ainv, int mv, double blobRadius, bool useFast, float w, int img)
{
    double A_SL[9], A_SLInv[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += R[r * 3 + k] * Ainv[k * 3 + c];
            A_SL[r * 3 + c] = acc;
        }
    h_inv3(A_SL, A_SLInv);
    h_make_space(S, A_SL, A_SLInv, mv, blobRadius, useFast, w, img);
}
}  // namespace

struct xh_rf {
    xh_ctx *ctx;
    xh_rf_params p;
    int D, P, mv, sizeX, sizeY;
    float iDeltaSqrt, iDeltaFourier;
    std::vector<float> blobTableSqrt;
    std::vector<double> fourierBlobTable;
    XhBuf d_blob;     // float[10000]
    XhBuf d_twP32;    // float2 twiddles for length P
    XhBuf d_twP64;    // double2 twiddles for length P
    XhPlanBufs<float> planP32;    // length-P line transforms of any P (xh_plan.h)
    XhPlanBufs<double> planP64;
    XhBuf own_temp;   // 3*(mv+1)^3 floats when library-owned
    float *d_temp;    // active temp buffer
    XhBuf d_rows;     // intermediate of the 2-D FFT
    XhBuf d_ctfp;
    XhBuf d_fin;      // finaliser scratch
    XhBuf d_shiftCoef, d_shiftXY;   // xh_rf_shift_images scratch
    XhBuf d_tileCounter;            // ints [32,41): class offsets of the tile list, [128,384): the stream counters
    XhBuf d_cull, d_pack, d_superList, d_superCount, d_superVec;
    XhBuf d_gtiles[2], d_grecs, d_gweights, d_planes, d_spectra;   // d_planes, d_spectra: scratch of xh_rf_insert_images   // k_rf_grid: tile list (16 x 16 x 8 tiles), records, per-image weights
    // pinned staging of the small host arrays (records, shifts, CTF parameters): uploads never wait for the stream
    // pinned staging area of the small per-call host arrays: two halves used in turn, an event per half (recorded after the half's last
    // copy).  Entering a half waits for ITS event -- recorded a whole half ago -- so the host never waits for the copy it has just
    // queued.  (One area with one event made the host wait, at every wrap, until the stream had reached the previous call's copies:
    // with 200-byte CTF records a 4096-projection call wraps every second step, 9 ms of idle device per step in `--mode grid`.)
    unsigned char *h_stage = nullptr;
    size_t stageCap = 0, stageUsed = 0;          // capacity of ONE half, bytes used in the current half
    int stageHalf = 0;
    hipEvent_t stageEv[2] = {nullptr, nullptr};
    bool stagePending[2] = {false, false};
    XhBuf d_sym, d_angles;          // device-side inputs of k_rf_spaces
    bool packCtf = false;           // xh_rf_insert_images: the pack kernel evaluates the CTF of d_ctfp itself
    const float *packImgs = nullptr; // ... and the records come straight from the images (k_rf_colsA + k_rf_rowsB<PACK>)
    int tile_max_spaces;
    int fft_variant;      // 0: register-blocked 2-D FFT of the projections where P allows; 1: radix-2 LDS kernels
    // HIP-event bracket of every gridding-kernel launch (bench.py's roofline), drained lazily
    std::vector<hipEvent_t> evPool;
    size_t evUsed;
    double kernelMs;
    int64_t kernelLaunches;
    double meanFactor2;   // cached mean of sinc^2 over the output window (< 0: not computed yet)
    bool cropped;
    int unit_z = 4;       // depth of a gridding unit (8 x 8 x unit_z voxels per wave): 4 or 8
    int grid_waves = 0;   // waves per CU of the gridding kernel; 0: the default of the unit depth
    int grid_tile_budget = 0;   // tiles a workgroup of the gridding kernel processes before it retires; 0: persistent workgroups, one per CU
    int ntiles[2] = {0, 0};
    int fuse_ctf = 1;     // xh_rf_insert_images: evaluate the CTF inside the pack kernel (0: through planes, for A/B)
    XhBuf d_finSpec, d_finVol, d_finFbt;   // the finaliser's expanded spectrum, output volume and Fourier blob table
    int shift_bands = 1;  // 256-px images shifted band by band out of LDS (k_rf_shift_band; 0: k_rf_shift, for A/B)
    int skip_far_cells = 1;   // the pack kernels skip the record cells no tap can reach (0: every cell written, for A/B)
    int fftSkipR2 = 0x7fffffff;   // set by xh_rf_insert_images* around its own projection FFT: spectra cells beyond that radius are not stored
    int order_spaces = 1; // the traverse spaces of a launch ordered by plane, so that k_rf_grid reuses voxel queues (0: input order, for A/B)
    XhBuf d_spacePos;
    int ctf_fast = 1;     // envelope-free CTFs through d_ctf_pixel_fast (0: the general double-precision formula everywhere, for A/B)
    int records_from_images = 0;   // ... and write the records from the row pass of the FFT (measured slower: profiles/README.md)
};

// Copies a host array to the device behind everything already enqueued, without waiting for the stream: the bytes pass
// through a pinned area owned by the handle, which is only waited for (its last copy, not the stream) when it wraps.
static int stage_upload(xh_rf *rf, void *d_dst, const void *h_src, size_t bytes)
{
    if (bytes == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    const size_t need = (bytes + 255) & ~(size_t)255;
    for (int h = 0; h < 2; ++h)
        if (!rf->stageEv[h]) XH_HIP(hipEventCreateWithFlags(&rf->stageEv[h], hipEventDisableTiming));
    if (rf->stageUsed + need > rf->stageCap) {
        if (need > rf->stageCap) {
            // a larger area: every copy out of the old one must have left it
            for (int h = 0; h < 2; ++h)
                if (rf->stagePending[h]) { XH_HIP(hipEventSynchronize(rf->stageEv[h])); rf->stagePending[h] = false; }
            if (rf->h_stage) XH_HIP(hipHostFree(rf->h_stage));
            rf->h_stage = nullptr;
            rf->stageCap = std::max(need, std::max<size_t>(2 * rf->stageCap, (size_t)8 << 20));
            XH_HIP(hipHostMalloc((void **)&rf->h_stage, 2 * rf->stageCap, hipHostMallocDefault));
            rf->stageHalf = 0;
        } else {
            rf->stageHalf ^= 1;
            if (rf->stagePending[rf->stageHalf]) { XH_HIP(hipEventSynchronize(rf->stageEv[rf->stageHalf])); rf->stagePending[rf->stageHalf] = false; }
        }
        rf->stageUsed = 0;
    }
    unsigned char *slot = rf->h_stage + (size_t)rf->stageHalf * rf->stageCap + rf->stageUsed;
    memcpy(slot, h_src, bytes);
    XH_HIP(hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    rf->stageUsed += need;
    XH_HIP(hipEventRecord(rf->stageEv[rf->stageHalf], ctx->stream));
    rf->stagePending[rf->stageHalf] = true;
    return XH_OK;
}

// =========================================================================== device code
// ---- 2-D r2c FFT of the zero-padded, centred image (RFA:332-345) ------------------------
// pass 1: FFT along x of the D image rows; keeps kx < sizeX. rows[img][y][kx]
__global__ void __launch_bounds__(256)
k_rf_rows(const float *__restrict__ imgs, xh_cf *__restrict__ rows, XhPlan<float> plan,
          int D, int sizeX, int totalLines, int lpb)
{
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    const int P = plan.n, M = 1 << plan.logM;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int line0 = blockIdx.x * lpb;
    const int nl = min(lpb, totalLines - line0);
    for (int i = tid; i < lpb * M; i += nth) s[i] = xh_cf{0.f, 0.f};
    __syncthreads();
    const int half = D / 2;  // -FIRST_XMIPP_INDEX(D)
    for (int i = tid; i < nl * D; i += nth) {
        const int l = i / D, x = i - l * D;
        const int xl = x - half;              // logical coordinate
        const int px = xl < 0 ? xl + P : xl;  // (xl mod P): pad centred + CenterFFT(.,true)
        s[l * M + xh_plan_pos(plan, px)].x = imgs[(size_t)(line0 + l) * D + x];
    }
    __syncthreads();
    xh_plan_exec<float, false>(s, plan, lpb, tid, nth);
    for (int i = tid; i < nl * sizeX; i += nth) {
        const int l = i / sizeX, k = i - l * sizeX;
        rows[(size_t)(line0 + l) * sizeX + k] = s[l * M + k];
    }
}

// pass 2: FFT along y for each kept kx, then cropAndShift (RFA:271-298):
// out[img][myPadI][kx] for rows i < sizeX or i >= P-sizeX with freq^2 <= maxRes^2, scaled 1/P^2
__global__ void __launch_bounds__(256)
k_rf_cols(const xh_cf *__restrict__ rows, xh_cf *__restrict__ out, XhPlan<float> plan,
          int D, int sizeX, double maxResSqr, int lpb)
{
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    const int P = plan.n, M = 1 << plan.logM;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int groupsPerImg = (sizeX + lpb - 1) / lpb;
    const int img = blockIdx.x / groupsPerImg;
    const int kx0 = (blockIdx.x - img * groupsPerImg) * lpb;
    const int nl = min(lpb, sizeX - kx0);
    for (int i = tid; i < lpb * M; i += nth) s[i] = xh_cf{0.f, 0.f};
    __syncthreads();
    const int half = D / 2;
    const xh_cf *src = rows + (size_t)img * D * sizeX;
    for (int i = tid; i < D * nl; i += nth) {
        const int y = i / nl, l = i - y * nl;
        const int yl = y - half;
        const int py = yl < 0 ? yl + P : yl;
        s[l * M + xh_plan_pos(plan, py)] = src[(size_t)y * sizeX + kx0 + l];
    }
    __syncthreads();
    xh_plan_exec<float, false>(s, plan, lpb, tid, nth);
    const float scale = 1.0f / ((float)P * (float)P);
    const int sizeY = 2 * sizeX;
    xh_cf *dst = out + (size_t)img * sizeY * sizeX;
    for (int i = tid; i < sizeY * nl; i += nth) {
        const int r = i / nl, l = i - r * nl;   // r = myPadI
        const int ii = (r >= sizeX) ? r - sizeX : r + P - sizeX;  // FFT row index
        const int j = kx0 + l;
        const double fx = (double)j / (double)P;                  // j <= P/2
        const double fy = (double)(ii <= P / 2 ? ii : ii - P) / (double)P;
        xh_cf v = xh_cf{0.f, 0.f};
        if (!(fx * fx + fy * fy > maxResSqr)) {
            v = s[l * M + ii];
            v.x *= scale;
            v.y *= scale;
        }
        dst[(size_t)r * sizeX + j] = v;
    }
}


// ---- Image::readApplyGeo with only_apply_shifts (RFA:304-323): translation by (shiftX, shiftY)
// with cubic B-spline interpolation and wrapping (xmippCore applyGeometry(BSPLINE3, ..., WRAP)).
// out(x,y) samples the prefiltered input at (x - shiftX, y - shiftY).
// A thread owns XH_SHIFT_V vertically adjacent output pixels: they share the column x, hence the four source columns and
// their weights, and their footprints overlap in all but one row each, so the row sums (the inner loop of
// interpolatedElementBSpline2D) are formed once per source row -- XH_SHIFT_V + 3 of them instead of 4 XH_SHIFT_V -- with the
// expressions and the order d_interp uses: same bits.
#ifndef XH_SHIFT_V
#define XH_SHIFT_V 4
#endif
__global__ void __launch_bounds__(256)
k_rf_shift(const float *__restrict__ coefs, const float *__restrict__ imgs,
           const float2 *__restrict__ shifts, const unsigned char *__restrict__ flips,
           float *__restrict__ out, int D)
{
    const int p = blockIdx.y;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned groups = (D + XH_SHIFT_V - 1) / XH_SHIFT_V;
    if (t >= groups * (unsigned)D) return;
    const int g = t / (unsigned)D, j = t - g * D, i0 = g * XH_SHIFT_V;
    const float2 sh = shifts[p];
    const bool flip = flips && flips[p];
    const size_t base = (size_t)p * D * D;
    if (!flip && sh.x == 0.f && sh.y == 0.f) {
#pragma unroll
        for (int k = 0; k < XH_SHIFT_V; ++k)
            if (i0 + k < D) out[base + (size_t)(i0 + k) * D + j] = imgs[base + (size_t)(i0 + k) * D + j];
        return;
    }
    const int cen = D / 2;
    const float minp = -cen, maxp = D - cen - 1;
    // A = [[+-1,0,sx],[0,1,sy]] (flip negates the first row of the 2x2 part, xmippCore
    // geo2TransformationMatrix); IS_NOT_INV => sample the input at A^-1 (x,y)
    float xp = flip ? sh.x - (float)(j - cen) : (float)(j - cen) - sh.x;
    if (xp < minp - 1e-6f || xp > maxp + 1e-6f) xp = d_realwrap<float>(xp, minp - 0.5f, maxp + 0.5f);
    float yp[XH_SHIFT_V], ys[XH_SHIFT_V];
    int m1[XH_SHIFT_V];
    bool chain = true;
#pragma unroll
    for (int k = 0; k < XH_SHIFT_V; ++k) {
        yp[k] = (float)(i0 + k - cen) - sh.y;
        if (yp[k] < minp - 1e-6f || yp[k] > maxp + 1e-6f) yp[k] = d_realwrap<float>(yp[k], minp - 0.5f, maxp + 0.5f);
        ys[k] = yp[k] - (float)(-cen);                 // d_interp: y -= start
        m1[k] = (int)ceilf(ys[k] - 2.f);
        chain = chain && m1[k] == m1[0] + k;
    }
    const float *cf = coefs + base;
    if (!chain || i0 + XH_SHIFT_V > D) {               // a wrap inside the group (or the last, partial group): pixel by pixel
#pragma unroll
        for (int k = 0; k < XH_SHIFT_V; ++k)
            if (i0 + k < D) out[base + (size_t)(i0 + k) * D + j] = d_interp<float>(cf, D, xp, yp[k]);
        return;
    }
    const float xs = xp - (float)(-cen);
    const int l1 = (int)ceilf(xs - 2.f);
    float wx[4];
    d_bspline03_w4<float>(xs, l1, wx);
    int el[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int l = l1 + u; el[u] = l < 0 ? -l - 1 : (l >= D ? 2 * D - l - 1 : l); }
    float rows[XH_SHIFT_V + 3];
#pragma unroll
    for (int r = 0; r < XH_SHIFT_V + 3; ++r) {
        const int m = m1[0] + r;
        const int em = m < 0 ? -m - 1 : (m >= D ? 2 * D - m - 1 : m);
        const float *ref = cf + (size_t)em * D;
        float acc = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += ref[el[u]] * wx[u];
        rows[r] = acc;
    }
#pragma unroll
    for (int k = 0; k < XH_SHIFT_V; ++k) {
        float wy[4];
        d_bspline03_w4<float>(ys[k], m1[k], wy);
        float columns = 0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) columns += rows[k + tt] * wy[tt];
        out[base + (size_t)(i0 + k) * D + j] = columns;
    }
}

// The same for 256-px images, band by band: a workgroup owns XH_SHB output rows of an image, thread <-> column.  The XH_SHB + 3 rows of
// coefficients a band interpolates from are staged in LDS once (k_rf_shift's threads fetched 7 coefficients per output pixel through
// the vector cache; here 1.2), the row sums are formed out of LDS, same expressions in the same order: the same bits.  A band that
// holds the wrap of the shifted rows falls back to interpolatedElementBSpline2D pixel by pixel, like a group of k_rf_shift does.
#define XH_SHB 16
__global__ void __launch_bounds__(256)
k_rf_shift_band(const float *__restrict__ coefs, const float *__restrict__ imgs, const float2 *__restrict__ shifts,
                const unsigned char *__restrict__ flips, float *__restrict__ out, int D)
{
    __shared__ float sC[XH_SHB + 3][256];
    const int p = blockIdx.y, i0 = blockIdx.x * XH_SHB, j = threadIdx.x;
    const float2 sh = shifts[p];
    const bool flip = flips && flips[p];
    const size_t base = (size_t)p * D * D;
    if (!flip && sh.x == 0.f && sh.y == 0.f) {
#pragma unroll
        for (int k = 0; k < XH_SHB; ++k) out[base + (size_t)(i0 + k) * D + j] = imgs[base + (size_t)(i0 + k) * D + j];
        return;
    }
    const int cen = D / 2;
    const float minp = -cen, maxp = D - cen - 1;
    float xp = flip ? sh.x - (float)(j - cen) : (float)(j - cen) - sh.x;
    if (xp < minp - 1e-6f || xp > maxp + 1e-6f) xp = d_realwrap<float>(xp, minp - 0.5f, maxp + 0.5f);
    float yp[XH_SHB], ys[XH_SHB];
    int m1[XH_SHB];
    bool chain = true;
#pragma unroll
    for (int k = 0; k < XH_SHB; ++k) {
        yp[k] = (float)(i0 + k - cen) - sh.y;
        if (yp[k] < minp - 1e-6f || yp[k] > maxp + 1e-6f) yp[k] = d_realwrap<float>(yp[k], minp - 0.5f, maxp + 0.5f);
        ys[k] = yp[k] - (float)(-cen);
        m1[k] = (int)ceilf(ys[k] - 2.f);
        chain = chain && m1[k] == m1[0] + k;
    }
    const float *cf = coefs + base;
    if (!chain) {                                       // (the same for every thread of the band: the rows depend on the row index only)
#pragma unroll
        for (int k = 0; k < XH_SHB; ++k) out[base + (size_t)(i0 + k) * D + j] = d_interp<float>(cf, D, xp, yp[k]);
        return;
    }
    float st[XH_SHB + 3];
#pragma unroll
    for (int r = 0; r < XH_SHB + 3; ++r) {
        const int m = m1[0] + r;
        const int em = m < 0 ? -m - 1 : (m >= D ? 2 * D - m - 1 : m);
        st[r] = cf[(size_t)em * D + j];
    }
#pragma unroll
    for (int r = 0; r < XH_SHB + 3; ++r) sC[r][j] = st[r];
    __syncthreads();
    const float xs = xp - (float)(-cen);
    const int l1 = (int)ceilf(xs - 2.f);
    float wx[4];
    d_bspline03_w4<float>(xs, l1, wx);
    int el[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int l = l1 + u; el[u] = l < 0 ? -l - 1 : (l >= D ? 2 * D - l - 1 : l); }
    float rows[XH_SHB + 3];
#pragma unroll
    for (int r = 0; r < XH_SHB + 3; ++r) {
        float acc = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += sC[r][el[u]] * wx[u];
        rows[r] = acc;
    }
#pragma unroll
    for (int k = 0; k < XH_SHB; ++k) {
        float wy[4];
        d_bspline03_w4<float>(ys[k], m1[k], wy);
        float columns = 0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) columns += rows[k + tt] * wy[tt];
        out[base + (size_t)(i0 + k) * D + j] = columns;
    }
}

// ---- CTF planes (RFA:548-592; data/ctf.h:452-502,1002-1029; data/ctf.cpp:645-679,1392-1402)
struct XhCtfDev {
    double K1, K2, K3, K5, K6, K7, Ksin, Kcos, rad_azimuth, defocus_average, defocus_deviation;
    double DeltaR, K, envR0, envR1, envR2, phase_shift, VPP_radius;
    // d_ctf_pixel_fast (filled by ctf_params_upload; fast == 0: the general formula only):
    //   Ksin sin(a) - Kcos cos(a) = amp sin(a - phi);  E = E0 + envR2 u^2;  cos / sin of twice the azimuth
    double amp = 0, phi = 0, E0 = 0, c2az = 0, s2az = 0;
    int fast = 0, pad_ = 0;
};
// ---- register-blocked variants of k_rf_rows / k_rf_cols for P = R1*R2 (xh_fftreg.h) ---------------
// Same arithmetic contract (pad about the Xmipp origin, CenterFFT, forward FFT, crop, 1/P^2), one LDS
// round trip per transform instead of log2(P), no zero-fill pass: the padding is known by position.
template <int R1, int R2>
__global__ void __launch_bounds__(256)
k_rf_rows2(const float *__restrict__ imgs, xh_cf *__restrict__ rows, const xh_cf *__restrict__ W, int D, int sizeX, int totalLines)
{
    typedef TrGeom<R1, R2, float> G;
    constexpr int P = G::D;
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    xh_cf *sW = s + (size_t)G::LN * G::LS;
    const int tid = threadIdx.x;
    const int line0 = blockIdx.x * G::LN;
    for (int i = tid; i < P; i += 256) sW[i] = W[i];
    __syncthreads();
    const int half = D / 2;
    xh_cf v[G::RM];
    if (tid < G::LN * R2) {
        const int l = tid / R2, n2 = tid - l * R2;
        const bool live = line0 + l < totalLines;
        const float *src = imgs + (size_t)(line0 + l) * D;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int px = n1 * R2 + n2;          // padded, centred position: logical x = px (px < half) or px - P
            float val = 0.f;
            if (live) {
                if (px < D - half) val = src[px + half];
                else if (px >= P - half) val = src[px - P + half];
            }
            v[n1] = xh_cf{val, 0.f};
        }
        tr_fwd1<R1, R2>(v, s + l * G::LS, sW, n2);
    }
    __syncthreads();
    if (tid < G::LN * R1) {
        const int l = tid / R1, k1 = tid - l * R1;
        tr_fwd2<R1, R2>(v, s + l * G::LS, k1);
        if (line0 + l < totalLines) {
            xh_cf *dst = rows + (size_t)(line0 + l) * sizeX;
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2)
                if (k1 + R1 * k2 < sizeX) dst[k1 + R1 * k2] = v[k2];
        }
    }
}

template <int R1, int R2>
__global__ void __launch_bounds__(256)
k_rf_cols2(const xh_cf *__restrict__ rows, xh_cf *__restrict__ out, const xh_cf *__restrict__ W, int D, int sizeX, double maxResSqr)
{
    typedef TrGeom<R1, R2, float> G;
    constexpr int P = G::D;
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    xh_cf *sW = s + (size_t)G::LN * G::LS;
    const int tid = threadIdx.x;
    const int groupsPerImg = (sizeX + G::LN - 1) / G::LN;
    const int img = blockIdx.x / groupsPerImg;
    const int kx0 = (blockIdx.x - img * groupsPerImg) * G::LN;
    for (int i = tid; i < P; i += 256) sW[i] = W[i];
    __syncthreads();
    const int half = D / 2;
    const xh_cf *src = rows + (size_t)img * D * sizeX;
    xh_cf v[G::RM];
    if (tid < G::LN * R2) {
        const int cl = tid % G::LN, n2 = tid / G::LN;      // neighbouring threads, neighbouring columns
        const bool live = kx0 + cl < sizeX;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int py = n1 * R2 + n2;
            xh_cf val = xh_cf{0.f, 0.f};
            if (live) {
                if (py < D - half) val = src[(size_t)(py + half) * sizeX + kx0 + cl];
                else if (py >= P - half) val = src[(size_t)(py - P + half) * sizeX + kx0 + cl];
            }
            v[n1] = val;
        }
        tr_fwd1<R1, R2>(v, s + cl * G::LS, sW, n2);
    }
    __syncthreads();
    if (tid < G::LN * R1) {
        const int cl = tid % G::LN, k1 = tid / G::LN;
        tr_fwd2<R1, R2>(v, s + cl * G::LS, k1);
        const int j = kx0 + cl;
        if (j < sizeX) {
            const float scale = 1.0f / ((float)P * (float)P);
            const int sizeY = 2 * sizeX;
            xh_cf *dst = out + (size_t)img * sizeY * sizeX;
            const double fx = (double)j / (double)P;                  // j <= P/2
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int ii = k1 + R1 * k2;                           // FFT row index
                int r;                                                 // output row (myPadI)
                if (ii < sizeX) r = ii + sizeX;
                else if (ii >= P - sizeX) r = ii - (P - sizeX);
                else continue;                                         // cropped away (only when P > 2*sizeX)
                const double fy = (double)(ii <= P / 2 ? ii : ii - P) / (double)P;
                xh_cf o = xh_cf{0.f, 0.f};
                if (!(fx * fx + fy * fy > maxResSqr)) { o = v[k2]; o.x *= scale; o.y *= scale; }
                dst[(size_t)r * sizeX + j] = o;
            }
        }
    }
}

// ---- the same transform, columns first (k_rf_colsA), rows last (k_rf_rowsB) ------------------------------------------
// The projections are real: the column pass transforms two image columns per complex line and keeps ky = 0 .. sizeX
// ([m][sizeX + 1][D] complex, as many bytes as the row-first intermediate); the row pass then runs over contiguous lines
// and every transformed line ky gives two output rows, F(kx, ky) and F(kx, -ky) = conj F(-kx, ky), each written as one
// contiguous row -- of the half spectrum (xh_rf_prepare_images) or, with PACK, of the padded records the gridding kernel
// reads, CTF factor and modulator evaluated on the spot (xh_rf_insert_images: the half spectra and the CTF planes are
// never written). Same conventions as k_rf_rows2 / k_rf_cols2: pad about the Xmipp origin, CenterFFT, forward FFT, crop,
// 1/P^2, cut beyond max_resolution.
// where k_rf_colsA<.., LN> stores image column x of a line of T
template <int LN> __device__ __forceinline__ int xh_rf_tpos(int x)
{
    const int xl = x & (4 * LN - 1);
    return (x - xl) + 2 * LN * ((xl >> 1) & 1) + 2 * (xl >> 2) + (xl & 1);
}
template <int R1, int R2>
__global__ void __launch_bounds__(256)
k_rf_colsA(const float *__restrict__ imgs, xh_cf *__restrict__ T, const xh_cf *__restrict__ W, int D, int TD, int sizeX)
{
    typedef TrGeom<R1, R2, float> G;
    constexpr int P = G::D, ZS = P + 1;                      // ZS: line stride of the natural-order copy
    constexpr int NC = 4 * G::LN;                            // image columns of a block: two rounds of LN column pairs
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    xh_cf *sW = s + (size_t)G::LN * G::LS;
    const int tid = threadIdx.x;
    const int groupsPerImg = (D + NC - 1) / NC;
    const int img = blockIdx.x / groupsPerImg;
    const int x0 = (blockIdx.x - img * groupsPerImg) * NC;
    for (int i = tid; i < P; i += 256) sW[i] = W[i];
    const int half = D / 2;
    const float *src = imgs + (size_t)img * D * D;
    xh_cf *dst = T + (size_t)img * (sizeX + 1) * TD;
    // a thread fetches four neighbouring columns of its rows at once (16 bytes; eight threads cover a 128-byte line): the
    // first two are its line of round 0, the other two of round 1
    float4 q[R1];
    const int cl = tid % G::LN, n2 = tid / G::LN;            // neighbouring threads, neighbouring column quadruples
    const int xa = x0 + 4 * cl;
    if (tid < G::LN * R2) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int py = n1 * R2 + n2;                     // padded, centred position along y
            int row = -1;
            if (py < D - half) row = py + half;
            else if (py >= P - half) row = py - P + half;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) {
                const float *r = src + (size_t)row * D + xa;
                if (xa + 3 < D && (D & 3) == 0) val = *reinterpret_cast<const float4 *>(r);
                else { if (xa < D) val.x = r[0]; if (xa + 1 < D) val.y = r[1]; if (xa + 2 < D) val.z = r[2]; if (xa + 3 < D) val.w = r[3]; }
            }
            q[n1] = val;
        }
    }
    for (int round = 0; round < 2; ++round) {
        __syncthreads();                                     // sW ready / the previous round's lines consumed
        xh_cf v[G::RM];
        if (tid < G::LN * R2) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1] = round ? xh_cf{q[n1].z, q[n1].w} : xh_cf{q[n1].x, q[n1].y};
            tr_fwd1<R1, R2>(v, s + cl * G::LS, sW, n2);
        }
        __syncthreads();
        if (tid < G::LN * R1) tr_fwd2<R1, R2>(v, s + (tid % G::LN) * G::LS, tid / G::LN);
        __syncthreads();
        if (tid < G::LN * R1) {
            const int c2 = tid % G::LN, k1 = tid / G::LN;
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) s[c2 * ZS + k1 + R1 * k2] = v[k2];
        }
        __syncthreads();
        // Z = FFT(g_a + i g_b)  ->  G_a[k] = (Z[k] + conj Z[-k]) / 2,  G_b[k] = (Z[k] - conj Z[-k]) / 2i
        // line c2 of this round holds the image columns x0 + 4 c2 + 2 round (+ 1); they are stored round by round (the
        // columns of a line of T are permuted within the block's NC, xh_rf_tpos: k_rf_rowsB reads them back the same way),
        // so that every (ky, round) is one contiguous piece
        for (int it = tid; it < (sizeX + 1) * 2 * G::LN; it += 256) {
            const int c = it % (2 * G::LN), ky = it / (2 * G::LN);
            const xh_cf a = s[(c >> 1) * ZS + ky], bq = s[(c >> 1) * ZS + ((P - ky) & (P - 1))];
            dst[(size_t)ky * TD + x0 + 2 * G::LN * round + c] = (c & 1) ? xh_cf{0.5f * (a.y + bq.y), -0.5f * (a.x - bq.x)} : xh_cf{0.5f * (a.x + bq.x), 0.5f * (a.y - bq.y)};
        }
    }
}

__device__ double d_bessj0(double x)
{
    double ax = fabs(x);
    if (ax < 8.0) {
        double y = x * x;
        double a1 = 57568490574.0 + y * (-13362590354.0 + y * (651619640.7 + y * (-11214424.18 + y * (77392.33017 + y * (-184.9052456)))));
        double a2 = 57568490411.0 + y * (1029532985.0 + y * (9494680.718 + y * (59272.64853 + y * (267.8532712 + y * 1.0))));
        return a1 / a2;
    }
    double z = 8.0 / ax, y = z * z, xx = ax - 0.785398164;
    double a1 = 1.0 + y * (-0.1098628627e-2 + y * (0.2734510407e-4 + y * (-0.2073370639e-5 + y * 0.2093887211e-6)));
    double a2 = -0.1562499995e-1 + y * (0.1430488765e-3 + y * (-0.6911147651e-5 + y * (0.7621095161e-6 - y * 0.934935152e-7)));
    return sqrt(0.636619772 / ax) * (cos(xx) * a1 - z * sin(xx) * a2);
}
// one pixel of preloadCTF (RFA:548-592 / RFG:552-593): CTF factor and modulator
__device__ __forceinline__ void d_ctf_pixel(const XhCtfDev &c, int x, int y, int P, double iTs, double minCTF, int phaseFlipped,
                                            float &ctfOut, float &modOut)
{
    // a power-of-two P divides exactly: the products by 1/P are the quotients, bit for bit
    const bool pow2 = (P & (P - 1)) == 0;
    const float freqY = pow2 ? (y - (P / 2.f)) * (1.0f / (float)P) : (y - (P / 2.f)) / (float)P;
    const double xr = (double)(x <= P / 2 ? x : x - P);
    float freqX = (float)(pow2 ? xr * (1.0 / (double)P) : xr / (double)P);
    const double X = freqX * iTs, Y = freqY * iTs;
    const double u2 = X * X + Y * Y;
    const double u = sqrt(u2);
    const double u4 = u2 * u2;
    // Per-image (block-uniform) shortcuts, each bit-identical to the general formula (ctf.h:376-501):
    // x + 0*cos(.) == x and exp(-0*finite) == 1 exactly, so a non-astigmatic CTF needs no atan2/cos and a
    // CTF without energy-spread / convergence-cone terms no exp.
    double deltaf;
    if (fabs(X) < 1e-6 && fabs(Y) < 1e-6) deltaf = 0;
    else if (c.defocus_deviation == 0) deltaf = c.defocus_average;
    else deltaf = c.defocus_average + c.defocus_deviation * cos(2 * (atan2(Y, X) - c.rad_azimuth));
    double VPP = 0.0;
    if (round(c.VPP_radius * 1000) != 0) VPP = -c.phase_shift * (1 - exp(-u2 / (2 * c.VPP_radius * c.VPP_radius)));
    const double argument = VPP + c.K1 * deltaf * u2 + c.K2 * u4;
    double sine_part, cosine_part;
    sincos(argument, &sine_part, &cosine_part);
    const double Eespr = c.K3 == 0 ? 1.0 : exp(-c.K3 * u4);
    // d_bessj0(0) is the quotient of the two leading coefficients (not 1): the compiler folds the same IEEE division
    const double EdeltaF = c.K5 == 0 ? 57568490574.0 / 57568490411.0 : d_bessj0(c.K5 * u2);
    const double xs = u * c.DeltaR;
    const double EdeltaR = (xs == 0) ? 1.0 : sin(3.14159265358979323846 * xs) / (3.14159265358979323846 * xs);
    const double aux = (c.K7 * u2 * u + deltaf * u);
    const double Ealpha = c.K6 == 0 ? 1.0 : exp(-c.K6 * aux * aux);
    double E = Eespr * EdeltaF * EdeltaR * Ealpha + c.envR0 + c.envR1 * u + c.envR2 * u2;
    if (E < 0) E = 0;
    const double pure = -c.K * (c.Ksin * sine_part - c.Kcos * cosine_part) * E;
    float CTFVal = (float)(c.K * pure);   // getValuePureNoKAt multiplies by K (ctf.h:499-502)
    float modulatorVal = 1.f;
    if (isnan(CTFVal)) {
        if (x == 0 && y == 0) modulatorVal = CTFVal = 1.0f;
        else modulatorVal = CTFVal = 0.0f;
    }
    if (fabs((double)CTFVal) < minCTF) {
        modulatorVal = fabsf(CTFVal);
        CTFVal = (CTFVal >= 0) ? 1.f : -1.f;
    } else CTFVal = (float)(1.0 / (double)CTFVal);
    if (phaseFlipped) CTFVal = fabsf(CTFVal);
    ctfOut = CTFVal;
    modOut = modulatorVal;
}
// The same pixel for the CTFs most particles carry -- no envelope terms, no phase plate (fast != 0) --, an order of
// magnitude cheaper: Ksin sin(a) - Kcos cos(a) is ONE sinusoid amp sin(a - phi); its argument is formed and reduced to
// [-pi/2, pi/2] in double (the argument reaches hundreds of radians), the sine of the reduced argument is a float
// polynomial (relative error 1.2e-7, also next to the zeros of the CTF, where 1 / CTF amplifies absolute errors), the
// astigmatic defocus needs no atan2 / cos: cos(2 (theta - az)) u^2 = (X^2 - Y^2) cos 2az + 2 X Y sin 2az.  The value is a
// float in the reference too (RFA:566); it differs from the general formula's by float rounding (2e-7 relative).  The
// one discontinuous decision, |CTF| < minCTF, is never taken on such a value: within 1e-5 of the threshold (and for
// anything that is not finite) the function returns false and the caller evaluates the general formula.
__device__ __forceinline__ bool d_ctf_pixel_fast(const XhCtfDev &c, int x, int y, int P, double iTs, double minCTF, int phaseFlipped,
                                                 float &ctfOut, float &modOut)
{
    const bool pow2 = (P & (P - 1)) == 0;
    const float freqY = pow2 ? (y - (P / 2.f)) * (1.0f / (float)P) : (y - (P / 2.f)) / (float)P;
    const double xr = (double)(x <= P / 2 ? x : x - P);
    const float freqX = (float)(pow2 ? xr * (1.0 / (double)P) : xr / (double)P);
    const double X = freqX * iTs, Y = freqY * iTs;
    const double X2 = X * X, Y2 = Y * Y;
    const double u2 = X2 + Y2;
    double dfu2;                                             // deltaf u^2
    if (fabs(X) < 1e-6 && fabs(Y) < 1e-6) dfu2 = 0;
    else if (c.defocus_deviation == 0) dfu2 = c.defocus_average * u2;
    else dfu2 = c.defocus_average * u2 + c.defocus_deviation * ((X2 - Y2) * c.c2az + 2.0 * (X * Y) * c.s2az);
    const double a = fma(c.K1, dfu2, fma(c.K2 * u2, u2, -c.phi));
    const double n = rint(a * 0.31830988618379067154);
    double r = fma(-n, 3.141592653589793116, a);
    r = fma(-n, 1.2246467991473532e-16, r);
    const float rf = (float)r, t = rf * rf;
    float p = -2.3866480347578545e-08f;
    p = __builtin_fmaf(p, t, 2.7523994958755793e-06f);
    p = __builtin_fmaf(p, t, -0.00019840836466755718f);
    p = __builtin_fmaf(p, t, 0.008333330973982811f);
    p = __builtin_fmaf(p, t, -0.1666666716337204f);
    float sn = __builtin_fmaf(rf * t, p, rf);
    if (((long long)n) & 1) sn = -sn;
    double E = c.E0 + c.envR2 * u2;
    if (E < 0) E = 0;
    float CTFVal = (float)((-(c.K * c.K) * c.amp * E) * (double)sn);
    const double mag = fabs((double)CTFVal);
    if (!(fabs(mag - minCTF) > 1e-5 * minCTF) || !(mag < 1e30)) return false;       // (NaN compares false: general path)
    float modulatorVal = 1.f;
    if (mag < minCTF) {
        modulatorVal = fabsf(CTFVal);
        CTFVal = (CTFVal >= 0) ? 1.f : -1.f;
    } else CTFVal = 1.0f / CTFVal;
    if (phaseFlipped) CTFVal = fabsf(CTFVal);
    ctfOut = CTFVal;
    modOut = modulatorVal;
    return true;
}
// (a real call: inlined, the general formula's double-precision sincos / exp / Bessel code sets the register budget of every kernel
// that evaluates a CTF -- 178 registers and two waves per SIMD for the row pass of the FFT that packs the records)
__device__ __attribute__((noinline)) void d_ctf_pixel_call(const XhCtfDev &c, int x, int y, int P, double iTs, double minCTF, int phaseFlipped,
                                                           float &ctfOut, float &modOut)
{
    d_ctf_pixel(c, x, y, P, iTs, minCTF, phaseFlipped, ctfOut, modOut);
}
// what the kernels call: the fast form where the image's CTF allows it (block-uniform), the general formula for the rest
template <bool CALL = false>
__device__ __forceinline__ void d_ctf_eval(const XhCtfDev &c, int x, int y, int P, double iTs, double minCTF, int phaseFlipped,
                                           float &ctfOut, float &modOut)
{
    if (c.fast && d_ctf_pixel_fast(c, x, y, P, iTs, minCTF, phaseFlipped, ctfOut, modOut)) return;
    if (CALL) d_ctf_pixel_call(c, x, y, P, iTs, minCTF, phaseFlipped, ctfOut, modOut);
    else d_ctf_pixel(c, x, y, P, iTs, minCTF, phaseFlipped, ctfOut, modOut);
}
// thread per pixel of the rows at or above the DC row; its mirror row (-freqY) is written by the same thread: a
// non-astigmatic CTF depends on (X, Y) through X*X + Y*Y only, so the value is the same bit for bit and is computed once
__global__ void k_rf_ctf(const XhCtfDev *__restrict__ cp, float *__restrict__ ctf, float *__restrict__ mod,
                         int n, int sizeX, int sizeY, int P, double iTs, double minCTF, int phaseFlipped)
{
    const int dc = P / 2;                                   // row of freqY = 0
    const int up = max(sizeY - dc, dc + 1);                 // rows dc .. dc+up-1 cover every pair (dc+k, dc-k)
    const unsigned rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= (unsigned)sizeX * up) return;
    const int img = blockIdx.y;
    const int k = rem / (unsigned)sizeX, x = rem - k * sizeX;
    const XhCtfDev c = cp[img];
    const size_t base = (size_t)img * sizeX * sizeY;
    const int y1 = dc + k, y2 = dc - k;
    float cv = 0.f, mv_ = 0.f;
    bool have = false;
    if (y1 < sizeY) {
        d_ctf_eval(c, x, y1, P, iTs, minCTF, phaseFlipped, cv, mv_);
        ctf[base + (size_t)y1 * sizeX + x] = cv;
        mod[base + (size_t)y1 * sizeX + x] = mv_;
        have = true;
    }
    if (k > 0 && y2 >= 0 && y2 < sizeY) {
        // rows dc+k and dc-k have opposite freqY only for even P (freqY = (y - P/2)/P)
        if (!(have && c.defocus_deviation == 0 && (P & 1) == 0)) d_ctf_eval(c, x, y2, P, iTs, minCTF, phaseFlipped, cv, mv_);
        ctf[base + (size_t)y2 * sizeX + x] = cv;
        mod[base + (size_t)y2 * sizeX + x] = mv_;
    }
}

template <int R1, int R2, bool PACK>
__global__ void __launch_bounds__(256)
k_rf_rowsB(const xh_cf *__restrict__ T, xh_cf *__restrict__ out, XgCell *__restrict__ pk, const XhCtfDev *__restrict__ cp,
           const float *__restrict__ weights, const xh_cf *__restrict__ W, int D, int TD, int sizeX, double maxResSqr, int nlines,
           double iTs, double minCTF, int phaseFlipped, int skipR2)
{
    typedef TrGeom<R1, R2, float> G;
    constexpr int P = G::D, ZS = P + 1;
    constexpr int PAD = PACK ? XG_PAD : 0;
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cf *s = reinterpret_cast<xh_cf *>(smem);
    xh_cf *sW = s + (size_t)G::LN * G::LS;
    const int tid = threadIdx.x;
    const int groupsPerImg = (nlines + G::LN - 1) / G::LN;    // nlines = sizeX + 1 (+ PAD: the zero rows of the frame)
    const int img = blockIdx.x / groupsPerImg;
    const int k0 = (blockIdx.x - img * groupsPerImg) * G::LN;
    for (int i = tid; i < P; i += 256) sW[i] = W[i];
    __syncthreads();
    const int half = D / 2, sizeY = 2 * sizeX;
    const xh_cf *src = T + (size_t)img * (sizeX + 1) * TD;
    xh_cf v[G::RM];
    if (tid < G::LN * R2) {
        const int l = tid / R2, n2 = tid - l * R2;
        const bool live = k0 + l <= sizeX;
        const xh_cf *line = src + (size_t)(k0 + l) * TD;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int px = n1 * R2 + n2;
            xh_cf val = xh_cf{0.f, 0.f};
            if (live) {
                if (px < D - half) val = line[xh_rf_tpos<G::LN>(px + half)];
                else if (px >= P - half) val = line[xh_rf_tpos<G::LN>(px - P + half)];
            }
            v[n1] = val;
        }
        tr_fwd1<R1, R2>(v, s + l * G::LS, sW, n2);
    }
    __syncthreads();
    if (tid < G::LN * R1) {
        const int l = tid / R1, k1 = tid - l * R1;
        tr_fwd2<R1, R2>(v, s + l * G::LS, k1);
    }
    __syncthreads();
    if (tid < G::LN * R1) {
        const int l = tid / R1, k1 = tid - l * R1;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) s[l * ZS + k1 + R1 * k2] = v[k2];
    }
    __syncthreads();
    const float scale = 1.0f / ((float)P * (float)P);
    const int SX = sizeX + 2 * PAD, SY = sizeY + 2 * PAD;
    XhCtfDev par;
    float w = 1.f;
    if (PACK) { par = cp[img]; w = weights ? weights[img] : 1.f; }
    for (int l = 0; l < G::LN; ++l) {
        const int k = k0 + l;
        if (k >= nlines) break;
        const int r1 = sizeX + k, r2 = sizeX - k;            // output rows of ky = k and ky = -k
        const bool row1 = r1 < sizeY + PAD, row2 = k >= 1 && r2 >= -PAD;
        const double fy = (double)k / (double)P;
        if (PACK && tid < 2 * PAD) {                          // the frame cells left and right of the two rows
            const int xc = tid < PAD ? tid : sizeX + tid;
            XgCell *dst = pk + (size_t)img * SX * SY;
            if (row1) xg_put(dst + (size_t)(r1 + PAD) * SX + xc, 0.f, 0.f, 0.f);
            if (row2) xg_put(dst + (size_t)(r2 + PAD) * SX + xc, 0.f, 0.f, 0.f);
        }
        for (int j = tid; j < sizeX; j += 256) {             // thread <-> kx: every wave has the same number of CTF values to find
            const int xc = j + PAD;
            const bool in1 = r1 < sizeY, in2 = k >= 1 && r2 >= 0;
            xh_cf o1 = xh_cf{0.f, 0.f}, o2 = o1;
            if (in1 || in2) {
                const double fx = (double)j / (double)P;     // j <= P/2
                if (!(fx * fx + fy * fy > maxResSqr)) {
                    const xh_cf a = s[l * ZS + j], b = s[l * ZS + ((P - j) & (P - 1))];
                    o1 = xh_cf{a.x * scale, a.y * scale};
                    o2 = xh_cf{b.x * scale, -b.y * scale};
                }
            }
            if constexpr (!PACK) {
                // (skipR2: the caller is the gridding path, which never looks at a pixel beyond every tap's reach, k_rf_pack_grid_ctf)
                if (j * j + k * k > skipR2) continue;
                if (in1) out[((size_t)img * sizeY + r1) * sizeX + j] = o1;
                if (in2) out[((size_t)img * sizeY + r2) * sizeX + j] = o2;
            } else {
                float cv = 0.f, mv_ = 0.f;
                float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
                if (in1) {
                    d_ctf_eval<true>(par, j, r1, P, iTs, minCTF, phaseFlipped, cv, mv_);
                    const float mw = mv_ * w;
                    v1 = make_float4(o1.x * mw * cv, o1.y * mw * cv, mw, 0.f);
                }
                if (in2) {
                    // rows sizeX + k and sizeX - k have opposite freqY (sizeX = P / 2, P even): see k_rf_ctf
                    if (!(in1 && par.defocus_deviation == 0)) d_ctf_eval<true>(par, j, r2, P, iTs, minCTF, phaseFlipped, cv, mv_);
                    const float mw = mv_ * w;
                    v2 = make_float4(o2.x * mw * cv, o2.y * mw * cv, mw, 0.f);
                }
                XgCell *dst = pk + (size_t)img * SX * SY;
                if (row1) xg_put(dst + (size_t)(r1 + PAD) * SX + xc, v1.x, v1.y, v1.z);
                if (row2) xg_put(dst + (size_t)(r2 + PAD) * SX + xc, v2.x, v2.y, v2.z);
            }
        }
    }
}

// ---- gridding (RFA:627-700 processVoxelBlob, :595-625 processVoxel, :710-763 traversal) --
__device__ __forceinline__ bool d_getX(float &x, float y, float z, const float *a, const float *b, const float *p0)
{
    // RFA:479-490
    const float x0 = p0[0], y0 = p0[1], z0 = p0[2];
    const float u = ((z - z0) * a[1] + (y0 - y) * a[2]) / (a[1] * b[2] - b[1] * a[2]);
    const float t = (-y0 + y - u * b[1]) / (a[1]);
    x = x0 + t * a[0] + u * b[0];
    return (t > 0.f) && (t < 1.f) && (u > 0.f) && (u < 1.f);
}

#define XH_TSZ 8
// Two-level culling: one block per 16^3 super-tile (2x2x2 tiles) lists, in launch order, the projections whose
// slab comes within reach of it; a tile / sub-cube then tests ~6 % of the launch instead of all of it. The
// super-tile test is the finer test with the half extent of the larger cube: the centres of a super-tile's tiles
// lie within 4 per axis (4*sqrt(3) = 6.93) of its centre, those of its sub-cubes within 6 per axis (10.4), so with the
// half extent 7.5 (sphere 13.3) it keeps every projection a tile (3.5 / 6.1) or a sub-cube (1.5 / 2.65) would keep.
#define XH_SUPERSHIFT 1                        // tiles per super-tile edge = 1 << XH_SUPERSHIFT
#define XH_SUPER (XH_TSZ << XH_SUPERSHIFT)     // voxels per super-tile edge
#define XH_SUPERH 7.5f                         // half extent of the cube of voxel centres
#define XH_SUPERRHO 13.3f                      // its half diagonal (12.99) + slack
__global__ void __launch_bounds__(256)
k_rf_supercull(const float4 *__restrict__ cullN, const float4 *__restrict__ cullX, int nspaces, int mv, float fr,
               int superDim, int superCap, int *__restrict__ superList, int *__restrict__ superCount,
               float4 *__restrict__ superN, float4 *__restrict__ superX)
{
    __shared__ int sCnt[4];
    __shared__ int sBase;
    const int sup = blockIdx.x;
    const int sx = sup % superDim, sy = (sup / superDim) % superDim, sz = sup / (superDim * superDim);
    const float cx = sx * XH_SUPER + XH_SUPERH - mv / 2, cy = sy * XH_SUPER + XH_SUPERH - mv / 2, cz = sz * XH_SUPER + XH_SUPERH - mv / 2;
    const float rho = XH_SUPERRHO, sizeX = (float)(mv / 2);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) sBase = 0;
    __syncthreads();
    // the next round's vectors travel while this round is counted and written
    float4 nNext = make_float4(0.f, 0.f, 0.f, 0.f), xNext = nNext;
    if ((int)threadIdx.x < nspaces) { nNext = cullN[threadIdx.x]; xNext = cullX[threadIdx.x]; }
    for (int s0 = 0; s0 < nspaces; s0 += 256) {
        const int s = s0 + threadIdx.x;
        bool hit = false;
        float4 n = nNext, r0 = xNext;
        if (s + 256 < nspaces) { nNext = cullN[s + 256]; xNext = cullX[s + 256]; }
        if (s < nspaces) {
            const float dn = n.x * cx + n.y * cy + n.z * cz;
            const float dx = r0.x * cx + r0.y * cy + r0.z * cz;
            // box bounds (voxel centres within +-XH_SUPERH of the super-tile centre per axis), never wider than the sphere bound
            const float hn = fminf(rho, XH_SUPERH * n.w + 0.05f), hx = fminf(rho, XH_SUPERH * r0.w + 0.05f);
            hit = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) sCnt[wv] = __popcll(bal);
        __syncthreads();
        int base = sBase, total = 0;
        for (int w = 0; w < 4; ++w) { const int c = sCnt[w]; if (w < wv) base += c; total += c; }
        if (hit) {
            const size_t o = (size_t)sup * superCap + base + __popcll(bal & ((1ull << lane) - 1ull));
            superList[o] = s;
            if (superN) { superN[o] = n; superX[o] = r0; }    // inline copies: the sub-cube cull streams them, no gather
        }
        __syncthreads();
        if (threadIdx.x == 0) sBase += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) superCount[sup] = sBase;
}

// hit part of getX (RFA:479-490; the x it would return is not needed): same operations in the same order as d_getX
__device__ __forceinline__ bool d_hit(float y, float z, float a1, float a2, float b1, float b2, float y0, float z0)
{
    const float u = ((z - z0) * a1 + (y0 - y) * a2) / (a1 * b2 - b1 * a2);
    // t = tn / a1 is only compared with 0 and 1. For IEEE division, round to nearest: fl(tn/a1) > 0 <=> tn and a1 have the
    // same sign (unless the quotient underflows: left to the division), and fl(tn/a1) < 1 <=> |tn| < |a1| for a positive
    // quotient (two floats with |tn| < |a1| have a quotient <= 1 - 2^-24, which never rounds up to 1; with |tn| > |a1| it
    // is >= 1 + 2^-23 (1 - 2^-24)^-1 ... > 1 + 2^-24 and never rounds down to 1). NaN / zero operands: both forms say no.
    const float tn = -y0 + y - u * b1;
    bool tOk;
    if (fabsf(tn) > 1e-30f && fabsf(a1) < 1e6f)      // the quotient cannot underflow: at least 1e-36
        tOk = ((tn > 0.f) == (a1 > 0.f)) && (fabsf(tn) < fabsf(a1));
    else {
        const float t = tn / a1;
        tOk = (t > 0.f) && (t < 1.f);
    }
    return tOk && (u > 0.f) && (u < 1.f);
}

#include "xh_rf_grid.h"

// ---- finaliser ---------------------------------------------------------------------------
// mirrorAndCrop RFA:861-887 in gather form. in: (mv+1)^3, out: (mv+1)^2 (half+1)
__global__ void k_rf_mirror(const xh_cf *__restrict__ inV, const float *__restrict__ inW,
                            xh_cf *__restrict__ outV, float *__restrict__ outW, int mv)
{
    const int half = mv / 2, dim = mv + 1, nx = half + 1;
    const size_t total = (size_t)dim * dim * nx;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int xo = idx % nx;
    const int y = (idx / nx) % dim;
    const int z = idx / ((size_t)nx * dim);
    const size_t d = ((size_t)z * dim + y) * dim + (xo + half);
    xh_cf v = inV[d];
    float w = inW[d];
    if (xo >= 1) {
        const size_t m = ((size_t)(mv - z) * dim + (mv - y)) * dim + (half - xo);
        // reference order: the mirrored contribution (x < half) is added before or after the
        // direct one depending on traversal; float addition of two terms commutes
        xh_cf mvv = inV[m];
        v.x += mvv.x;
        v.y += -mvv.y;
        w += inW[m];
    }
    outV[idx] = v;
    outW[idx] = w;
}

// applyBlob (--fast) RFA:793-831 on the cropped spaces
template <typename T>
__global__ void k_rf_applyblob(const T *__restrict__ in, T *__restrict__ out, const float *__restrict__ blobTable,
                               int mv, float blobSize, float iDeltaSqrt, int ncomp)
{
    const int half = mv / 2, dim = mv + 1, nx = half + 1;
    const size_t total = (size_t)dim * dim * nx;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = idx % nx;
    const int j = (idx / nx) % dim;
    const int i = idx / ((size_t)nx * dim);
    const float blobSizeSqr = blobSize * blobSize;
    const int blob = (int)floorf(blobSize);
    for (int c = 0; c < ncomp; ++c) {
        float tmp = 0;
        for (int z = max(0, i - blob); z <= min(mv, i + blob); z++) {
            const float dZSqr = (i - z) * (i - z);
            for (int y = max(0, j - blob); y <= min(mv, j + blob); y++) {
                const float dYSqr = (j - y) * (j - y);
                for (int x = max(0, k - blob); x <= min(half, k + blob); x++) {
                    const float dXSqr = (k - x) * (k - x);
                    const float distanceSqr = dZSqr + dYSqr + dXSqr;
                    if (distanceSqr > blobSizeSqr) continue;
                    const int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                    tmp += blobTable[aux] * in[(((size_t)z * dim + y) * nx + x) * ncomp + c];
                }
            }
        }
        out[idx * ncomp + c] = tmp;
    }
}

// forceHermitianSymmetry RFA:889-906 (x = 0 plane); each unordered pair handled once, by the
// member the reference's loop meets first
__global__ void k_rf_hermitian(xh_cf *__restrict__ V, float *__restrict__ W, int mv)
{
    const int half = mv / 2, dim = mv + 1, nx = half + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= dim * (half + 1)) return;
    const int z = idx / (half + 1), y = idx - z * (half + 1);
    if (y == half && z > half) return;
    const size_t o = ((size_t)z * dim + y) * nx;
    const size_t n = ((size_t)(mv - z) * dim + (mv - y)) * nx;
    const xh_cf vn = V[n], vo = V[o];
    xh_cf t;
    t.x = 0.5f * (vn.x + vo.x);
    t.y = 0.5f * (vn.y + (-vo.y));
    const float tw = 0.5f * (W[n] + W[o]);
    V[n] = t;
    V[o] = xh_cf{t.x, -t.y};
    W[n] = tw;
    W[o] = tw;
}

// processWeights RFA:908-924
__global__ void k_rf_weights(xh_cf *__restrict__ V, const float *__restrict__ W, size_t total, float corr2D_3D)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float weight = W[idx];
    xh_cf v = V[idx];
    if ((double)weight > 0.001) {   // "weight > ACCURACY": ACCURACY is a double constant
        const float f = corr2D_3D / weight;
        v.x *= f;
        v.y *= f;
    } else v = xh_cf{0.f, 0.f};
    V[idx] = v;
}

// convertToExpectedSpace RFA:834-851 in gather form: out [P][P][P/2+1] complex<double>
__global__ void k_rf_expand(const xh_cf *__restrict__ V, xh_cd *__restrict__ out, int mv, int P)
{
    const int half = mv / 2, dim = mv + 1, nx = half + 1, xh = P / 2 + 1;
    const size_t total = (size_t)P * P * xh;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int n0 = idx % xh;
    const int n1 = (idx / xh) % P;
    const int n2 = idx / ((size_t)xh * P);
    xh_cd acc = xh_cd{0., 0.};
    if (n0 <= half) {
        // pre-images of n1: y = n1 + half (y >= half) and y = n1 - (P - half) (y < half)
        int ys[2], zs[2], ny = 0, nz = 0;
        if (n1 + half <= mv) ys[ny++] = n1 + half;
        if (n1 - (P - half) >= 0 && n1 - (P - half) < half) ys[ny++] = n1 - (P - half);
        if (n2 + half <= mv) zs[nz++] = n2 + half;
        if (n2 - (P - half) >= 0 && n2 - (P - half) < half) zs[nz++] = n2 - (P - half);
        // reference accumulation order: z ascending, then y ascending
        if (nz == 2 && zs[0] > zs[1]) { int t = zs[0]; zs[0] = zs[1]; zs[1] = t; }
        if (ny == 2 && ys[0] > ys[1]) { int t = ys[0]; ys[0] = ys[1]; ys[1] = t; }
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b) {
                const xh_cf v = V[((size_t)zs[a] * dim + ys[b]) * nx + n0];
                acc.x += (double)v.x;
                acc.y += (double)v.y;
            }
    }
    out[idx] = acc;
}

// last pass of the 3-D c2r inverse + CenterFFT(.,false) + window to D^3 + blob/sinc correction
// (RFA:1028-1052). One line = (z,y) of the padded volume; only lines inside the window run.
__global__ void __launch_bounds__(256)
k_rf_c2r_window(const xh_cd *__restrict__ F, XhPlan<double> plan, double *__restrict__ vol,
                const double *__restrict__ fourierBlob, int D, double iDeltaFourier,
                double ipad_relation, double meanFactor2, int lpb)
{
    extern __shared__ __align__(16) unsigned char smem[];
    xh_cd *s = reinterpret_cast<xh_cd *>(smem);
    const int P = plan.n, M = 1 << plan.logM, xh = P / 2 + 1;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int line0 = blockIdx.x * lpb;
    const int totalLines = D * D;
    const int nl = min(lpb, totalLines - line0);
    const int s0 = -(D / 2);
    for (int i = tid; i < lpb * P; i += nth) {
        const int l = i / P, e = i - l * P;
        xh_cd v = xh_cd{0., 0.};
        if (l < nl) {
            const int ln = line0 + l;
            const int k = ln / D, ii = ln - k * D;          // output (z,y) index in the window
            const int rk = (k + s0 + P) % P, ri = (ii + s0 + P) % P;      // raw FFT indices
            const xh_cd *row = F + ((size_t)rk * P + ri) * xh;
            // Hermitian extension of the half row; c2r ignores Im of DC and Nyquist
            if (e < xh) { v = row[e]; if (e == 0 || 2 * e == P) v.y = 0; }
            else { v = row[P - e]; v.y = -v.y; }
        }
        s[l * M + xh_plan_pos(plan, e)] = v;
    }
    __syncthreads();
    xh_plan_exec<double, true>(s, plan, lpb, tid, nth);
    for (int i = tid; i < nl * D; i += nth) {
        const int l = i / D, j = i - l * D;
        const int ln = line0 + l;
        const int k = ln / D, ii = ln - k * D;
        const int lk = k + s0, li = ii + s0, lj = j + s0;
        const int rj = (lj + P) % P;
        double val = s[l * M + rj].x;
        const double radius = sqrt((double)(lk * lk + li * li + lj * lj));
        const double aux = radius * iDeltaFourier;
        const double factor = fourierBlob[(int)floor(aux + 0.5)];
        const double xs = radius / (2 * D);
        const double sinc = (xs == 0) ? 1.0 : sin(3.14159265358979323846 * xs) / (3.14159265358979323846 * xs);
        const double factor2 = sinc * sinc;
        // meanFactor2 < 0: xmipp_reconstruct_fourier --iter 0 divides by the blob transform only (RF:1160-1166)
        if (meanFactor2 < 0) val /= (ipad_relation * factor);
        else val = val / (ipad_relation * factor2 * factor) * meanFactor2;
        vol[((size_t)k * D + ii) * D + j] = val;
    }
}

// =========================================================================== host API
static void drain_events(xh_rf *rf)
{
    for (size_t e = 0; e + 1 < rf->evUsed; e += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(rf->evPool[e + 1]) == hipSuccess && hipEventElapsedTime(&ms, rf->evPool[e], rf->evPool[e + 1]) == hipSuccess) {
            rf->kernelMs += ms;
            rf->kernelLaunches++;
        }
    }
    rf->evUsed = 0;
}
static hipEvent_t next_event(xh_rf *rf)
{
    if (rf->evUsed == rf->evPool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        rf->evPool.push_back(e);
    }
    return rf->evPool[rf->evUsed++];
}

static int make_twiddles(xh_ctx *ctx, int n, XhBuf &b32, XhBuf &b64)
{
    // fp32 table: all n entries (the register-blocked kernels index j < n, the radix-2 ones j < n/2)
    std::vector<xh_cf> w32(n);
    std::vector<xh_cd> w64(n / 2);
    for (int j = 0; j < n; ++j) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * j / n;
        if (j < n / 2) w64[j] = xh_cd{(double)cosl(a), (double)sinl(a)};
        w32[j] = xh_cf{(float)cosl(a), (float)sinl(a)};
    }
    XH_TRY(xh_buf_alloc(ctx, b32, sizeof(xh_cf) * n));
    XH_TRY(xh_buf_alloc(ctx, b64, sizeof(xh_cd) * (n / 2)));
    XH_HIP(hipMemcpy(b32.p, w32.data(), b32.bytes, hipMemcpyHostToDevice));
    XH_HIP(hipMemcpy(b64.p, w64.data(), b64.bytes, hipMemcpyHostToDevice));
    return XH_OK;
}

__global__ void __launch_bounds__(256) k_rf_scale_cd(xh_cd *__restrict__ v, size_t n, double f)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) { v[t].x *= f; v[t].y *= f; }
}
__global__ void __launch_bounds__(256) k_rf_add_d(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) dst[t] += src[t];
}

// Tail of finishComputations (RFA:1017-1054 / RF:1127-1178) shared by the two programs: the expanded spectrum (P x P x (P/2+1) complex
// doubles, FFTW layout) -> inverse transform, CenterFFT, window to D^3, blob (and sinc^2) correction -> host volume.
static int finish_from_spectrum(xh_ctx *ctx, const XhPlan<double> &plan, xh_cd *specp, double *volp, const double *fbtp, int D, double iDeltaFourier,
                                double padding_proj, double padding_vol, double &meanFactor2Cache, bool sincCorrection, double *h_volume)
{
    const int P = plan.n, xh = P / 2 + 1;
    struct { void *p; } spec{specp}, vol{volp}, fbt{(void *)fbtp};
    const size_t volBytes = sizeof(double) * (size_t)D * D * D;
#define XH_HIP_C(call) XH_HIP(call)
    const int lpb = xh_plan_lpb(plan, 64 * 1024, 8);
    const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << plan.logM;
    // inverse along z: lines (y,x), element stride P*xh
    {
        const size_t nlines = (size_t)P * xh;
        hipLaunchKernelGGL((xh_k_fft_lines<double, true>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream,
                           (xh_cd *)spec.p, plan, nlines, nlines, (size_t)0, (size_t)1, (size_t)P * xh, lpb);
        XH_HIP_C(hipGetLastError());
    }
    // inverse along y: lines (z,x): offset z*P*xh + x, element stride xh
    {
        const size_t nlines = (size_t)P * xh;
        hipLaunchKernelGGL((xh_k_fft_lines<double, true>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream,
                           (xh_cd *)spec.p, plan, nlines, (size_t)xh, (size_t)P * xh, (size_t)1, (size_t)xh, lpb);
        XH_HIP_C(hipGetLastError());
    }
    // meanFactor2 = mean over the D^3 window of sinc^2(radius/(2D)) (RFA:1040-1050). It depends on D only:
    // computed once, grouped by the integer squared radius (the grouping changes the sum by ~1e-15 relative).
    if (meanFactor2Cache < 0) {
        const int s0 = -(D / 2), s1 = s0 + D - 1;
        const int maxr2 = 3 * std::max(s0 * s0, s1 * s1);
        std::vector<long long> cnt((size_t)maxr2 + 1, 0);
        std::vector<int> c1(D);
        for (int i = 0; i < D; ++i) c1[i] = (i + s0) * (i + s0);
        for (int k = 0; k < D; ++k)
            for (int i = 0; i < D; ++i) {
                const int b = c1[k] + c1[i];
                for (int j = 0; j < D; ++j) ++cnt[b + c1[j]];
            }
        double acc = 0;
        for (int r2 = 0; r2 <= maxr2; ++r2)
            if (cnt[r2]) {
                const double radius = std::sqrt((double)r2);
                const double xs = radius / (2 * D);
                const double sinc = (xs == 0) ? 1.0 : std::sin(kPI * xs) / (kPI * xs);
                acc += (double)cnt[r2] * std::pow(sinc, 2);
            }
        meanFactor2Cache = acc / ((double)D * D * D);
    }
    const double meanFactor2 = sincCorrection ? meanFactor2Cache : -1.0;
    const double pr0 = padding_proj / padding_vol;
    const double ipad_relation = 1.0 / (pr0 * pr0 * pr0);
    hipLaunchKernelGGL(k_rf_c2r_window, dim3((unsigned)(((size_t)D * D + lpb - 1) / lpb)), dim3(256), smem, ctx->stream,
                       (const xh_cd *)spec.p, plan, (double *)vol.p, (const double *)fbt.p, D,
                       iDeltaFourier, ipad_relation, meanFactor2, lpb);
    XH_HIP_C(hipGetLastError());
    XH_HIP_C(hipMemcpyAsync(h_volume, vol.p, volBytes, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP_C(hipStreamSynchronize(ctx->stream));
#undef XH_HIP_C
    return XH_OK;
}

extern "C" {

int xh_rf_create(xh_ctx *ctx, const xh_rf_params *p, xh_rf **out)
{
    XH_CHECK(ctx && p && out, XH_ERR_ARG, "xh_rf_create: null argument");
    XH_CHECK(p->imgSize >= 4, XH_ERR_ARG, "xh_rf_create: imgSize %d too small", p->imgSize);
    XH_CHECK(p->blob_order == 0 || p->blob_order == 2, XH_ERR_ARG,
             "xh_rf_create: blob order %d unsupported (kaiser_Fourier_value handles 0 and 2 only, blobs.cpp:146-147)",
             p->blob_order);
    XH_HIP(hipSetDevice(ctx->device));
    xh_rf *rf = new xh_rf;
    rf->ctx = ctx;
    rf->p = *p;
    rf->D = p->imgSize;
    // RFA:196-199
    rf->P = (int)(rf->D * p->padding_vol);
    size_t conserveRows = (size_t)std::ceil((double)rf->P * p->max_resolution * 2.0);
    conserveRows = (size_t)std::ceil((double)conserveRows / 2.0);
    rf->mv = 2 * (int)conserveRows;
    rf->sizeX = rf->mv / 2;
    rf->sizeY = rf->mv;
    if (rf->P > 2048 || rf->mv > rf->P) {
        xh_set_error("xh_rf_create: padded size %d (imgSize %d x padding %g) must be <= 2048 "
                     "and max_resolution <= 0.5 for the device FFT", rf->P, rf->D, p->padding_vol);
        delete rf;
        return XH_ERR_UNSUPPORTED;
    }
    // tables, RFA:201-239
    rf->blobTableSqrt.resize(XH_BLOB_TABLE);
    rf->fourierBlobTable.resize(XH_BLOB_TABLE);
    const int Xdim = rf->D;
    const double rFourier = p->blob_radius / (p->padding_vol * Xdim);
    const double rNorm = p->blob_radius / (p->padding_proj / p->padding_vol);
    const double deltaSqrt = (p->blob_radius * p->blob_radius) / (XH_BLOB_TABLE - 1);
    const double deltaFourier = (std::sqrt(3.) * Xdim / 2.) / (XH_BLOB_TABLE - 1);
    const double iw0 = 1.0 / h_kaiser_fourier(0.0, rNorm, p->blob_alpha, p->blob_order);
    double padXdim3 = p->padding_vol * Xdim;
    padXdim3 = padXdim3 * padXdim3 * padXdim3;
    const double blobTableSize = p->blob_radius * std::sqrt(1. / (XH_BLOB_TABLE - 1));
    for (int i = 0; i < XH_BLOB_TABLE; i++) {
        rf->blobTableSqrt[i] = h_kaiser_value(blobTableSize * std::sqrt((double)i), p->blob_radius, p->blob_alpha, p->blob_order) * iw0;
        rf->fourierBlobTable[i] = h_kaiser_fourier(deltaFourier * i, rFourier, p->blob_alpha, p->blob_order) * padXdim3 * iw0;
    }
    rf->iDeltaSqrt = 1 / deltaSqrt;
    rf->iDeltaFourier = 1 / deltaFourier;
    rf->d_temp = nullptr;
    rf->cropped = false;
    rf->tile_max_spaces = 8192;
    rf->fft_variant = 0;
    rf->evUsed = 0;
    rf->kernelMs = 0;
    rf->kernelLaunches = 0;
    rf->meanFactor2 = -1;
    int r = xh_buf_alloc(ctx, rf->d_blob, sizeof(float) * XH_BLOB_TABLE);
    if (r == XH_OK) r = (hipMemcpy(rf->d_blob.p, rf->blobTableSqrt.data(), rf->d_blob.bytes, hipMemcpyHostToDevice) == hipSuccess) ? XH_OK : XH_ERR_HIP;
    if (r == XH_OK) r = make_twiddles(ctx, rf->P, rf->d_twP32, rf->d_twP64);
    if (r == XH_OK) r = xh_plan_create<float>(ctx, rf->P, rf->planP32);
    if (r == XH_OK) r = xh_plan_create<double>(ctx, rf->P, rf->planP64);
    if (r == XH_OK) r = xh_buf_alloc(ctx, rf->d_tileCounter, sizeof(int) * 512);
    // k_rf_grid: tiles of 2 x 2 x 2 units (16 x 16 x 8 voxels for units 4 deep, 16^3 for units 8 deep) that a projection can
    // reach (sphere of radius sizeX + blob), in raster order (z, y, x) cut into 8 contiguous z-slabs of equal estimated work, one
    // per XCD (block b runs on XCD b % 8: a projection's patch is pulled into one or two L2s instead of all eight). A tile at
    // distance rho from the centre is crossed by a fraction ~1/rho of all central planes: that is its weight. Inside a class
    // Morton order (the waves of the chip work on a narrow band of consecutive tiles, and a compact band shares more of the
    // projections' patches in the L2 than a row of the raster). Class offsets at d_tileCounter + 32 (units 4 deep) and + 48 (8).
    for (int v = 0; v < 2 && r == XH_OK; ++v) {
        const int tzs = v ? 16 : 8;                            // voxels per tile in z
        const int tpx = (rf->mv + 1 + 15) / 16, tpz = (rf->mv + 1 + tzs - 1) / tzs;
        const double hz = 0.5 * tzs - 0.5;
        const double R = rf->sizeX + p->blob_radius + std::sqrt(2 * 7.5 * 7.5 + hz * hz) + 1.0;
        std::vector<unsigned> packed;
        std::vector<double> wsum;
        double acc = 0;
        for (int tz = 0; tz < tpz; ++tz)
            for (int ty = 0; ty < tpx; ++ty)
                for (int tx = 0; tx < tpx; ++tx) {
                    const double cx = tx * 16 + 7.5 - rf->mv / 2, cy = ty * 16 + 7.5 - rf->mv / 2, cz = tz * tzs + hz - rf->mv / 2;
                    const double d = std::sqrt(cx * cx + cy * cy + cz * cz);
                    if (d <= R) { packed.push_back((unsigned)(tx | (ty << 10) | (tz << 20))); acc += 1.0 / std::max(d, 8.0); wsum.push_back(acc); }
                }
        int classOff[9];
        classOff[0] = 0;
        for (int c = 1; c < 8; ++c)
            classOff[c] = (int)(std::lower_bound(wsum.begin(), wsum.end(), acc * c / 8.0) - wsum.begin());
        classOff[8] = (int)packed.size();
        auto spread = [](unsigned v) { unsigned long long x = v & 0x3ff; x = (x | x << 16) & 0x30000ffULL; x = (x | x << 8) & 0x300f00fULL; x = (x | x << 4) & 0x30c30c3ULL; x = (x | x << 2) & 0x9249249ULL; return x; };
        // Morton order; tiles of 4-deep units are half as tall as wide: on (x, y, z / 2) with the low bit of z last
        auto key = [&](unsigned t) {
            if (tzs == 16) return spread(t & 0x3ff) | spread((t >> 10) & 0x3ff) << 1 | spread((t >> 20) & 0x3ff) << 2;
            return (spread(t & 0x3ff) | spread((t >> 10) & 0x3ff) << 1 | spread((t >> 21) & 0x1ff) << 2) << 1 | ((t >> 20) & 1);
        };
        for (int c = 0; c < 8; ++c)
            std::sort(packed.begin() + classOff[c], packed.begin() + classOff[c + 1], [&](unsigned u, unsigned w) { return key(u) < key(w); });
        rf->ntiles[v] = (int)packed.size();
        r = xh_buf_alloc(ctx, rf->d_gtiles[v], sizeof(unsigned) * std::max<size_t>(1, packed.size()));
        if (r == XH_OK) r = (hipMemcpy(rf->d_gtiles[v].p, packed.data(), sizeof(unsigned) * packed.size(), hipMemcpyHostToDevice) == hipSuccess) ? XH_OK : XH_ERR_HIP;
        if (r == XH_OK) r = (hipMemcpy((int *)rf->d_tileCounter.p + 32 + 16 * v, classOff, sizeof(classOff), hipMemcpyHostToDevice) == hipSuccess) ? XH_OK : XH_ERR_HIP;
    }
    if (r != XH_OK) { xh_rf_destroy(rf); return r; }
    *out = rf;
    return XH_OK;
}

int xh_rf_destroy(xh_rf *rf)
{
    if (!rf) return XH_OK;
    (void)hipSetDevice(rf->ctx->device);
    (void)hipStreamSynchronize(rf->ctx->stream);
    xh_buf_free(rf->d_blob); xh_buf_free(rf->d_twP32); xh_buf_free(rf->d_twP64);
    xh_plan_free(rf->planP32); xh_plan_free(rf->planP64);
    xh_buf_free(rf->own_temp); xh_buf_free(rf->d_rows);
    xh_buf_free(rf->d_ctfp); xh_buf_free(rf->d_fin);
    xh_buf_free(rf->d_shiftCoef); xh_buf_free(rf->d_shiftXY);
    xh_buf_free(rf->d_tileCounter); xh_buf_free(rf->d_cull); xh_buf_free(rf->d_pack);
    xh_buf_free(rf->d_superList); xh_buf_free(rf->d_superCount); xh_buf_free(rf->d_superVec);
    xh_buf_free(rf->d_sym); xh_buf_free(rf->d_angles); xh_buf_free(rf->d_spacePos);
    xh_buf_free(rf->d_finSpec); xh_buf_free(rf->d_finVol); xh_buf_free(rf->d_finFbt);
    if (rf->h_stage) (void)hipHostFree(rf->h_stage);
    for (int h = 0; h < 2; ++h)
        if (rf->stageEv[h]) (void)hipEventDestroy(rf->stageEv[h]);
    xh_buf_free(rf->d_gtiles[0]); xh_buf_free(rf->d_gtiles[1]); xh_buf_free(rf->d_grecs); xh_buf_free(rf->d_gweights); xh_buf_free(rf->d_planes); xh_buf_free(rf->d_spectra);
    for (hipEvent_t e : rf->evPool) (void)hipEventDestroy(e);
    delete rf;
    return XH_OK;
}

int xh_rf_kernel_ms(xh_rf *rf, double *h_ms, int64_t *h_launches, int32_t reset)
{
    XH_CHECK(rf && h_ms, XH_ERR_ARG, "xh_rf_kernel_ms: bad argument");
    drain_events(rf);
    *h_ms = rf->kernelMs;
    if (h_launches) *h_launches = rf->kernelLaunches;
    if (reset) { rf->kernelMs = 0; rf->kernelLaunches = 0; }
    return XH_OK;
}

int xh_rf_set_option(xh_rf *rf, const char *name, double value)
{
    XH_CHECK(rf && name, XH_ERR_ARG, "null argument");
    if (!strcmp(name, "unit_z")) {
        XH_CHECK((int)value == 4 || (int)value == 8, XH_ERR_ARG, "xh_rf_set_option: unit_z is 4 or 8");
        rf->unit_z = (int)value;
    }
    else if (!strcmp(name, "grid_waves")) rf->grid_waves = (int)value;
    else if (!strcmp(name, "grid_tile_budget")) {
#ifdef XH_DEBUG_HOOKS
        rf->grid_tile_budget = std::max(-1, (int)value);     // (-1: experiment, every interior visit reuses the previous queue: timing only, WRONG volume)
#else
        XH_CHECK(value >= 0, XH_ERR_ARG, "xh_rf_set_option: grid_tile_budget -1 (timing experiment with a wrong volume) needs a library built with XH_DEBUG_HOOKS");
        rf->grid_tile_budget = (int)value;
#endif
    }
    else if (!strcmp(name, "fuse_ctf")) rf->fuse_ctf = (int)value;
    else if (!strcmp(name, "ctf_fast")) rf->ctf_fast = (int)value;
    else if (!strcmp(name, "order_spaces")) rf->order_spaces = (int)value;
    else if (!strcmp(name, "skip_far_cells")) rf->skip_far_cells = (int)value;
    else if (!strcmp(name, "shift_bands")) rf->shift_bands = (int)value;
    else if (!strcmp(name, "records_from_images")) rf->records_from_images = (int)value;
    else if (!strcmp(name, "tile_max_spaces")) rf->tile_max_spaces = (int)value;
    else if (!strcmp(name, "fft_variant")) rf->fft_variant = (int)value;
    else { xh_set_error("xh_rf_set_option: unknown option %s", name); return XH_ERR_ARG; }
    return XH_OK;
}

int xh_rf_sizes(const xh_rf *rf, int32_t *P, int32_t *mv, int32_t *sx, int32_t *sy)
{
    XH_CHECK(rf, XH_ERR_ARG, "null handle");
    if (P) *P = rf->P;
    if (mv) *mv = rf->mv;
    if (sx) *sx = rf->sizeX;
    if (sy) *sy = rf->sizeY;
    return XH_OK;
}

int xh_rf_tables(const xh_rf *rf, float *bt, double *fbt, float *ids, float *idf)
{
    XH_CHECK(rf, XH_ERR_ARG, "null handle");
    if (bt) memcpy(bt, rf->blobTableSqrt.data(), sizeof(float) * XH_BLOB_TABLE);
    if (fbt) memcpy(fbt, rf->fourierBlobTable.data(), sizeof(double) * XH_BLOB_TABLE);
    if (ids) *ids = rf->iDeltaSqrt;
    if (idf) *idf = rf->iDeltaFourier;
    return XH_OK;
}

size_t xh_rf_temp_floats(const xh_rf *rf)
{
    const size_t d = rf->mv + 1;
    return 3 * d * d * d;
}
size_t xh_rf_cropped_floats(const xh_rf *rf)
{
    const size_t d = rf->mv + 1;
    return 3 * d * d * (size_t)(rf->mv / 2 + 1);
}

int xh_rf_attach_temp(xh_rf *rf, float *d_temp)
{
    XH_CHECK(rf && d_temp, XH_ERR_ARG, "null argument");
    rf->d_temp = d_temp;
    rf->cropped = false;
    return XH_OK;
}

static int ensure_temp(xh_rf *rf)
{
    if (rf->d_temp) return XH_OK;
    XH_TRY(xh_buf_alloc(rf->ctx, rf->own_temp, sizeof(float) * xh_rf_temp_floats(rf)));
    rf->d_temp = (float *)rf->own_temp.p;
    XH_HIP(hipMemsetAsync(rf->d_temp, 0, rf->own_temp.bytes, rf->ctx->stream));
    return XH_OK;
}

int xh_rf_temp_ptr(xh_rf *rf, float **d_temp)
{
    XH_CHECK(rf && d_temp, XH_ERR_ARG, "null argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_TRY(ensure_temp(rf));
    *d_temp = rf->d_temp;
    return XH_OK;
}

int xh_rf_reset(xh_rf *rf)
{
    XH_CHECK(rf, XH_ERR_ARG, "null handle");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_TRY(ensure_temp(rf));
    XH_HIP(hipMemsetAsync(rf->d_temp, 0, sizeof(float) * xh_rf_temp_floats(rf), rf->ctx->stream));
    rf->cropped = false;
    return XH_OK;
}

// can the projections take the columns-first / rows-last transform (k_rf_colsA, k_rf_rowsB)?
static bool fft_cols_rows_ok(const xh_rf *rf)
{
    return (rf->P == 512 || rf->P == 256 || rf->P == 128) && rf->sizeY == rf->P && rf->fft_variant == 0;
}
// n images -> half spectra (d_fft) or, with d_pk, the gridding kernel's padded records (CTF of rf->d_ctfp, weights or null)
static int fft_cols_rows(xh_rf *rf, const float *d_imgs, int n, xh_cf *d_fft, XgCell *d_pk, const float *d_weights)
{
    xh_ctx *ctx = rf->ctx;
    const int D = rf->D, P = rf->P, sizeX = rf->sizeX;
    // a line of T holds the D columns in blocks of NC = 4 LN, LN lines per workgroup of the FFT (xh_fftreg.h): 8 at P = 512, else 16
    const int NC = P == 512 ? 32 : 64, TD = (D + NC - 1) / NC * NC;
    const size_t perImg = (size_t)(sizeX + 1) * TD * sizeof(xh_cf);
    const int chunk = std::max(1, std::min(n, (int)((256u << 20) / perImg)));
    XH_TRY(xh_buf_reserve(ctx, rf->d_rows, (size_t)chunk * perImg));
    const double maxResSqr = rf->p.max_resolution * rf->p.max_resolution;
    const int nlines = sizeX + 1 + (d_pk ? XG_PAD : 0);
    const size_t recCells = (size_t)(sizeX + 2 * XG_PAD) * (rf->sizeY + 2 * XG_PAD);
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
#define XH_RFB(A_, B_)                                                                                                              \
    {                                                                                                                               \
        typedef TrGeom<A_, B_, float> G;                                                                                            \
        hipLaunchKernelGGL((k_rf_colsA<A_, B_>), dim3(m * ((D + 4 * G::LN - 1) / (4 * G::LN))), dim3(256), G::smem, ctx->stream,      \
                           d_imgs + (size_t)i0 * D * D, (xh_cf *)rf->d_rows.p, (const xh_cf *)rf->d_twP32.p, D, TD, sizeX);          \
        if (d_pk)                                                                                                                   \
            hipLaunchKernelGGL((k_rf_rowsB<A_, B_, true>), dim3(m * ((nlines + G::LN - 1) / G::LN)), dim3(256), G::smem, ctx->stream, \
                               (const xh_cf *)rf->d_rows.p, (xh_cf *)nullptr, d_pk + (size_t)i0 * recCells,                          \
                               (const XhCtfDev *)rf->d_ctfp.p + i0, d_weights ? d_weights + i0 : nullptr,                           \
                               (const xh_cf *)rf->d_twP32.p, D, TD, sizeX, maxResSqr, nlines, 1.0 / rf->p.sampling, rf->p.min_ctf,    \
                               rf->p.phase_flipped, 0x7fffffff);                                                                    \
        else                                                                                                                        \
            hipLaunchKernelGGL((k_rf_rowsB<A_, B_, false>), dim3(m * ((nlines + G::LN - 1) / G::LN)), dim3(256), G::smem, ctx->stream, \
                               (const xh_cf *)rf->d_rows.p, d_fft + (size_t)i0 * rf->sizeY * sizeX, (XgCell *)nullptr,               \
                               (const XhCtfDev *)nullptr, (const float *)nullptr, (const xh_cf *)rf->d_twP32.p, D, TD, sizeX, maxResSqr, \
                               nlines, 0.0, 0.0, 0, rf->fftSkipR2);                                                                 \
    }
        if (P == 512) XH_RFB(16, 32)
        else if (P == 256) XH_RFB(16, 16)
        else XH_RFB(16, 8)
#undef XH_RFB
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

int xh_rf_prepare_images(xh_rf *rf, const float *d_imgs, int32_t n, float *d_fft)
{
    XH_CHECK(rf && d_imgs && d_fft && n >= 0, XH_ERR_ARG, "xh_rf_prepare_images: bad argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    if (n == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    if (fft_cols_rows_ok(rf)) return fft_cols_rows(rf, d_imgs, n, (xh_cf *)d_fft, nullptr, nullptr);
    const int D = rf->D, P = rf->P, sizeX = rf->sizeX;
    // chunk so that the row-pass intermediate stays modest
    const int chunk = std::max(1, std::min(n, (int)((256u << 20) / ((size_t)D * sizeX * sizeof(xh_cf)))));
    XH_TRY(xh_buf_reserve(ctx, rf->d_rows, (size_t)chunk * D * sizeX * sizeof(xh_cf)));
    const XhPlan<float> &plan = rf->planP32.plan;
    const int lpb = xh_plan_lpb(plan, 64 * 1024, 16);
    const size_t smem = ((size_t)lpb * sizeof(xh_cf)) << plan.logM;
    const double maxResSqr = rf->p.max_resolution * rf->p.max_resolution;
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        const int totalLines = m * D;
        if ((P == 512 || P == 256 || P == 128) && rf->sizeY == P && rf->fft_variant == 2) {     // the row-first form, kept for A/B
#define XH_RF2(A_, B_)                                                                                                          \
    {                                                                                                                           \
        typedef TrGeom<A_, B_, float> G;                                                                                        \
        hipLaunchKernelGGL((k_rf_rows2<A_, B_>), dim3((totalLines + G::LN - 1) / G::LN), dim3(256), G::smem, ctx->stream,         \
                           d_imgs + (size_t)i0 * D * D, (xh_cf *)rf->d_rows.p, (const xh_cf *)rf->d_twP32.p, D, sizeX, totalLines); \
        hipLaunchKernelGGL((k_rf_cols2<A_, B_>), dim3(m * ((sizeX + G::LN - 1) / G::LN)), dim3(256), G::smem, ctx->stream,        \
                           (const xh_cf *)rf->d_rows.p, (xh_cf *)d_fft + (size_t)i0 * rf->sizeY * sizeX,                        \
                           (const xh_cf *)rf->d_twP32.p, D, sizeX, maxResSqr);                                                \
    }
            if (P == 512) XH_RF2(16, 32)
            else if (P == 256) XH_RF2(16, 16)
            else XH_RF2(16, 8)
#undef XH_RF2
            XH_LAUNCH_CHECK();
            continue;
        }
        hipLaunchKernelGGL(k_rf_rows, dim3((totalLines + lpb - 1) / lpb), dim3(256), smem, ctx->stream,
                           d_imgs + (size_t)i0 * D * D, (xh_cf *)rf->d_rows.p, plan, D, sizeX, totalLines, lpb);
        XH_LAUNCH_CHECK();
        const int groups = (sizeX + lpb - 1) / lpb;
        hipLaunchKernelGGL(k_rf_cols, dim3(m * groups), dim3(256), smem, ctx->stream, (const xh_cf *)rf->d_rows.p,
                           (xh_cf *)d_fft + (size_t)i0 * rf->sizeY * sizeX, plan, D, sizeX, maxResSqr, lpb);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}


// shifts as [n] float2 and flips as [n] bytes (or null) already on the device
static int shift_images_run(xh_rf *rf, const float *d_imgs, const float *d_coefs, const float2 *d_shiftXY, const unsigned char *d_flip, int n, float *d_out);

int xh_rf_shift_images_coefs(xh_rf *rf, const float *d_imgs, const float *d_coefs, const float *h_shiftXY, const uint8_t *h_flip, int32_t n, float *d_out)
{
    XH_CHECK(rf && d_imgs && h_shiftXY && d_out && n >= 0, XH_ERR_ARG, "xh_rf_shift_images: bad argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_CHECK(d_imgs != d_out, XH_ERR_ARG, "xh_rf_shift_images: in-place operation is not supported");
    if (n == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    XH_TRY(xh_buf_reserve(ctx, rf->d_shiftXY, sizeof(float) * 2 * (size_t)n + (size_t)n));
    XH_TRY(stage_upload(rf, rf->d_shiftXY.p, h_shiftXY, sizeof(float) * 2 * (size_t)n));
    unsigned char *d_flip = nullptr;
    if (h_flip) {
        d_flip = (unsigned char *)rf->d_shiftXY.p + sizeof(float) * 2 * (size_t)n;
        XH_TRY(stage_upload(rf, d_flip, h_flip, (size_t)n));
    }
    return shift_images_run(rf, d_imgs, d_coefs, (const float2 *)rf->d_shiftXY.p, d_flip, n, d_out);
}

// the matcher's outputs as they lie on the device (xh_pm_translate: double shifts; xh_pm_match: flips)
__global__ void k_rf_shift_args(const double *__restrict__ sx, const double *__restrict__ sy, float2 *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2((float)sx[i], (float)sy[i]);
}

int xh_rf_shift_images_dev(xh_rf *rf, const float *d_imgs, const float *d_coefs, const double *d_shiftX, const double *d_shiftY,
                           const uint8_t *d_flip, int32_t n, float *d_out)
{
    XH_CHECK(rf && d_imgs && d_shiftX && d_shiftY && d_out && n >= 0, XH_ERR_ARG, "xh_rf_shift_images_dev: bad argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_CHECK(d_imgs != d_out, XH_ERR_ARG, "xh_rf_shift_images_dev: in-place operation is not supported");
    if (n == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    XH_TRY(xh_buf_reserve(ctx, rf->d_shiftXY, sizeof(float) * 2 * (size_t)n + (size_t)n));
    hipLaunchKernelGGL(k_rf_shift_args, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_shiftX, d_shiftY, (float2 *)rf->d_shiftXY.p, n);
    XH_LAUNCH_CHECK();
    return shift_images_run(rf, d_imgs, d_coefs, (const float2 *)rf->d_shiftXY.p, d_flip, n, d_out);
}

static int shift_images_run(xh_rf *rf, const float *d_imgs, const float *d_coefs, const float2 *d_shiftXY, const unsigned char *d_flip, int n, float *d_out)
{
    xh_ctx *ctx = rf->ctx;
    const int D = rf->D;
    if (!d_coefs) XH_TRY(xh_buf_reserve(ctx, rf->d_shiftCoef, sizeof(float) * (size_t)n * D * D));
    if (d_coefs) {
        // the caller's coefficients
    } else if (D >= 2 * XH_FIR_K) {
        // fp32 prefilter in its convolution form (xh_bspline.h); d_out is free until the shift kernel writes it
        xh_prefilter_fir_launch(ctx->stream, d_imgs, (float *)rf->d_shiftCoef.p, D, (size_t)n);
        XH_LAUNCH_CHECK();
    } else {
    const int TR = std::max(1, std::min(32, (int)(60000 / ((D + 1) * sizeof(float)))));
    const int tiles = (D + TR - 1) / TR;
    hipLaunchKernelGGL((k_pm_prefilter_rows<float, float>), dim3(n * tiles), dim3(64), sizeof(float) * TR * (D + 1), ctx->stream,
                       d_imgs, (const int *)nullptr, (float *)rf->d_shiftCoef.p, D, TR, (const int *)nullptr);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_pm_prefilter_cols<float>), dim3((n * D + 63) / 64), dim3(64), 0, ctx->stream, (float *)rf->d_shiftCoef.p, D, n,
                       (const int *)nullptr);
    XH_LAUNCH_CHECK();
    }
    const float *coefs = d_coefs ? d_coefs : (const float *)rf->d_shiftCoef.p;
    for (int i0 = 0; i0 < n; i0 += 65535) {          // blockIdx.y: image
        const int m = std::min(65535, n - i0);
        const size_t o = (size_t)i0 * D * D;
        if (D == 256 && rf->shift_bands)
            hipLaunchKernelGGL(k_rf_shift_band, dim3(D / XH_SHB, m), dim3(256), 0, ctx->stream, coefs + o, d_imgs + o, d_shiftXY + i0, d_flip ? d_flip + i0 : nullptr,
                               d_out + o, D);
        else
        hipLaunchKernelGGL(k_rf_shift, dim3((D * ((D + XH_SHIFT_V - 1) / XH_SHIFT_V) + 255) / 256, m), dim3(256), 0, ctx->stream, coefs + o, d_imgs + o,
                           d_shiftXY + i0, d_flip ? d_flip + i0 : nullptr, d_out + o, D);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

int xh_rf_shift_images(xh_rf *rf, const float *d_imgs, const float *h_shiftXY, const uint8_t *h_flip, int32_t n, float *d_out)
{
    return xh_rf_shift_images_coefs(rf, d_imgs, nullptr, h_shiftXY, h_flip, n, d_out);
}

// produceSideInfo of n CTF descriptions -> rf->d_ctfp (device)
static int ctf_params_upload(xh_rf *rf, const xh_ctf_params *h_ctf, int n)
{
    xh_ctx *ctx = rf->ctx;
    std::vector<XhCtfDev> hc(n);
    for (int i = 0; i < n; ++i) {
        const xh_ctf_params &c = h_ctf[i];
        // produceSideInfo, data/ctf.cpp:645-679,1392-1402
        const double local_Cs = c.Cs * 1e7, local_Ca = c.Ca * 1e7, local_kV = c.kV * 1e3, local_ispr = c.ispr * 1e6;
        const double lambda = 12.2643247 / std::sqrt(local_kV * (1. + 0.978466e-6 * local_kV));
        XhCtfDev &d = hc[i];
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
        d.phase_shift = c.phase_shift; d.VPP_radius = c.VPP_radius;
        // d_ctf_pixel_fast: no phase plate, no energy spread / focal spread / convergence cone / DeltaR term, no envR1 (which needs u)
        d.amp = std::hypot(d.Ksin, d.Kcos);
        d.phi = std::atan2(d.Kcos, d.Ksin);
        d.E0 = 57568490574.0 / 57568490411.0 + c.envR0;      // the J0(0) quotient of d_ctf_pixel
        d.c2az = std::cos(2 * d.rad_azimuth); d.s2az = std::sin(2 * d.rad_azimuth);
        d.fast = rf->ctf_fast && std::round(c.VPP_radius * 1000) == 0 && d.K3 == 0 && d.K5 == 0 && d.K6 == 0 && c.DeltaR == 0 && c.envR1 == 0
                 && std::isfinite(d.amp) && std::isfinite(d.E0) && rf->p.min_ctf > 0;
    }
    XH_TRY(xh_buf_reserve(ctx, rf->d_ctfp, sizeof(XhCtfDev) * n));
    XH_TRY(stage_upload(rf, rf->d_ctfp.p, hc.data(), sizeof(XhCtfDev) * n));
    return XH_OK;
}

int xh_rf_ctf_arrays(xh_rf *rf, const xh_ctf_params *h_ctf, int32_t n, float *d_ctf, float *d_mod)
{
    XH_CHECK(rf && h_ctf && d_ctf && d_mod && n >= 0, XH_ERR_ARG, "xh_rf_ctf_arrays: bad argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    if (n == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    XH_TRY(ctf_params_upload(rf, h_ctf, n));
    const int dcRow = rf->P / 2;
    const unsigned perImg = (unsigned)rf->sizeX * std::max(rf->sizeY - dcRow, dcRow + 1);
    for (int i0 = 0; i0 < n; i0 += 65535) {            // blockIdx.y: image
        const int m = std::min(65535, n - i0);
        const size_t o = (size_t)i0 * rf->sizeX * rf->sizeY;
        hipLaunchKernelGGL(k_rf_ctf, dim3((perImg + 255) / 256, m), dim3(256), 0, ctx->stream,
                           (const XhCtfDev *)rf->d_ctfp.p + i0, d_ctf + o, d_mod + o, m, rf->sizeX, rf->sizeY, rf->P,
                           1.0 / rf->p.sampling, rf->p.min_ctf, rf->p.phase_flipped);
    }
    XH_LAUNCH_CHECK();
    return XH_OK;
}

// ---- the product path: k_rf_grid ------------------------------------------------------------------------------------
// spaces -> records + cull vectors (host), packed projection records, super-tile lists (k_rf_supercull), one launch per
// chunk of at most tile_max_spaces traverse spaces
static int grid_run(xh_rf *rf, int ns, const float *d_fft, const float *d_ctf, const float *d_mod, const float *d_weights, int n);

static int grid_insert(xh_rf *rf, std::vector<XhSpace> &spaces, const float *d_fft, const float *d_ctf, const float *d_mod,
                       const float *h_weights, int n)
{
    xh_ctx *ctx = rf->ctx;
    const int ns = (int)spaces.size();
    const double br = rf->p.blob_radius;
    // cull vectors, records, per-image weights
    std::vector<float4> cull((size_t)ns * 2);
    std::vector<XgRec> grecs(ns);
    for (int i = 0; i < ns; ++i) xg_fill_rec(grecs[i], cull[i], cull[(size_t)ns + i], spaces[i], br);
    XH_TRY(xh_buf_reserve(ctx, rf->d_cull, sizeof(float4) * cull.size()));
    XH_TRY(xh_buf_reserve(ctx, rf->d_grecs, sizeof(XgRec) * (size_t)ns));
    XH_TRY(stage_upload(rf, rf->d_cull.p, cull.data(), sizeof(float4) * cull.size()));
    XH_TRY(stage_upload(rf, rf->d_grecs.p, grecs.data(), sizeof(XgRec) * (size_t)ns));
    const float *d_weights = nullptr;
    if (h_weights) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_gweights, sizeof(float) * (size_t)n));
        XH_TRY(stage_upload(rf, rf->d_gweights.p, h_weights, sizeof(float) * (size_t)n));
        d_weights = (const float *)rf->d_gweights.p;
    }
    return grid_run(rf, ns, d_fft, d_ctf, d_mod, d_weights, n);
}

// the device side: d_cull ([2][ns] float4) and d_grecs ([ns]) are in place
static int grid_run(xh_rf *rf, int ns, const float *d_fft, const float *d_ctf, const float *d_mod, const float *d_weights, int n)
{
    xh_ctx *ctx = rf->ctx;
    const double br = rf->p.blob_radius;
    const bool fast = rf->p.use_fast != 0;
    XH_CHECK(br < 3.0, XH_ERR_UNSUPPORTED, "xh_rf_insert: blob radius %g: the gridding kernel covers footprints up to 6 x 6 pixels (radius < 3)", br);
    XH_CHECK((rf->mv + 16) / 16 < 1024 && (rf->mv + 8) / 8 < 1024, XH_ERR_UNSUPPORTED, "xh_rf_insert: volume too large for the tile list");
    const int SXp = rf->sizeX + 2 * XG_PAD, SYp = rf->sizeY + 2 * XG_PAD;
    const size_t cells = (size_t)n * SXp * SYp;
    XH_CHECK(cells < ((size_t)1 << 31), XH_ERR_ARG, "xh_rf_insert: more than 2^31 record cells in one call; insert in smaller batches");
    const size_t d = rf->mv + 1;
    float *tempV = rf->d_temp, *tempW = rf->d_temp + 2 * d * d * d;
    {
        // the record buffer is zeroed when it is (re)allocated: the pack kernels leave the cells no tap can reach alone, and what lies there
        // must be finite (it only ever meets the table's zero entry)
        const void *before = rf->d_pack.p;
        XH_TRY(xh_buf_reserve(ctx, rf->d_pack, cells * sizeof(XgCell) + 16));          // (the patch copy reads 16 bytes from the last record too)
        if (rf->d_pack.p != before) XH_HIP(hipMemsetAsync(rf->d_pack.p, 0, rf->d_pack.bytes, ctx->stream));
    }
    // pixels further than sizeX + 2 r (+ 2) from the origin of the half spectrum are beyond every tap (xh_rf_grid.h)
    const int skipR = rf->sizeX + (int)std::ceil(2.0 * br) + 2, skipR2 = rf->skip_far_cells ? skipR * skipR : 0x7fffffff;
    if (rf->packImgs) XH_TRY(fft_cols_rows(rf, rf->packImgs, n, nullptr, (XgCell *)rf->d_pack.p, d_weights));
    else
    for (int i0 = 0; i0 < n; i0 += 65535) {          // blockIdx.y: image
        const int m = std::min(65535, n - i0);
        const size_t o = (size_t)i0 * rf->sizeX * rf->sizeY;
        if (rf->packCtf) {
            const int dc = rf->P / 2, K = std::max(rf->sizeY + XG_PAD - dc, dc + XG_PAD + 1);
            hipLaunchKernelGGL(k_rf_pack_grid_ctf, dim3((unsigned)((K * SXp + 255) / 256), m), dim3(256), 0, ctx->stream, (const xh_cf *)d_fft + o,
                               (const XhCtfDev *)rf->d_ctfp.p + i0, d_weights ? d_weights + i0 : nullptr,
                               (XgCell *)rf->d_pack.p + (size_t)i0 * SXp * SYp, rf->sizeX, rf->sizeY, rf->P, 1.0 / rf->p.sampling,
                               rf->p.min_ctf, rf->p.phase_flipped, skipR2);
        } else
        hipLaunchKernelGGL(k_rf_pack_grid, dim3((unsigned)((SXp * SYp + XG_PACK_CELLS - 1) / XG_PACK_CELLS), m), dim3(256), 0, ctx->stream,
                           (const xh_cf *)d_fft + o, d_ctf ? d_ctf + o : nullptr, d_mod ? d_mod + o : nullptr, d_weights ? d_weights + i0 : nullptr,
                           (XgCell *)rf->d_pack.p + (size_t)i0 * SXp * SYp, m, rf->sizeX, rf->sizeY, skipR2);
        XH_LAUNCH_CHECK();
    }
    // float thresholds equivalent to the double reach tests of the sparse pass (a voxel with no pixel within reach adds
    // nothing): (double)ix + r >= 0, (double)ix - r <= sizeX - 1, the same for iy, are monotone in the float, so the
    // smallest / largest floats that pass, found with the very expressions, decide the same thing
    auto lowest = [&](auto ok, float guess) { float f = guess; while (ok(f)) f = std::nextafterf(f, -INFINITY); while (!ok(f)) f = std::nextafterf(f, INFINITY); return f; };
    auto highest = [&](auto ok, float guess) { float f = guess; while (ok(f)) f = std::nextafterf(f, INFINITY); while (!ok(f)) f = std::nextafterf(f, -INFINITY); return f; };
    const int sX = rf->sizeX, sY = rf->sizeY;
    const float4 reach = make_float4(lowest([&](float f) { return (double)f + br >= 0.0; }, (float)-br),
                                     highest([&](float f) { return (double)f - br <= (double)(sX - 1); }, (float)(sX - 1 + br)),
                                     lowest([&](float f) { return (double)f + br >= 0.0; }, (float)-br),
                                     highest([&](float f) { return (double)f - br <= (double)(sY - 1); }, (float)(sY - 1 + br)));
    // the super-tile lists are sized for the worst case (every projection of the launch in every list): bound the launch
    // so that they stay within 8 GB
    const int superDim = (rf->mv + 1 + 15) / 16;
    const int nsuper = superDim * superDim * superDim;
    const int maxByLists = (int)std::max<size_t>(256, ((size_t)8 << 30) / ((size_t)nsuper * 36));
    const int maxsp = std::min(std::max(64, rf->tile_max_spaces), maxByLists);
    for (int s0 = 0; s0 < ns; s0 += maxsp) {
        const int m = std::min(maxsp, ns - s0);
        XH_HIP(hipMemsetAsync((int *)rf->d_tileCounter.p + 128, 0, sizeof(int) * 256, ctx->stream));
        if (rf->evUsed >= 256) drain_events(rf);
        hipEvent_t ev0 = next_event(rf), ev1 = next_event(rf);
        XH_TRY(xh_buf_reserve(ctx, rf->d_superList, sizeof(int) * (size_t)nsuper * m));
        XH_TRY(xh_buf_reserve(ctx, rf->d_superCount, sizeof(int) * (size_t)nsuper));
        XH_TRY(xh_buf_reserve(ctx, rf->d_superVec, 2 * sizeof(float4) * (size_t)nsuper * m));
        float4 *superN = (float4 *)rf->d_superVec.p, *superX = superN + (size_t)nsuper * m;
        hipLaunchKernelGGL(k_rf_supercull, dim3(nsuper), dim3(256), 0, ctx->stream, (const float4 *)rf->d_cull.p + s0,
                           (const float4 *)rf->d_cull.p + ns + s0, m, rf->mv, fast ? 0.5f : (float)br, superDim, m,
                           (int *)rf->d_superList.p, (int *)rf->d_superCount.p, superN, superX);
        XH_LAUNCH_CHECK();
        if (ev0 && ev1) XH_HIP(hipEventRecord(ev0, ctx->stream));    // the events bracket the gridding kernel alone
        // unit depth and waves per CU: 8-deep units halve the visits (and the patch bytes they fetch) for the price of a larger
        // accumulator block per wave, which the LDS holds for nine waves instead of twelve
        const int uz = rf->unit_z, tl = uz == 8 ? 1 : 0;
        const int nw = rf->grid_waves ? rf->grid_waves : (uz == 8 ? (br < 2.0 || fast ? 9 : 8) : 12);      // (6 x 6 footprints: larger patches)
        // persistent workgroups (one per CU, multiples of eight: XCD classes), or, with a tile budget, as many as the tiles need plus
        // one round of CUs (a workgroup that finds its streams empty retires at once)
        const unsigned cuBlocks = 8u * (unsigned)std::max(1, ctx->num_cus / 8);
        const unsigned gridBlocks = rf->grid_tile_budget > 0 ? cuBlocks + 8u * (unsigned)((rf->ntiles[tl] / rf->grid_tile_budget + 7) / 8) : cuBlocks;
#ifndef XG_ABL
#define XG_ABL 0
#endif
#define XH_GRID(W_, F_, Z_, N_)                                                                                                  \
    hipLaunchKernelGGL((k_rf_grid<W_, F_, Z_, N_, XG_ABL>), dim3(gridBlocks), dim3(64 * N_), 0, ctx->stream, \
                       (const XgRec *)rf->d_grecs.p + s0, (const XgCell *)rf->d_pack.p, (const float *)rf->d_blob.p, tempV, tempW, \
                       rf->mv, rf->iDeltaSqrt, br, (const unsigned *)rf->d_gtiles[tl].p, (const int *)rf->d_tileCounter.p + 32 + 16 * tl, \
                       (int *)rf->d_tileCounter.p + 128, (const int *)rf->d_superList.p, (const int *)rf->d_superCount.p,         \
                       superDim, m, (const float4 *)superN, (const float4 *)superX, reach, rf->grid_tile_budget)
#define XH_GRID_WF(W_, F_)                                                                                                       \
    do {                                                                                                                         \
        if (uz == 8 && nw == 9 && W_ == 4) XH_GRID(4, F_, 8, 9);                                                                 \
        else if (uz == 8 && nw == 8) XH_GRID(W_, F_, 8, 8);                                                                      \
        else if (uz == 4 && nw == 12) XH_GRID(W_, F_, 4, 12);                                                                    \
        else if (uz == 4 && nw == 16 && W_ == 4) XH_GRID(4, F_, 4, 16);                                                          \
        else if (uz == 4 && nw == 8) XH_GRID(W_, F_, 4, 8);                                                                      \
        else { xh_set_error("xh_rf_insert: no gridding kernel for unit_z %d with %d waves", uz, nw); return XH_ERR_UNSUPPORTED; } \
    } while (0)
        if (fast) XH_GRID_WF(4, true);
        else if (br < 2.0) XH_GRID_WF(4, false);
        else XH_GRID_WF(6, false);
#undef XH_GRID_WF
#undef XH_GRID
        XH_LAUNCH_CHECK();
        if (ev0 && ev1) XH_HIP(hipEventRecord(ev1, ctx->stream));
    }
    return XH_OK;
}


// one traverse space per (projection with non-zero weight, symmetry matrix): RFA:939-966
static void build_spaces(xh_rf *rf, const double *h_ainv, const float *h_weights, int n, const double *h_sym, int nsym,
                         std::vector<XhSpace> &spaces)
{
    static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!h_sym) { h_sym = ident; nsym = 1; }
    spaces.reserve((size_t)n * nsym);
    for (int i = 0; i < n; ++i) {
        const float w = h_weights ? h_weights[i] : 1.0f;
        if (h_weights && w == 0.f) continue;  // RFA:327-329
        for (int s = 0; s < nsym; ++s) {
            XhSpace S;
            h_place(S, h_sym + 9 * s, h_ainv + 9 * (size_t)i, rf->mv, rf->p.blob_radius, rf->p.use_fast != 0, w, i);
            spaces.push_back(S);
        }
    }
}

static int insert_common(xh_rf *rf, const float *d_fft, const float *d_ctf, const float *d_mod,
                         const double *h_ainv, const float *h_weights, int n, const double *h_sym, int nsym)
{
    XH_CHECK(rf && d_fft && h_ainv && n >= 0, XH_ERR_ARG, "xh_rf_insert: bad argument");
    XH_CHECK((d_ctf == nullptr) == (d_mod == nullptr), XH_ERR_ARG, "xh_rf_insert: ctf and modulator go together");
    XH_CHECK(!rf->cropped, XH_ERR_STATE, "xh_rf_insert: temp spaces already mirrored/cropped; call xh_rf_reset");
    XH_HIP(hipSetDevice(rf->ctx->device));
    if (n == 0) return XH_OK;
    XH_TRY(ensure_temp(rf));
    std::vector<XhSpace> spaces;
    build_spaces(rf, h_ainv, h_weights, n, h_sym, nsym, spaces);
    const int ns = (int)spaces.size();
    if (ns == 0) return XH_OK;
    // launch order: spaces that share a plane next to each other (k_rf_grid takes the previous visit's voxel queue for them); the
    // order only permutes the float additions of the launch
    if (rf->order_spaces)
        std::stable_sort(spaces.begin(), spaces.end(), [](const XhSpace &a, const XhSpace &b) {
            if (a.tInv[6] != b.tInv[6]) return a.tInv[6] < b.tInv[6];
            if (a.tInv[7] != b.tInv[7]) return a.tInv[7] < b.tInv[7];
            return a.tInv[8] < b.tInv[8];
        });
    return grid_insert(rf, spaces, d_fft, d_ctf, d_mod, h_weights, n);
}

int xh_rf_insert_matrices(xh_rf *rf, const float *d_fft, const float *d_ctf, const float *d_mod,
                          const double *h_ainv, const float *h_weights, int32_t n, const double *h_sym, int32_t nsym)
{
    return insert_common(rf, d_fft, d_ctf, d_mod, h_ainv, h_weights, n, h_sym, nsym);
}

int xh_rf_insert(xh_rf *rf, const float *d_fft, const float *d_ctf, const float *d_mod,
                 const double *h_angles, const float *h_weights, int32_t n, const double *h_sym, int32_t nsym)
{
    XH_CHECK(h_angles && n >= 0, XH_ERR_ARG, "xh_rf_insert: bad argument");
    std::vector<double> ainv((size_t)n * 9);
    for (int i = 0; i < n; ++i) {
        double A[9];
        h_euler(h_angles[3 * i], h_angles[3 * i + 1], h_angles[3 * i + 2], A);
        double *T = &ainv[(size_t)i * 9];   // localAInv = A^T (RFA:348-350)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[c * 3 + r];
    }
    return insert_common(rf, d_fft, d_ctf, d_mod, ainv.data(), h_weights, n, h_sym, nsym);
}

// processBufferGPU in one call (reconstruction_cuda/cuda_gpu_reconstruct_fourier.h:130-157 takes images and does the FFT on the
// device too): shifted images + CTF parameters + orientations -> temp spaces = xh_rf_ctf_arrays + xh_rf_prepare_images +
// xh_rf_insert on scratch owned by the handle. (Letting the column pass of the FFT write the packed records itself was built
// and measured: 5.8 instead of 2.1 + 3.1 ms per 4096 projections -- a block of that kernel owns eight columns, so its record
// writes and CTF-plane reads come in 128- and 32-byte pieces where the pack kernel streams; profiles/README.md.)
int xh_rf_insert_images(xh_rf *rf, const float *d_imgs, const xh_ctf_params *h_ctf, const double *h_angles, const float *h_weights,
                        int32_t n, const double *h_sym, int32_t nsym)
{
    XH_CHECK(rf && d_imgs && h_angles && n >= 0, XH_ERR_ARG, "xh_rf_insert_images: bad argument");
    XH_CHECK(!rf->cropped, XH_ERR_STATE, "xh_rf_insert_images: temp spaces already mirrored/cropped; call xh_rf_reset");
    XH_HIP(hipSetDevice(rf->ctx->device));
    if (n == 0) return XH_OK;
    xh_ctx *ctx = rf->ctx;
    const size_t plane = (size_t)n * rf->sizeX * rf->sizeY;
    float *d_ctf = nullptr, *d_mod = nullptr;
    const bool fuse = h_ctf && rf->fuse_ctf;
    if (fuse) XH_TRY(ctf_params_upload(rf, h_ctf, n));      // the pack kernel evaluates the CTF: no planes
    else if (h_ctf) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_planes, 2 * plane * sizeof(float)));
        d_ctf = (float *)rf->d_planes.p; d_mod = d_ctf + plane;
        XH_TRY(xh_rf_ctf_arrays(rf, h_ctf, n, d_ctf, d_mod));
    }
    const bool fromImages = fuse && rf->records_from_images && fft_cols_rows_ok(rf);   // records straight from the images (A/B)
    if (!fromImages) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_spectra, plane * sizeof(xh_cf)));
        // (the spectra go to the pack kernels only, which skip what no tap can reach: the row pass need not store it)
        const int skipR = rf->sizeX + (int)std::ceil(2.0 * rf->p.blob_radius) + 2;
        rf->fftSkipR2 = rf->skip_far_cells ? skipR * skipR : 0x7fffffff;
        const int rcp = xh_rf_prepare_images(rf, d_imgs, n, (float *)rf->d_spectra.p);
        rf->fftSkipR2 = 0x7fffffff;
        if (rcp != XH_OK) return rcp;
    }
    rf->packCtf = fuse;
    rf->packImgs = fromImages ? d_imgs : nullptr;
    const int rc = xh_rf_insert(rf, fromImages ? d_imgs : (const float *)rf->d_spectra.p, d_ctf, d_mod, h_angles, h_weights, n, h_sym, nsym);
    rf->packCtf = false;
    rf->packImgs = nullptr;
    return rc;
}

// The same with the orientations where the matcher left them: d_angles [n][3] doubles (rot, tilt, psi in degrees),
// d_weights [n] or null. The records are built on the device (k_rf_spaces: the host's own functions, compiled for both);
// nothing in the call waits for the stream.
int xh_rf_insert_images_dev(xh_rf *rf, const float *d_imgs, const xh_ctf_params *h_ctf, const double *d_angles, const float *d_weights,
                            int32_t n, const double *h_sym, int32_t nsym)
{
    XH_CHECK(rf && d_imgs && d_angles && n >= 0, XH_ERR_ARG, "xh_rf_insert_images_dev: bad argument");
    XH_CHECK(!rf->cropped, XH_ERR_STATE, "xh_rf_insert_images_dev: temp spaces already mirrored/cropped; call xh_rf_reset");
    XH_HIP(hipSetDevice(rf->ctx->device));
    if (n == 0) return XH_OK;
    XH_TRY(ensure_temp(rf));
    xh_ctx *ctx = rf->ctx;
    if (!h_sym) nsym = 1;
    XH_CHECK(nsym >= 1 && (size_t)n * nsym < ((size_t)1 << 30), XH_ERR_ARG, "xh_rf_insert_images_dev: bad symmetry count");
    const size_t plane = (size_t)n * rf->sizeX * rf->sizeY;
    float *d_ctf = nullptr, *d_mod = nullptr;
    const bool fuse = h_ctf && rf->fuse_ctf;
    if (fuse) XH_TRY(ctf_params_upload(rf, h_ctf, n));      // the pack kernel evaluates the CTF: no planes
    else if (h_ctf) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_planes, 2 * plane * sizeof(float)));
        d_ctf = (float *)rf->d_planes.p; d_mod = d_ctf + plane;
        XH_TRY(xh_rf_ctf_arrays(rf, h_ctf, n, d_ctf, d_mod));
    }
    const bool fromImages = fuse && rf->records_from_images && fft_cols_rows_ok(rf);   // records straight from the images (A/B)
    if (!fromImages) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_spectra, plane * sizeof(xh_cf)));
        const int skipR = rf->sizeX + (int)std::ceil(2.0 * rf->p.blob_radius) + 2;      // (as in xh_rf_insert_images)
        rf->fftSkipR2 = rf->skip_far_cells ? skipR * skipR : 0x7fffffff;
        const int rcp = xh_rf_prepare_images(rf, d_imgs, n, (float *)rf->d_spectra.p);
        rf->fftSkipR2 = 0x7fffffff;
        if (rcp != XH_OK) return rcp;
    }
    const int ns = n * nsym;
    const double *d_sym = nullptr;
    if (h_sym) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_sym, sizeof(double) * 9 * (size_t)nsym));
        XH_TRY(stage_upload(rf, rf->d_sym.p, h_sym, sizeof(double) * 9 * (size_t)nsym));
        d_sym = (const double *)rf->d_sym.p;
    }
    XH_TRY(xh_buf_reserve(ctx, rf->d_cull, sizeof(float4) * 2 * (size_t)ns));
    XH_TRY(xh_buf_reserve(ctx, rf->d_grecs, sizeof(XgRec) * (size_t)ns));
    const int *d_pos = nullptr;
    if (rf->order_spaces) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_spacePos, sizeof(int) * (size_t)ns));
        hipLaunchKernelGGL(k_rf_space_order, dim3(1), dim3(1024), 0, ctx->stream, d_angles, n, nsym, (int *)rf->d_spacePos.p);
        d_pos = (const int *)rf->d_spacePos.p;
    }
    hipLaunchKernelGGL(k_rf_spaces, dim3((ns + 63) / 64), dim3(64), 0, ctx->stream, d_angles, d_weights, d_sym, n, nsym, rf->mv,
                       rf->p.blob_radius, rf->p.use_fast, (XgRec *)rf->d_grecs.p, (float4 *)rf->d_cull.p, (float4 *)rf->d_cull.p + ns, d_pos);
    XH_LAUNCH_CHECK();
    rf->packCtf = fuse;
    rf->packImgs = fromImages ? d_imgs : nullptr;
    const int rc = grid_run(rf, ns, fromImages ? d_imgs : (const float *)rf->d_spectra.p, d_ctf, d_mod, d_weights, n);
    rf->packCtf = false;
    rf->packImgs = nullptr;
    return rc;
}

int xh_rf_mirror_and_crop(xh_rf *rf)
{
    XH_CHECK(rf, XH_ERR_ARG, "null handle");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_CHECK(!rf->cropped, XH_ERR_STATE, "xh_rf_mirror_and_crop: already cropped");
    XH_TRY(ensure_temp(rf));
    xh_ctx *ctx = rf->ctx;
    const size_t d = rf->mv + 1, nx = rf->mv / 2 + 1;
    const size_t total = d * d * nx;
    XH_TRY(xh_buf_reserve(ctx, rf->d_fin, 3 * total * sizeof(float)));
    float *outV = (float *)rf->d_fin.p, *outW = outV + 2 * total;
    hipLaunchKernelGGL(k_rf_mirror, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const xh_cf *)rf->d_temp, (const float *)(rf->d_temp + 2 * d * d * d), (xh_cf *)outV, outW, rf->mv);
    XH_LAUNCH_CHECK();
    XH_HIP(hipMemcpyAsync(rf->d_temp, outV, 3 * total * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    rf->cropped = true;
    return XH_OK;
}

// ---- the cropped spaces as data: half-set bookkeeping (RF:991-1053) and the sum of per-device partial
// reconstructions on one node (what mpi_reconstruct_fourier_accel.cpp:245-266 does with MPI_Reduce)
__global__ void __launch_bounds__(256) k_rf_add(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = dst[i];
        const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}
__global__ void __launch_bounds__(256) k_rf_add_tail(float *__restrict__ dst, const float *__restrict__ src, size_t first, size_t n)
{
    const size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

static int rf_add_into(xh_rf *rf, const float *d_src)
{
    xh_ctx *ctx = rf->ctx;
    const size_t n = xh_rf_cropped_floats(rf), n4 = n / 4;
    hipLaunchKernelGGL(k_rf_add, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, (size_t)ctx->num_cus * 16)), dim3(256), 0, ctx->stream,
                       (float4 *)rf->d_temp, (const float4 *)d_src, n4);
    XH_LAUNCH_CHECK();
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(k_rf_add_tail, dim3(1), dim3(256), 0, ctx->stream, rf->d_temp, d_src, n4 * 4, n);
        XH_LAUNCH_CHECK();
    }
    return XH_OK;
}

int xh_rf_cropped_export(xh_rf *rf, float *d_dst)
{
    XH_CHECK(rf && d_dst, XH_ERR_ARG, "null argument");
    XH_CHECK(rf->cropped, XH_ERR_STATE, "xh_rf_cropped_export: call xh_rf_mirror_and_crop first");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_HIP(hipMemcpyAsync(d_dst, rf->d_temp, sizeof(float) * xh_rf_cropped_floats(rf), hipMemcpyDeviceToDevice, rf->ctx->stream));
    return XH_OK;
}

int xh_rf_cropped_import(xh_rf *rf, const float *d_src, int32_t add)
{
    XH_CHECK(rf && d_src, XH_ERR_ARG, "null argument");
    XH_CHECK(!add || rf->cropped, XH_ERR_STATE, "xh_rf_cropped_import(add): nothing cropped to add to");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_TRY(ensure_temp(rf));
    if (add) return rf_add_into(rf, d_src);
    XH_HIP(hipMemcpyAsync(rf->d_temp, d_src, sizeof(float) * xh_rf_cropped_floats(rf), hipMemcpyDeviceToDevice, rf->ctx->stream));
    rf->cropped = true;
    return XH_OK;
}

// ---- sum over the devices of one process: one RCCL all-reduce (parallel/mpi_reconstruct_fourier_accel.cpp:245-266) ----
// RCCL is bound at run time (dlopen): a process that already carries a copy (torch ships one) keeps using it, and a
// single-device run never loads it.
namespace {
struct XhRccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<int> devs;             // the communicators below span these devices, in this order
    std::vector<ncclComm_t> comms;
};
XhRccl g_rccl;
bool rccl_load()
{
    if (g_rccl.lib) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;      // a copy the process already has
    if (!h) for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return false;
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(h, "ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) return false;
    g_rccl.lib = h;
    return true;
}
// the earlier exchange, kept as the loud fallback: binary tree, peer copies into the receiver's scratch + one add kernel
int reduce_tree(xh_rf *const *rfs, int n, size_t bytes)
{
    for (int stride = 1; stride < n; stride <<= 1) {
        for (int i = 0; i + stride < n; i += 2 * stride) {
            xh_rf *src = rfs[i + stride];
            XH_HIP(hipSetDevice(src->ctx->device));
            XH_HIP(hipStreamSynchronize(src->ctx->stream));
        }
        for (int i = 0; i + stride < n; i += 2 * stride) {
            xh_rf *dst = rfs[i], *src = rfs[i + stride];
            XH_HIP(hipSetDevice(dst->ctx->device));
            const float *from = src->d_temp;
            if (src->ctx->device != dst->ctx->device) {
                XH_TRY(xh_buf_reserve(dst->ctx, dst->d_fin, bytes));
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dst->ctx->device, src->ctx->device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(src->ctx->device, 0);   // already enabled is fine
                (void)hipGetLastError();
                XH_HIP(hipMemcpyPeerAsync(dst->d_fin.p, dst->ctx->device, src->d_temp, src->ctx->device, bytes, dst->ctx->stream));
                from = (const float *)dst->d_fin.p;
            }
            XH_TRY(rf_add_into(dst, from));
        }
        for (int i = 0; i + stride < n; i += 2 * stride) {
            xh_rf *dst = rfs[i];
            XH_HIP(hipSetDevice(dst->ctx->device));
            XH_HIP(hipStreamSynchronize(dst->ctx->stream));
        }
    }
    return XH_OK;
}
}  // namespace

int xh_rf_reduce(xh_rf *const *rfs, int32_t n)
{
    XH_CHECK(rfs && n >= 1, XH_ERR_ARG, "xh_rf_reduce: no handles");
    for (int i = 0; i < n; ++i) {
        XH_CHECK(rfs[i], XH_ERR_ARG, "xh_rf_reduce: null handle");
        XH_CHECK(rfs[i]->cropped, XH_ERR_STATE, "xh_rf_reduce: call xh_rf_mirror_and_crop on every handle first");
        XH_CHECK(rfs[i]->mv == rfs[0]->mv, XH_ERR_ARG, "xh_rf_reduce: handles of different geometry");
        for (int j = 0; j < i; ++j) XH_CHECK(rfs[j] != rfs[i], XH_ERR_ARG, "xh_rf_reduce: the same handle twice");
    }
    const size_t count = xh_rf_cropped_floats(rfs[0]);
    const size_t bytes = sizeof(float) * count;
    // handles that share a device are summed there first (one streaming add each); the first handle of every device
    // then takes part in the exchange, rfs[0] among them
    std::vector<xh_rf *> lead;
    for (int i = 0; i < n; ++i) {
        xh_rf *l = nullptr;
        for (xh_rf *c : lead) if (c->ctx->device == rfs[i]->ctx->device) l = c;
        if (!l) { lead.push_back(rfs[i]); continue; }
        XH_HIP(hipSetDevice(l->ctx->device));
        XH_HIP(hipStreamSynchronize(rfs[i]->ctx->stream));
        XH_TRY(rf_add_into(l, rfs[i]->d_temp));
    }
    const int nd = (int)lead.size();
    for (xh_rf *l : lead) { XH_HIP(hipSetDevice(l->ctx->device)); XH_HIP(hipStreamSynchronize(l->ctx->stream)); }
    if (nd > 1) {
        std::vector<int> devs(nd);
        for (int i = 0; i < nd; ++i) devs[i] = lead[i]->ctx->device;
        bool ok = rccl_load();
        if (ok && g_rccl.devs != devs) {
            for (ncclComm_t c : g_rccl.comms) g_rccl.CommDestroy(c);
            g_rccl.comms.assign(nd, nullptr);
            g_rccl.devs.clear();
            const ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), nd, devs.data());
            if (r == ncclSuccess) g_rccl.devs = devs;
            else { g_rccl.comms.clear(); ok = false; fprintf(stderr, "xh_rf_reduce: ncclCommInitAll failed (%s)\n", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); }
        }
        if (ok) {
            // ONE in-place all-reduce of [volume | weights] over every xGMI link; every leader ends up with the sum
            ncclResult_t r = g_rccl.GroupStart();
            for (int i = 0; i < nd && r == ncclSuccess; ++i)
                r = g_rccl.AllReduce(lead[i]->d_temp, lead[i]->d_temp, count, ncclFloat, ncclSum, g_rccl.comms[i], lead[i]->ctx->stream);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess) {
                xh_set_error("xh_rf_reduce: RCCL all-reduce failed (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(r != ncclSuccess ? r : r2) : "?");
                return XH_ERR_HIP;
            }
            for (xh_rf *l : lead) { XH_HIP(hipSetDevice(l->ctx->device)); XH_HIP(hipStreamSynchronize(l->ctx->stream)); }
        } else {
            fprintf(stderr, "xh_rf_reduce: RCCL is not available in this process; falling back to the peer-copy tree "
                            "(one xGMI link per pair and level instead of all of them)\n");
            XH_TRY(reduce_tree(lead.data(), nd, bytes));
        }
    }
    XH_HIP(hipSetDevice(rfs[0]->ctx->device));
    return XH_OK;
}

int xh_rf_finish(xh_rf *rf, double *h_volume)
{
    XH_CHECK(rf && h_volume, XH_ERR_ARG, "null argument");
    XH_HIP(hipSetDevice(rf->ctx->device));
    XH_CHECK(rf->cropped, XH_ERR_STATE, "xh_rf_finish: call xh_rf_mirror_and_crop first (RFA:149-152)");
    xh_ctx *ctx = rf->ctx;
    const int mv = rf->mv, P = rf->P, D = rf->D, xh = P / 2 + 1;
    const size_t d = mv + 1, nx = mv / 2 + 1, total = d * d * nx;
    xh_cf *V = (xh_cf *)rf->d_temp;
    float *W = rf->d_temp + 2 * total;
    if (rf->p.use_fast) {
        XH_TRY(xh_buf_reserve(ctx, rf->d_fin, 3 * total * sizeof(float)));
        float *tv = (float *)rf->d_fin.p, *tw = tv + 2 * total;
        hipLaunchKernelGGL((k_rf_applyblob<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float *)V, tv, (const float *)rf->d_blob.p, mv, (float)rf->p.blob_radius, rf->iDeltaSqrt, 2);
        XH_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_rf_applyblob<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float *)W, tw, (const float *)rf->d_blob.p, mv, (float)rf->p.blob_radius, rf->iDeltaSqrt, 1);
        XH_LAUNCH_CHECK();
        XH_HIP(hipMemcpyAsync(rf->d_temp, tv, 3 * total * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(k_rf_hermitian, dim3((unsigned)((d * (mv / 2 + 1) + 255) / 256)), dim3(256), 0, ctx->stream, V, W, mv);
    XH_LAUNCH_CHECK();
    const float corr2D_3D = std::pow(rf->p.padding_proj, 2.) / (D * std::pow(rf->p.padding_vol, 3.));
    hipLaunchKernelGGL(k_rf_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, V, (const float *)W, total, corr2D_3D);
    XH_LAUNCH_CHECK();
    // expanded spectrum + output volume live in scratch
    const size_t specElems = (size_t)P * P * xh;
    // (kept with the handle: allocating and freeing 1.2 GB per call -- hipFree waits for the device -- was most of the 36-60 ms a
    // finish took; its kernels are 10 ms)
    XhBuf &spec = rf->d_finSpec, &vol = rf->d_finVol, &fbt = rf->d_finFbt;
    int r = xh_buf_reserve(ctx, spec, specElems * sizeof(xh_cd));
    if (r == XH_OK) r = xh_buf_reserve(ctx, vol, sizeof(double) * (size_t)D * D * D);
    if (r == XH_OK) r = xh_buf_reserve(ctx, fbt, sizeof(double) * XH_BLOB_TABLE);
    if (r != XH_OK) return r;
    auto cleanup = [&]() {};
#define XH_HIP_C(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { xh_set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return XH_ERR_HIP; } } while (0)
    XH_HIP_C(hipMemcpyAsync(fbt.p, rf->fourierBlobTable.data(), fbt.bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rf_expand, dim3((unsigned)((specElems + 255) / 256)), dim3(256), 0, ctx->stream, (const xh_cf *)V, (xh_cd *)spec.p, mv, P);
    XH_HIP_C(hipGetLastError());
    {
        const int rc = finish_from_spectrum(ctx, rf->planP64.plan, (xh_cd *)spec.p, (double *)vol.p, (const double *)fbt.p, D, (double)rf->iDeltaFourier,
                                            rf->p.padding_proj, rf->p.padding_vol, rf->meanFactor2, true, h_volume);
        if (rc != XH_OK) { cleanup(); return rc; }
    }
#undef XH_HIP_C
    cleanup();
    return XH_OK;
}


}  // extern "C"

// =====================================================================================================================
// ProgRecFourier's own arithmetic (reconstruction/reconstruct_fourier.cpp, "RF"): the program behind the name
// xmipp_reconstruct_fourier. Everything in double: the padded projection's transform (RF:386-404), the image-driven scatter
// of every Fourier pixel within --max_resolution into the FFTW-layout volume with wrap and, beyond the half, the point-mirrored
// conjugated slot (RF:571-793), correctWeight with its re-processing passes (RF:1056-1101, forceWeightSymmetry RF:1186-1221),
// enforceHermitianSymmetry + PROCESS_WEIGHTS (RF:451-480,1103-1126) and the shared finaliser tail. The scatter adds with
// double-precision atomics: the sums differ from the reference's sequential ones by their order only (1e-16 relative). This is
// BASELINE config 1's plumbing path: sized for exactness, not for throughput (the accel program is the fast one).
namespace {
__device__ __forceinline__ int d_wrap(int x, int n) { int r = x % n; return r < 0 ? r + n : r; }       // intWRAP(x, 0, n - 1)

// zero-pad about the Xmipp origin + CenterFFT(true) (RF:386-401): image pixel (i, j) lands at ((i + off + P/2) % P, ...)
__global__ void __launch_bounds__(256) k_rf2_pad(const float *__restrict__ imgs, int D, xh_cd *__restrict__ out, int P, int n)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)n * D * D) return;
    const int img = (int)(t / ((size_t)D * D)), r = (int)(t - (size_t)img * D * D), i = r / D, j = r - i * D;
    const int off = -(D / 2) + P / 2, sh = P / 2;
    out[((size_t)img * P + (i + off + sh) % P) * P + (j + off + sh) % P] = xh_cd{(double)imgs[t], 0.0};
}

// CTF value of RF:600-625 (getValuePureNoKAt at the double digital frequency / Ts)
__device__ __forceinline__ double d_rf2_ctf(const XhCtfDev &c, double X, double Y)
{
    const double u2 = X * X + Y * Y, u = sqrt(u2), u4 = u2 * u2;
    double deltaf;
    if (fabs(X) < 1e-6 && fabs(Y) < 1e-6) deltaf = 0;
    else deltaf = c.defocus_average + c.defocus_deviation * cos(2 * (atan2(Y, X) - c.rad_azimuth));
    double VPP = 0.0;
    if (round(c.VPP_radius * 1000) != 0) VPP = -c.phase_shift * (1 - exp(-u2 / (2 * c.VPP_radius * c.VPP_radius)));
    const double argument = VPP + c.K1 * deltaf * u2 + c.K2 * u4;
    double sine_part, cosine_part;
    sincos(argument, &sine_part, &cosine_part);
    const double Eespr = exp(-c.K3 * u4);
    const double EdeltaF = d_bessj0(c.K5 * u2);
    const double xs = u * c.DeltaR;
    const double EdeltaR = (xs == 0) ? 1.0 : sin(3.14159265358979323846 * xs) / (3.14159265358979323846 * xs);
    const double aux = (c.K7 * u2 * u + deltaf * u);
    const double Ealpha = exp(-c.K6 * aux * aux);
    double E = Eespr * EdeltaF * EdeltaR * Ealpha + c.envR0 + c.envR1 * u + c.envR2 * u2;
    if (E < 0) E = 0;
    return c.K * (-c.K * (c.Ksin * sine_part - c.Kcos * cosine_part) * E);
}

// one thread per (projection x symmetry matrix, Fourier pixel of the half spectrum)
__global__ void __launch_bounds__(256)
k_rf2_scatter(const xh_cd *__restrict__ spectra, const double *__restrict__ A_SL, const float *__restrict__ weights, const int *__restrict__ imgOf,
              const XhCtfDev *__restrict__ ctf, int nspaces, int P, int V, double maxRes2, double radius, double iDeltaSqrt,
              const double *__restrict__ table, xh_cd *__restrict__ F, double *__restrict__ W, double iTs, double minCTF, int phaseFlipped,
              int reprocess)
{
    const int pxh = P / 2 + 1, xh = V / 2 + 1;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)nspaces * P * pxh) return;
    const int sp = (int)(t / ((size_t)P * pxh)), r = (int)(t - (size_t)sp * P * pxh), i = r / pxh, j = r - i * pxh;
    const int img = imgOf[sp];
    const double weight = weights ? (double)weights[img] : 1.0;
    if (weight == 0.0) return;
    const double fx = (j <= (P >> 1)) ? (double)j / P : (double)(j - P) / P, fy = (i <= (P >> 1)) ? (double)i / P : (double)(i - P) / P;
    if (fx * fx + fy * fy > maxRes2) return;
    double wCTF = 1, wMod = 1;
    if (ctf && !reprocess) {
        wCTF = d_rf2_ctf(ctf[img], fx * iTs, fy * iTs);
        if (isnan(wCTF)) { if (i == 0 && j == 0) wMod = wCTF = 1.0; else wMod = wCTF = 0.0; }
        if (fabs(wCTF) < minCTF) { wMod = fabs(wCTF); wCTF = (wCTF >= 0) ? 1.0 : -1.0; }
        else wCTF = 1.0 / wCTF;
        if (phaseFlipped) wCTF = fabs(wCTF);
    }
    const double *A = A_SL + 9 * (size_t)sp;
    const double rx = (A[0] * fx + A[1] * fy) * V, ry = (A[3] * fx + A[4] * fy) * V, rz = (A[6] * fx + A[7] * fy) * V;
    const int x1 = (int)ceil(rx - radius), x2 = (int)floor(rx + radius);
    const int y1 = (int)ceil(ry - radius), y2 = (int)floor(ry + radius);
    const int z1 = (int)ceil(rz - radius), z2 = (int)floor(rz + radius);
    const double r2 = radius * radius;
    const xh_cd in = reprocess ? xh_cd{0., 0.} : spectra[((size_t)img * P + i) * P + j];
    const int xsize_1 = xh - 1;
    for (int iz = z1; iz <= z2; ++iz) {
        const double dz = iz - rz, z2v = dz * dz;
        const int kz = d_wrap(iz, V), kzn = d_wrap(-kz, V);
        for (int iy = y1; iy <= y2; ++iy) {
            const double dy = iy - ry, y2z2 = dy * dy + z2v;
            if (y2z2 > r2) continue;
            const int ky = d_wrap(iy, V), kyn = d_wrap(-ky, V);
            for (int ix = x1; ix <= x2; ++ix) {
                const double dx = ix - rx, d2 = dx * dx + y2z2;
                if (d2 > r2) continue;
                const double w = table[(int)(d2 * iDeltaSqrt + 0.5)] * weight * wMod;
                const int kx = d_wrap(ix, V);
                bool cj = false;
                int pz = kz, py = ky, px = kx;
                if (kx > xsize_1) { pz = kzn; py = kyn; px = d_wrap(-kx, V); cj = true; }
                const size_t o = ((size_t)pz * V + py) * xh + px;
                if (reprocess) atomicAdd(&W[o], w * F[o].x);            // RF:770-775: F holds the current 1 / w estimate
                else {
                    const double we = w * wCTF;
                    atomicAdd(&F[o].x, we * in.x);
                    atomicAdd(&F[o].y, cj ? -we * in.y : we * in.y);
                    atomicAdd(&W[o], w);
                }
            }
        }
    }
}

// forceWeightSymmetry (RF:1186-1221) / enforceHermitianSymmetry (xmippCore xmipp_fftw.cpp, 3-D) on the x = 0 plane:
// mode 0 averages the weights of (k, i, 0) and (-k, -i, 0), mode 1 the coefficients with their conjugated point mirror
__global__ void __launch_bounds__(256) k_rf2_plane_symmetry(xh_cd *__restrict__ F, double *__restrict__ W, int V, int mode)
{
    const int xh = V / 2 + 1;
    int half = V / 2; if (V % 2 == 0) half--;
    const int t = blockIdx.x * 256 + threadIdx.x;
    // threads [0, V * half): rows i = 1 .. half of every k; then [V * half, V * half + half): the column i = 0, k = 1 .. half
    int k, i;
    if (t < V * half) { k = t / half; i = t - k * half + 1; }
    else if (t < V * half + half) { k = t - V * half + 1; i = 0; }
    else return;
    const int ks = d_wrap(-k, V), is = d_wrap(-i, V);
    const size_t a = ((size_t)k * V + i) * xh, b = ((size_t)ks * V + is) * xh;
    if (mode == 0) { const double m = 0.5 * (W[a] + W[b]); W[a] = W[b] = m; }
    else { const xh_cd m = xh_cd{0.5 * (F[a].x + F[b].x), 0.5 * (F[a].y - F[b].y)}; F[a] = m; F[b] = xh_cd{m.x, -m.y}; }
}

// the element-wise steps of correctWeight (RF:1056-1101): 0 weights = 1; 1 F.re = 1 / w where |w| > 1e-3; 2 F.re /= w where |w| > 1e-3;
// 3 w = F.re
__global__ void __launch_bounds__(256) k_rf2_weight_step(xh_cd *__restrict__ F, double *__restrict__ W, size_t n, int step)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    if (step == 0) W[t] = 1;
    else if (step == 1) { if (fabs(W[t]) > 1e-3) F[t].x = 1.0 / W[t]; }
    else if (step == 2) { if (fabs(W[t]) > 1e-3) F[t].x = F[t].x / W[t]; }
    else W[t] = F[t].x;
}

// PROCESS_WEIGHTS (RF:451-480)
__global__ void __launch_bounds__(256) k_rf2_process_weights(const xh_cd *__restrict__ F, const double *__restrict__ W, xh_cd *__restrict__ out, size_t n,
                                                             double corr2D_3D, int niter)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    xh_cd v = F[t];
    if (niter == 0) { v.x *= corr2D_3D; v.y *= corr2D_3D; }
    else {
        const double w = W[t];
        if (1.0 / w > 1e-6) { v.x *= corr2D_3D * w; v.y *= corr2D_3D * w; }
        else v = xh_cd{0., 0.};
    }
    out[t] = v;
}
}  // namespace

struct xh_rf2 {
    xh_ctx *ctx;
    xh_rf_params p;
    int D, P, V, xh, niter;
    double iDeltaSqrt, iDeltaFourier, meanFactor2;
    std::vector<double> table, ftable;
    XhPlanBufs<double> planP, planV;
    XhBuf d_F, d_W, d_Fsave, d_table, d_spec, d_A, d_img, d_w, d_ctf, d_imgOf;
    size_t nF;
};

extern "C" {

int xh_rf2_destroy(xh_rf2 *h)
{
    if (!h) return XH_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    xh_plan_free(h->planP); xh_plan_free(h->planV);
    xh_buf_free(h->d_F); xh_buf_free(h->d_W); xh_buf_free(h->d_Fsave); xh_buf_free(h->d_table); xh_buf_free(h->d_spec); xh_buf_free(h->d_A);
    xh_buf_free(h->d_img); xh_buf_free(h->d_w); xh_buf_free(h->d_ctf); xh_buf_free(h->d_imgOf);
    delete h;
    return XH_OK;
}

int xh_rf2_reset(xh_rf2 *h)
{
    XH_CHECK(h, XH_ERR_ARG, "xh_rf2_reset: null handle");
    XH_HIP(hipSetDevice(h->ctx->device));
    XH_HIP(hipMemsetAsync(h->d_F.p, 0, h->d_F.bytes, h->ctx->stream));
    XH_HIP(hipMemsetAsync(h->d_W.p, 0, h->d_W.bytes, h->ctx->stream));
    return XH_OK;
}

int xh_rf2_create(xh_ctx *ctx, const xh_rf_params *p, int32_t niter_weight, xh_rf2 **out)
{
    XH_CHECK(ctx && p && out && niter_weight >= 0, XH_ERR_ARG, "xh_rf2_create: bad argument");
    XH_CHECK(p->imgSize >= 4 && (p->blob_order == 0 || p->blob_order == 2), XH_ERR_ARG, "xh_rf2_create: bad image size or blob order");
    XH_CHECK(!p->use_fast, XH_ERR_UNSUPPORTED, "xh_rf2_create: ProgRecFourier has no --fast");
    XH_HIP(hipSetDevice(ctx->device));
    xh_rf2 *h = new xh_rf2;
    h->ctx = ctx; h->p = *p; h->D = p->imgSize; h->niter = niter_weight;
    h->P = (int)(h->D * p->padding_proj);            // RF:229-231
    h->V = (int)(h->D * p->padding_vol);
    h->xh = h->V / 2 + 1;
    h->meanFactor2 = -1;
    h->nF = (size_t)h->V * h->V * h->xh;
    // tables, RF:222-269 (the same as RFA's, kept in double)
    const int Xdim = h->D;
    h->table.resize(XH_BLOB_TABLE); h->ftable.resize(XH_BLOB_TABLE);
    const double rFourier = p->blob_radius / (p->padding_vol * Xdim), rNorm = p->blob_radius / (p->padding_proj / p->padding_vol);
    const double deltaSqrt = (p->blob_radius * p->blob_radius) / (XH_BLOB_TABLE - 1), deltaFourier = (std::sqrt(3.) * Xdim / 2.) / (XH_BLOB_TABLE - 1);
    const double iw0 = 1.0 / h_kaiser_fourier(0.0, rNorm, p->blob_alpha, p->blob_order);
    double pad3 = p->padding_vol * Xdim;
    pad3 = pad3 * pad3 * pad3;
    const double tsz = p->blob_radius * std::sqrt(1. / (XH_BLOB_TABLE - 1));
    for (int i = 0; i < XH_BLOB_TABLE; ++i) {
        h->table[i] = h_kaiser_value(tsz * std::sqrt((double)i), p->blob_radius, p->blob_alpha, p->blob_order) * iw0;
        h->ftable[i] = h_kaiser_fourier(deltaFourier * i, rFourier, p->blob_alpha, p->blob_order) * pad3 * iw0;
    }
    h->iDeltaSqrt = 1 / deltaSqrt; h->iDeltaFourier = 1 / deltaFourier;
    int rc = (h->P <= 2048 && h->V <= 2048) ? XH_OK : XH_ERR_UNSUPPORTED;
    if (rc != XH_OK) xh_set_error("xh_rf2_create: padded sizes %d / %d exceed 2048", h->P, h->V);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, h->P, h->planP);
    if (rc == XH_OK) rc = xh_plan_create<double>(ctx, h->V, h->planV);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->d_F, sizeof(xh_cd) * h->nF);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->d_W, sizeof(double) * h->nF);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, h->d_table, sizeof(double) * 2 * XH_BLOB_TABLE);
    if (rc == XH_OK && hipMemcpy(h->d_table.p, h->table.data(), sizeof(double) * XH_BLOB_TABLE, hipMemcpyHostToDevice) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK && hipMemcpy((double *)h->d_table.p + XH_BLOB_TABLE, h->ftable.data(), sizeof(double) * XH_BLOB_TABLE, hipMemcpyHostToDevice) != hipSuccess) rc = XH_ERR_HIP;
    if (rc == XH_OK) rc = xh_rf2_reset(h);
    if (rc != XH_OK) { xh_rf2_destroy(h); return rc; }
    *out = h;
    return XH_OK;
}

// n projections (shifts already applied), their CTFs (nullable), orientations (rot, tilt, psi), weights (nullable), symmetry
// matrices (nullable = identity). reprocess != 0: the weight re-processing pass of correctWeight (images ignored).
int xh_rf2_insert(xh_rf2 *h, const float *d_imgs, const xh_ctf_params *h_ctf, const double *h_angles, const float *h_weights, int32_t n,
                  const double *h_sym, int32_t nsym, int32_t reprocess)
{
    XH_CHECK(h && h_angles && n >= 0 && (reprocess || d_imgs), XH_ERR_ARG, "xh_rf2_insert: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    if (n == 0) return XH_OK;
    static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!h_sym) { h_sym = ident; nsym = 1; }
    const int P = h->P, D = h->D, pxh = P / 2 + 1;
    // A_SL = R * localAInv per (projection, symmetry matrix), RF:560-566
    std::vector<double> A((size_t)n * nsym * 9);
    std::vector<int> imgOf((size_t)n * nsym);
    for (int i = 0; i < n; ++i) {
        double E[9], T[9];
        h_euler(h_angles[3 * i], h_angles[3 * i + 1], h_angles[3 * i + 2], E);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[r * 3 + c] = E[c * 3 + r];
        for (int s_ = 0; s_ < nsym; ++s_) {
            double *o = &A[((size_t)i * nsym + s_) * 9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += h_sym[9 * s_ + r * 3 + k] * T[k * 3 + c];
                    o[r * 3 + c] = acc;
                }
            imgOf[(size_t)i * nsym + s_] = i;
        }
    }
    XH_TRY(xh_buf_reserve(ctx, h->d_A, sizeof(double) * A.size()));
    XH_TRY(xh_buf_reserve(ctx, h->d_imgOf, sizeof(int) * imgOf.size()));
    XH_HIP(hipMemcpyAsync(h->d_A.p, A.data(), sizeof(double) * A.size(), hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(hipMemcpyAsync(h->d_imgOf.p, imgOf.data(), sizeof(int) * imgOf.size(), hipMemcpyHostToDevice, ctx->stream));
    const float *d_w = nullptr;
    if (h_weights) {
        XH_TRY(xh_buf_reserve(ctx, h->d_w, sizeof(float) * (size_t)n));
        XH_HIP(hipMemcpyAsync(h->d_w.p, h_weights, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        d_w = (const float *)h->d_w.p;
    }
    const XhCtfDev *d_c = nullptr;
    std::vector<XhCtfDev> hc;
    if (h_ctf && !reprocess) {
        hc.resize(n);
        for (int i = 0; i < n; ++i) {
            const xh_ctf_params &c = h_ctf[i];
            const double local_Cs = c.Cs * 1e7, local_Ca = c.Ca * 1e7, local_kV = c.kV * 1e3, local_ispr = c.ispr * 1e6;
            const double lambda = 12.2643247 / std::sqrt(local_kV * (1. + 0.978466e-6 * local_kV));
            XhCtfDev &d = hc[i];
            d.K1 = kPI * lambda; d.K2 = kPI / 2 * local_Cs * lambda * lambda * lambda;
            d.K3 = std::pow(0.25 * kPI * local_Ca * lambda * (c.espr / c.kV + 2 * local_ispr), 2) / std::log(2.0);
            d.K5 = kPI * c.DeltaF * lambda; d.K6 = kPI * kPI * c.alpha * c.alpha; d.K7 = local_Cs * lambda * lambda;
            d.Ksin = std::sqrt(1 - c.Q0 * c.Q0); d.Kcos = c.Q0; d.rad_azimuth = c.azimuthal_angle * kPI / 180.;
            d.defocus_average = -(c.DeltafU + c.DeltafV) * 0.5; d.defocus_deviation = -(c.DeltafU - c.DeltafV) * 0.5;
            d.DeltaR = c.DeltaR; d.K = c.K; d.envR0 = c.envR0; d.envR1 = c.envR1; d.envR2 = c.envR2;
            d.phase_shift = c.phase_shift; d.VPP_radius = c.VPP_radius;
        }
        XH_TRY(xh_buf_reserve(ctx, h->d_ctf, sizeof(XhCtfDev) * (size_t)n));
        XH_HIP(hipMemcpyAsync(h->d_ctf.p, hc.data(), sizeof(XhCtfDev) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        d_c = (const XhCtfDev *)h->d_ctf.p;
    }
    if (!reprocess) {
        // RF:386-404: pad, centre, transform (normalised by 1 / P^2)
        const size_t tot = (size_t)n * P * P;
        XH_TRY(xh_buf_reserve(ctx, h->d_spec, sizeof(xh_cd) * tot));
        XH_HIP(hipMemsetAsync(h->d_spec.p, 0, sizeof(xh_cd) * tot, ctx->stream));
        hipLaunchKernelGGL(k_rf2_pad, dim3((unsigned)(((size_t)n * D * D + 255) / 256)), dim3(256), 0, ctx->stream, d_imgs, D, (xh_cd *)h->d_spec.p, P, n);
        XH_LAUNCH_CHECK();
        const XhPlan<double> &plan = h->planP.plan;
        const int lpb = xh_plan_lpb(plan, 64 * 1024, 16);
        const size_t smem = ((size_t)lpb * sizeof(xh_cd)) << plan.logM;
        const size_t nlines = (size_t)n * P;
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, (xh_cd *)h->d_spec.p, plan,
                           nlines, (size_t)1, (size_t)P, (size_t)0, (size_t)1, lpb);
        XH_LAUNCH_CHECK();
        hipLaunchKernelGGL((xh_k_fft_lines<double, false>), dim3((unsigned)((nlines + lpb - 1) / lpb)), dim3(256), smem, ctx->stream, (xh_cd *)h->d_spec.p, plan,
                           nlines, (size_t)P, (size_t)P * P, (size_t)1, (size_t)P, lpb);
        XH_LAUNCH_CHECK();
        const double inv = 1.0 / ((double)P * P);
        hipLaunchKernelGGL(k_rf_scale_cd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (xh_cd *)h->d_spec.p, tot, inv);
        XH_LAUNCH_CHECK();
    }
    const size_t threads = (size_t)n * nsym * P * pxh;
    hipLaunchKernelGGL(k_rf2_scatter, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, (const xh_cd *)h->d_spec.p, (const double *)h->d_A.p,
                       d_w, (const int *)h->d_imgOf.p, d_c, n * nsym, P, h->V, h->p.max_resolution * h->p.max_resolution, h->p.blob_radius, h->iDeltaSqrt,
                       (const double *)h->d_table.p, (xh_cd *)h->d_F.p, (double *)h->d_W.p, 1.0 / h->p.sampling, h->p.min_ctf, h->p.phase_flipped, reprocess);
    XH_LAUNCH_CHECK();
    XH_HIP(hipStreamSynchronize(ctx->stream));          // the host arrays of this call go out of scope
    return XH_OK;
}

// correctWeight, split so that the caller replays the projections (xh_rf2_insert with reprocess = 1) between the steps:
//   step 0 begin;  niter - 1 times { step 1 iteration begin; replay; step 2 iteration end };  step 3 end
int xh_rf2_weights_step(xh_rf2 *h, int32_t step)
{
    XH_CHECK(h && step >= 0 && step <= 3, XH_ERR_ARG, "xh_rf2_weights_step: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int V = h->V;
    int half = V / 2; if (V % 2 == 0) half--;
    const unsigned nbSym = (unsigned)((V * half + half + 255) / 256), nbAll = (unsigned)((h->nF + 255) / 256);
    auto sym = [&]() { hipLaunchKernelGGL(k_rf2_plane_symmetry, dim3(nbSym), dim3(256), 0, ctx->stream, (xh_cd *)h->d_F.p, (double *)h->d_W.p, V, 0); };
    auto el = [&](int s_) { hipLaunchKernelGGL(k_rf2_weight_step, dim3(nbAll), dim3(256), 0, ctx->stream, (xh_cd *)h->d_F.p, (double *)h->d_W.p, h->nF, s_); };
    if (step == 0) {
        sym();
        if (h->niter == 0) { el(0); XH_LAUNCH_CHECK(); return XH_OK; }
        XH_TRY(xh_buf_reserve(ctx, h->d_Fsave, sizeof(xh_cd) * h->nF));
        XH_HIP(hipMemcpyAsync(h->d_Fsave.p, h->d_F.p, sizeof(xh_cd) * h->nF, hipMemcpyDeviceToDevice, ctx->stream));
        sym();
        el(1);
    } else if (step == 1) XH_HIP(hipMemsetAsync(h->d_W.p, 0, sizeof(double) * h->nF, ctx->stream));
    else if (step == 2) { sym(); el(2); }
    else if (h->niter != 0) {
        el(3);
        XH_HIP(hipMemcpyAsync(h->d_F.p, h->d_Fsave.p, sizeof(xh_cd) * h->nF, hipMemcpyDeviceToDevice, ctx->stream));
    }
    XH_LAUNCH_CHECK();
    return XH_OK;
}

size_t xh_rf2_state_doubles(const xh_rf2 *h) { return h ? 3 * h->nF : 0; }
// the Fourier volume and its weights [F (2 nF) | W (nF)] to / from a device buffer of the caller (--prepare_fsc keeps the halves)
int xh_rf2_state_export(xh_rf2 *h, double *d_dst)
{
    XH_CHECK(h && d_dst, XH_ERR_ARG, "xh_rf2_state_export: bad argument");
    XH_HIP(hipSetDevice(h->ctx->device));
    XH_HIP(hipMemcpyAsync(d_dst, h->d_F.p, sizeof(xh_cd) * h->nF, hipMemcpyDeviceToDevice, h->ctx->stream));
    XH_HIP(hipMemcpyAsync(d_dst + 2 * h->nF, h->d_W.p, sizeof(double) * h->nF, hipMemcpyDeviceToDevice, h->ctx->stream));
    return XH_OK;
}
int xh_rf2_state_import(xh_rf2 *h, const double *d_src, int32_t add)
{
    XH_CHECK(h && d_src, XH_ERR_ARG, "xh_rf2_state_import: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    if (!add) {
        XH_HIP(hipMemcpyAsync(h->d_F.p, d_src, sizeof(xh_cd) * h->nF, hipMemcpyDeviceToDevice, ctx->stream));
        XH_HIP(hipMemcpyAsync(h->d_W.p, d_src + 2 * h->nF, sizeof(double) * h->nF, hipMemcpyDeviceToDevice, ctx->stream));
        return XH_OK;
    }
    hipLaunchKernelGGL(k_rf_add_d, dim3((unsigned)((2 * h->nF + 255) / 256)), dim3(256), 0, ctx->stream, (double *)h->d_F.p, d_src, 2 * h->nF);
    hipLaunchKernelGGL(k_rf_add_d, dim3((unsigned)((h->nF + 255) / 256)), dim3(256), 0, ctx->stream, (double *)h->d_W.p, d_src + 2 * h->nF, h->nF);
    XH_LAUNCH_CHECK();
    return XH_OK;
}

// finishComputations (RF:1103-1178): enforceHermitianSymmetry, PROCESS_WEIGHTS, inverse transform, window, corrections.
// The Fourier volume and the weights stay as they are (the program keeps them for --prepare_fsc).
int xh_rf2_finish(xh_rf2 *h, double *h_volume)
{
    XH_CHECK(h && h_volume, XH_ERR_ARG, "xh_rf2_finish: bad argument");
    xh_ctx *ctx = h->ctx;
    XH_HIP(hipSetDevice(ctx->device));
    const int V = h->V, D = h->D;
    int half = V / 2; if (V % 2 == 0) half--;
    XhBuf spec, vol;
    int rc = xh_buf_alloc(ctx, spec, sizeof(xh_cd) * h->nF);
    if (rc == XH_OK) rc = xh_buf_alloc(ctx, vol, sizeof(double) * (size_t)D * D * D);
    if (rc == XH_OK) {
        // on a copy: the symmetrised coefficients are what the reference transforms, and it transforms in place
        hipLaunchKernelGGL(k_rf2_plane_symmetry, dim3((unsigned)((V * half + half + 255) / 256)), dim3(256), 0, ctx->stream, (xh_cd *)h->d_F.p, (double *)h->d_W.p, V, 1);
        const double corr2D_3D = std::pow(h->p.padding_proj, 2.) / (D * std::pow(h->p.padding_vol, 3.));
        hipLaunchKernelGGL(k_rf2_process_weights, dim3((unsigned)((h->nF + 255) / 256)), dim3(256), 0, ctx->stream, (const xh_cd *)h->d_F.p, (const double *)h->d_W.p,
                           (xh_cd *)spec.p, h->nF, corr2D_3D, h->niter);
        if (hipGetLastError() != hipSuccess) { xh_set_error("xh_rf2_finish: kernel launch failed"); rc = XH_ERR_HIP; }
    }
    if (rc == XH_OK)
        rc = finish_from_spectrum(ctx, h->planV.plan, (xh_cd *)spec.p, (double *)vol.p, (const double *)h->d_table.p + XH_BLOB_TABLE, D, h->iDeltaFourier,
                                  h->p.padding_proj, h->p.padding_vol, h->meanFactor2, h->niter != 0, h_volume);
    xh_buf_free(spec); xh_buf_free(vol);
    return rc;
}

}  // extern "C"
