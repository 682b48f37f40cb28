// xh_rf_grid.h -- the gridding kernel (included by xh_rf.hip).
//
// processProjection / processVoxelBlob (reconstruct_fourier_accel.cpp:627-700,710-763), output-stationary and
// LDS-fed.  The previous form of this kernel let every lane gather its own 4x4 footprint from the packed projection
// records in global memory: 16 loads of 16 B per (voxel, projection) pair, ~26 cache lines per wave-instruction, and
// the vector L1 (one tag lookup per clock and CU, 64 B/clk) was the bound -- 15.8 G lookups per 4096-projection
// launch = 32 ms.  Here the taps come out of LDS (256 B/clk and CU):
//
//   unit        a wave owns an 8 x 8 x 4 (x, y, z) block of the temp volume from start to finish and keeps its 256
//               (weight, re, im) sums in LDS
//   cull        lane <-> projection: the block's 16^3 super-tile list (k_rf_supercull) is tested against the block
//               (plane distance + half-plane extent, box bounds); the survivors' indices go to an LDS list
//   visit       per surviving projection: (1) SPARSE pass, lane <-> (x, y) column, four z: the reference's own tests
//               (slab |iz| <= r, sphere, AABB rows, pixels within reach) decide which voxels the projection touches;
//               their 8-bit ids are compacted into an LDS queue; (2) the 16 x 16 pixel patch of the projection that
//               covers the block's footprint is copied global -> LDS by four LDS-DMA instructions (no registers, one
//               visit ahead: the copy of visit k+1 flies while visit k is worked on); (3) DENSE pass, lane <-> queued
//               voxel, 64 at a time and every lane busy: row-visit test (getX, RFA:479-490), 4x4 Kaiser-Bessel taps
//               (table and pixels both from LDS), sums added to the unit's LDS accumulators -- a voxel appears at most
//               once per visit, so plain read-add-write, no atomics
//   write-back  one plain read-modify-write of the unit's voxels
//
// Records are packed per projection as float4 (re*ctf*mod*w, im*ctf*mod*w, mod*w, 0) with a 4-pixel zero frame, so a
// tap is three FMAs and a footprint never needs clamping (pixels outside the image carry weight 0).  The decisions
// (which voxels, which taps, which table entry) use the reference's float/double expressions unchanged; only the
// products inside a tap are re-associated, so sums agree with the reference to float rounding (tests: 2e-6).
#ifndef XH_RF_GRID_H
#define XH_RF_GRID_H

#ifndef XG_ACC_ATOMIC
#define XG_ACC_ATOMIC 0         // 1: the unit's sums updated by LDS additions without a return (A/B: tools/build_variant.sh)
#endif
#ifndef XG_TAPMASK
#define XG_TAPMASK 0            // 1: taps beyond the blob switched off with an execution mask instead of a select (A/B: tools/build_variant.sh)
#endif
#ifndef XG_PAD
#define XG_PAD 6                // zero cells around a packed record: a 6 x 6 footprint (blob radius < 3) starts at ceil(-2 r) >= -5
#endif

#include "xh_rf_cell.h"

// Compile-time shape of a kernel instance. A unit is 8 x 8 x ZD voxels (ZD = 4 or 8); NW waves share a CU.
//   PN   pixels per patch row / column a footprint can reach: the image extent of a unit is at most its diagonal
//        (10.4 / 12.2 pixels), so first footprint pixels lie 0..11 / 0..13 behind the patch origin, and a footprint is W wide
//   PW   patch row stride in pixels: > PN and not a multiple of 16 (64 banks of 4 bytes): consecutive rows are skewed
//   NDMA LDS-DMA instructions per patch (64 pixels each)
//   KCAP surviving projections listed per cull phase, three words each (index + flag, patch cell, patch origin)
// With 14 or more waves the LDS has no room for skew columns or for the slots behind the last row: the patch is PN x PN pixels
// exactly (an odd row stride: rows are skewed anyway), the last copy instruction runs with the lanes beyond the patch switched
// off, and the cull phase lists 32 projections at a time.
template <int W, int ZD, int NW> struct XgCfg {
    static constexpr int PN = (ZD == 8 ? 14 : 12) + W - 1;
    static constexpr bool TIGHT = NW >= 14;
#ifdef XG_PW_OVERRIDE
    static constexpr int PW = (W == 4 && ZD == 4 && NW == 12) ? XG_PW_OVERRIDE : (TIGHT ? PN : (PN <= 17 ? 18 : 20));      // A/B builds (tools/build_variant.sh)
#else
    // (15 rows of 21 pixels fill the five copy instructions of the 4 x 4 footprint's patch; a record is four banks wide, so the
    // sixteen lanes a ds_read_b128 serves per cycle collide when their (row, column) differ by a multiple of sixteen in
    // 21 row + column -- (1, -5), (3, 1), (2, 6) ... -- where a stride of 18 had (1, -2): measured 3 % of the kernel)
    static constexpr int PW = TIGHT ? PN : (PN == 15 ? 21 : (PN <= 17 ? 18 : 20));
#endif
    static constexpr int NDMA = (PN * PW + 63) / 64;
    static constexpr int PATCH_BYTES = TIGHT ? PN * PW * 16 : NDMA * 1024;
    static constexpr int KCAP = TIGHT ? 32 : 64;
    // The 4 x 4 footprint in one go, two rows at a time (registers: fourteen waves and more) or -- the product's twelve waves -- row by
    // row with the records and table entries of the next row requested before this row's multiply-adds (PIPE: two register sets; 133
    // registers instead of 167 and 1.4 % off the launch, round 6: profiles/experiments/r06_ab_grid_pipeline.txt)
#ifdef XG_HALVES
    static constexpr int HALVES = XG_HALVES;               // A/B builds (tools/build_variant.sh)
#else
    static constexpr int HALVES = (W == 4 && NW == 12) ? 4 : (TIGHT ? 2 : 1);
#endif
#ifdef XG_PIPE
    static constexpr bool PIPE = XG_PIPE != 0 && HALVES >= 2;
#else
    static constexpr bool PIPE = HALVES == 4;
#endif
    static constexpr int NVOX = 64 * ZD;
    static constexpr int LDS_PATCH = 0;                                  // [NW][PATCH_BYTES]; first, so that LDS-DMA bases stay below 64 KB
    static constexpr int LDS_BLOB = NW * PATCH_BYTES;                    // float[XH_BLOB_TABLE + 4]; entry XH_BLOB_TABLE is 0
    static constexpr int LDS_ACC = LDS_BLOB + 4 * (XH_BLOB_TABLE + 4);   // [NW][3][NVOX] float
    static constexpr int LDS_KEPT = LDS_ACC + NW * 3 * NVOX * 4;         // [NW][3][KCAP] int
    static constexpr int LDS_QUEUE = LDS_KEPT + NW * 3 * KCAP * 4;       // [NW][NVOX] unsigned short
    static constexpr int LDS_RING = LDS_QUEUE + NW * NVOX * 2;           // tile ring: int[8] tiles, int[8] ready, ticket, hop
    static constexpr int LDS_TOTAL = LDS_RING + 4 * 32;
    static_assert(NW * PATCH_BYTES <= 65536, "LDS-DMA bases must stay below 64 KB");
    static_assert(LDS_TOTAL <= 163840, "LDS budget of a CU");
};

// One traverse space (projection x symmetry placement) as the kernel reads it, through scalar loads:
//   r0, r1, r2  rows of the inverse transform (RFA:643-647); .w: image index, minY | maxY << 16, minZ | maxZ << 16 (AABB rows)
//   h0, h1, h2  the row-visit test (getX, RFA:479-490) as affine forms of (y, z), see xg_hit: (Uy, Uz, U0, dU), (Ty, Tz, T0, dT),
//               (band of u, band of t, image extent of a unit in x + r, in y + r)
//   da, db, dc  getX operands for the exact evaluation (d_hit, d_getX): (u.y, u.z, v.y, v.z), (p0.y, p0.z, p4.y, p4.z), (u.x, v.x, p0.x, 0)
struct XgRec { float4 r0, r1, r2, h0, h1, h2, da, db, dc; };

// the patch copy: NDMA LDS-DMA instructions, every lane's 16 bytes at base + off[i] land at ldsBase + 1024 i + 16 lane
// (base, ldsBase wave-uniform). M0 carries the LDS address. M0 is a reserved register of the AMDGPU back end: it never holds a
// value across instructions (the compiler writes it right before every use it generates), and naming it as a clobber is
// rejected as undefined behaviour (-Winline-asm), so the clobber list names memory and SCC only.
template <int NDMA, int SLOTS>
__device__ __forceinline__ void xg_dma_patch(const void *base, const unsigned (&off)[NDMA], unsigned ldsBase, int lane)
{
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        // (SLOTS < 64 NDMA: the lanes of the last instruction that would write behind the patch stay out of it)
        if (64 * (i + 1) <= SLOTS || 64 * i + lane < SLOTS)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off[i]), "s"(base), "s"(ldsBase + 1024u * i) : "memory");
    }
}
__device__ __forceinline__ void xg_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
typedef float xg_v2f __attribute__((ext_vector_type(2)));
typedef float xg_v4f __attribute__((ext_vector_type(4)));

// float4 (re*ctf*mod*w, im*ctf*mod*w, mod*w, 0) per pixel, XG_PAD zero cells on every side. blockIdx.y: image;
// a block writes XG_PACK_CELLS consecutive cells of the padded record (a kilobyte per wave and store instruction).
#define XG_PACK_CELLS 1024
__global__ void __launch_bounds__(256)
k_rf_pack_grid(const xh_cf *__restrict__ ffts, const float *__restrict__ ctfs, const float *__restrict__ mods,
               const float *__restrict__ weights, XgCell *__restrict__ pk, int n, int sizeX, int sizeY, int skipR2)
{
    const unsigned SX = sizeX + 2 * XG_PAD, SY = sizeY + 2 * XG_PAD, cells = SX * SY;
    const unsigned img = blockIdx.y;
    const float w = weights ? weights[img] : 1.f;
    const size_t src = (size_t)img * sizeX * sizeY;
    XgCell *dst = pk + (size_t)img * cells;
    // all loads of the thread's four cells first (bytes in flight are what a streaming kernel runs on), then the stores
    constexpr int NC = XG_PACK_CELLS / 256;
    xh_cf f[NC];
    float cm[NC], cc[NC];
    bool in[NC], far[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const unsigned c = blockIdx.x * XG_PACK_CELLS + k * 256 + threadIdx.x;
        const unsigned row = c / SX;
        const int x = (int)(c - row * SX) - XG_PAD, y = (int)row - XG_PAD;
        in[k] = x >= 0 && x < sizeX && y >= 0 && y < sizeY;     // false beyond the last cell too (y >= sizeY)
        far[k] = x * x + (y - sizeY / 2) * (y - sizeY / 2) > skipR2;      // beyond every tap's reach (k_rf_pack_grid_ctf): left alone
        const size_t o = src + (size_t)(in[k] ? y * sizeX + x : 0);
        f[k] = xh_cf{0.f, 0.f}; cm[k] = 1.f; cc[k] = 1.f;
        if (in[k]) {
            f[k] = ffts[o];
            if (ctfs) { cm[k] = mods[o]; cc[k] = ctfs[o]; }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const unsigned c = blockIdx.x * XG_PACK_CELLS + k * 256 + threadIdx.x;
        if (c >= cells) break;
        if (far[k]) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[k]) {
            if (ctfs) { const float mw = cm[k] * w; v = make_float4(f[k].x * mw * cc[k], f[k].y * mw * cc[k], mw, 0.f); }
            else v = make_float4(f[k].x * w, f[k].y * w, w, 0.f);
        }
        xg_put(dst + c, v.x, v.y, v.z);
    }
}

// The same records with the CTF planes never written: the CTF factor and the modulator of a pixel (preloadCTF, RFA:548-592)
// are evaluated here by the function k_rf_ctf uses, so the records equal those packed from xh_rf_ctf_arrays' planes bit for
// bit, and 2 x 8 bytes per pixel of plane traffic (written, then read back) disappear. A thread owns the pixel (x, dc + k)
// and its mirror row (x, dc - k): a non-astigmatic CTF is the same on both, evaluated once (see k_rf_ctf).
__global__ void __launch_bounds__(256)
k_rf_pack_grid_ctf(const xh_cf *__restrict__ ffts, const XhCtfDev *__restrict__ cp, const float *__restrict__ weights,
                   XgCell *__restrict__ pk, int sizeX, int sizeY, int P, double iTs, double minCTF, int phaseFlipped, int skipR2)
{
    const int SX = sizeX + 2 * XG_PAD, SY = sizeY + 2 * XG_PAD;
    const int dc = P / 2;
    const int K = max(sizeY + XG_PAD - dc, dc + XG_PAD + 1);       // rows dc + k up to the lower frame, dc - k up to the upper
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (unsigned)K * SX) return;
    const int k = c / (unsigned)SX, x = (int)(c - k * SX) - XG_PAD;
    // Cells no tap can reach stay as they are (zero since the buffer was allocated, or an earlier batch's finite values): the
    // traversal keeps voxels within sizeX + r of the origin (RFA:640), their image coordinates lie within that radius too and a tap
    // within r of them, so a pixel further than sizeX + 2 r from the origin only ever meets the table's zero entry.  A quarter of the
    // record, of its CTF evaluations and of its bytes (round 6).  (The origin of the record's rows is sizeY / 2 -- where the gridding
    // kernel puts it --, which is the CTF's centre row dc = P / 2 only when nothing was cropped.)
    const int g1 = dc + k - sizeY / 2, g2 = dc - k - sizeY / 2;
    const bool far1 = x * x + g1 * g1 > skipR2, far2 = x * x + g2 * g2 > skipR2;
    if (far1 && far2) return;
    const unsigned img = blockIdx.y;
    const XhCtfDev par = cp[img];
    const float w = weights ? weights[img] : 1.f;
    const xh_cf *src = ffts + (size_t)img * sizeX * sizeY;
    XgCell *dst = pk + (size_t)img * SX * SY;
    const int y1 = dc + k, y2 = dc - k;
    const bool row1 = y1 < sizeY + XG_PAD && !far1, row2 = k > 0 && y2 >= -XG_PAD && y2 < sizeY + XG_PAD && !far2;
    const bool inx = x >= 0 && x < sizeX;
    const bool in1 = inx && y1 < sizeY && !far1, in2 = inx && k > 0 && y2 >= 0 && y2 < sizeY && !far2;
    xh_cf f1 = xh_cf{0.f, 0.f}, f2 = f1;
    if (in1) f1 = src[(size_t)y1 * sizeX + x];
    if (in2) f2 = src[(size_t)y2 * sizeX + x];
    float cv = 0.f, mv_ = 0.f;
    float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
    if (in1) {
        d_ctf_eval(par, x, y1, P, iTs, minCTF, phaseFlipped, cv, mv_);
        const float mw = mv_ * w;
        v1 = make_float4(f1.x * mw * cv, f1.y * mw * cv, mw, 0.f);
    }
    if (in2) {
        // rows dc+k and dc-k have opposite freqY only for even P (freqY = (y - P/2)/P)
        if (!(in1 && par.defocus_deviation == 0 && (P & 1) == 0)) d_ctf_eval(par, x, y2, P, iTs, minCTF, phaseFlipped, cv, mv_);
        const float mw = mv_ * w;
        v2 = make_float4(f2.x * mw * cv, f2.y * mw * cv, mw, 0.f);
    }
    if (row1) xg_put(dst + (size_t)(y1 + XG_PAD) * SX + (x + XG_PAD), v1.x, v1.y, v1.z);
    if (row2) xg_put(dst + (size_t)(y2 + XG_PAD) * SX + (x + XG_PAD), v2.x, v2.y, v2.z);
}

// Host side of the row-visit test. getX (RFA:479-490) intersects the voxel row (y, z) with a face of the slab:
//   u = ((z - z0) a1 + (y0 - y) a2) / (a1 b2 - b1 a2),  t = (y - y0 - u b1) / a1,  row visited iff 0 < u, t < 1 on either face.
// Both are affine in (y, z); the device evaluates the affine forms and only where a value comes within a band of 0 or 1
// -- the band covers the rounding of both evaluations -- does the lane repeat the reference's own float arithmetic.
__host__ __device__ static inline void xg_fill_hit(XgRec &G, const XhSpace &S, double blobRadius)
{
    const double a1 = S.u[1], a2 = S.u[2], b1 = S.v[1], b2 = S.v[2];
    const double den = a1 * b2 - b1 * a2;
    const double y01 = S.p0[1], z01 = S.p0[2], y02 = S.p4[1], z02 = S.p4[2];
    const float huge = 1e30f;
    G.da = make_float4(S.u[1], S.u[2], S.v[1], S.v[2]);
    G.db = make_float4(S.p0[1], S.p0[2], S.p4[1], S.p4[2]);
    G.dc = make_float4(S.u[0], S.v[0], S.p0[0], 0.f);
    if (!(std::fabs(den) > 1e-20) || !(std::fabs(a1) > 1e-20) || !std::isfinite(den)) {
        G.h0 = make_float4(0.f, 0.f, 0.5f, 0.f);
        G.h1 = make_float4(0.f, 0.f, 0.5f, 0.f);
        G.h2.x = huge; G.h2.y = huge;            // every lane takes the exact path
        return;
    }
    const double Uz = a1 / den, Uy = -a2 / den;
    const double U01 = (-z01 * a1 + y01 * a2) / den, U02 = (-z02 * a1 + y02 * a2) / den;
    const double Ty = (1.0 - Uy * b1) / a1, Tz = -Uz * b1 / a1;
    const double T01 = (-y01 - U01 * b1) / a1, T02 = (-y02 - U02 * b1) / a1;
    G.h0 = make_float4((float)Uy, (float)Uz, (float)U01, (float)(U02 - U01));
    G.h1 = make_float4((float)Ty, (float)Tz, (float)T01, (float)(T02 - T01));
    // bands: a dozen roundings of 2^-24 on the largest intermediate of either evaluation (coordinates stay below 1100)
    const double eps = 16.0 / 16777216.0, L = 1100.0;
    const double mu = L * (std::fabs(Uy) + std::fabs(Uz)) + std::fabs(U01) + std::fabs(U02) + 1.0;
    const double numMax = L * (std::fabs(a1) + std::fabs(a2)) + std::fabs(z01 * a1) + std::fabs(y01 * a2) + std::fabs(z02 * a1) + std::fabs(y02 * a2);
    const double bu = eps * (mu + numMax / std::fabs(den));
    const double mt = L * (std::fabs(Ty) + std::fabs(Tz)) + std::fabs(T01) + std::fabs(T02) + 1.0;
    const double bt = eps * (mt + (L + std::fabs(y01) + std::fabs(y02) + (mu + 1.0) * std::fabs(b1)) / std::fabs(a1)) + bu * std::fabs(b1 / a1);
    G.h2.x = (float)std::min(bu, 1e30); G.h2.y = (float)std::min(bt, 1e30);
}

// The record of a traverse space and its two cull vectors (plane normal, image x axis; .w: their 1-norms, the support
// function of a unit cube).
__host__ __device__ static inline void xg_fill_rec(XgRec &G, float4 &nv, float4 &xv, const XhSpace &S, double blobRadius)
{
    nv = make_float4(S.tInv[6], S.tInv[7], S.tInv[8], std::fabs(S.tInv[6]) + std::fabs(S.tInv[7]) + std::fabs(S.tInv[8]));
    xv = make_float4(S.tInv[0], S.tInv[1], S.tInv[2], std::fabs(S.tInv[0]) + std::fabs(S.tInv[1]) + std::fabs(S.tInv[2]));
    G.r0 = make_float4(S.tInv[0], S.tInv[1], S.tInv[2], __builtin_bit_cast(float, S.img));
    G.r1 = make_float4(S.tInv[3], S.tInv[4], S.tInv[5], __builtin_bit_cast(float, S.minY | (S.maxY << 16)));
    G.r2 = make_float4(S.tInv[6], S.tInv[7], S.tInv[8], __builtin_bit_cast(float, S.minZ | (S.maxZ << 16)));
    xg_fill_hit(G, S, blobRadius);
    // image extent of a unit (half extents 3.5, 3.5, 1.5 voxels) + blob radius: where its first footprint pixel lies
    G.h2.z = 3.5f * (std::fabs(S.tInv[0]) + std::fabs(S.tInv[1])) + 1.5f * std::fabs(S.tInv[2]) + 0.01f + (float)blobRadius;
    G.h2.w = 3.5f * (std::fabs(S.tInv[3]) + std::fabs(S.tInv[4])) + 1.5f * std::fabs(S.tInv[5]) + 0.01f + (float)blobRadius;
}

// Launch order of the traverse spaces: those that share a plane next to each other.  Particles assigned to one gallery direction
// carry the same (rot, tilt) and differ in their in-plane angle only; their slabs are one slab, and the gridding kernel, which
// visits the spaces of a list in launch order, takes the voxel queue of the previous visit when the plane is the same (k_rf_grid,
// `sameQueue`).  One workgroup; chunks of XG_ORD spaces are ordered independently by a bitonic sort of (key, index) pairs in LDS --
// key: a hash of the bits of rot and tilt and the symmetry index, so equal directions meet whatever else the order is.  The order
// only permutes the launch's float additions (RFA:300-388 grids in metadata order).  pos[idx]: the space's place in the launch.
#define XG_ORD 4096
__global__ void __launch_bounds__(1024) k_rf_space_order(const double *__restrict__ angles, int n, int nsym, int *__restrict__ pos)
{
    __shared__ unsigned long long sk[XG_ORD];
    __shared__ int si[XG_ORD];
    const int ns = n * nsym;
    for (int c0 = 0; c0 < ns; c0 += XG_ORD) {
        for (int t = threadIdx.x; t < XG_ORD; t += 1024) {
            const int idx = c0 + t;
            unsigned long long k = ~0ull;
            if (idx < ns) {
                const int i = idx / nsym, sy = idx - i * nsym;
                const unsigned long long a = (unsigned long long)__double_as_longlong(angles[3 * i]), b = (unsigned long long)__double_as_longlong(angles[3 * i + 1]);
                k = (a * 0x9E3779B97F4A7C15ull) ^ ((b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full) ^ ((unsigned long long)sy * 0x165667B19E3779F9ull);
                k ^= k >> 29;
                k >>= 1;                                    // below the padding key
            }
            sk[t] = k; si[t] = idx;
        }
        __syncthreads();
        for (int size = 2; size <= XG_ORD; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = threadIdx.x; t < XG_ORD / 2; t += 1024) {
                    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const unsigned long long ka = sk[lo], kb = sk[hi];
                    const int ia = si[lo], ib = si[hi];
                    const bool gt = ka > kb || (ka == kb && ia > ib);
                    if (gt == up) { sk[lo] = kb; sk[hi] = ka; si[lo] = ib; si[hi] = ia; }
                }
                __syncthreads();
            }
        for (int t = threadIdx.x; t < XG_ORD; t += 1024)
            if (si[t] < ns) pos[si[t]] = c0 + t;
        __syncthreads();
    }
}

// The same on the device, from Euler angles that never left it (xh_rf_insert_images_dev): one thread per (projection,
// symmetry matrix). A projection of weight 0 keeps its slot with NaN cull vectors, which no list admits (RFA:327-329).
__global__ void __launch_bounds__(64) k_rf_spaces(const double *__restrict__ angles, const float *__restrict__ weights,
                                                   const double *__restrict__ sym, int n, int nsym, int mv, double blobRadius,
                                                   int useFast, XgRec *__restrict__ recs, float4 *__restrict__ cullN,
                                                   float4 *__restrict__ cullX, const int *__restrict__ pos)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * nsym) return;
    const int i = idx / nsym, s = idx - i * nsym;
    const float w = weights ? weights[i] : 1.0f;
    double A[9], T[9], R[9];
    h_euler(angles[3 * i], angles[3 * i + 1], angles[3 * i + 2], A);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[c * 3 + r];   // localAInv = A^T (RFA:348-350)
    for (int k = 0; k < 9; ++k) R[k] = sym ? sym[9 * s + k] : (k % 4 == 0 ? 1.0 : 0.0);
    XhSpace S;
    h_place(S, R, T, mv, blobRadius, useFast != 0, w, i);
    XgRec G;
    float4 nv, xv;
    xg_fill_rec(G, nv, xv, S, blobRadius);
    if (weights && w == 0.f) {
        const float q = __builtin_nanf("");
        nv = make_float4(q, q, q, q);
        xv = nv;
    }
    const int o = pos ? pos[idx] : idx;            // launch order (k_rf_space_order)
    recs[o] = G;
    cullN[o] = nv;
    cullX[o] = xv;
}

// W: footprint width (4 for a blob radius below 2, 6 below 3). FAST: processVoxel (RFA:595-625), nearest pixel, one voxel per row.
// ZD: depth of a unit in z (4 or 8); NW: waves per workgroup (= per CU: the LDS budget admits one workgroup).
// ABL: ablation switch for profiling builds (tools/ab_grid.sh); 0 in the product: 1 no wait for the patch copy, 2 no dense pass,
// 3 no patch copy, 5 histogram of items per visit into tempV, 6 = 2 + 3, 7 = 6 without the sparse pass, 8 every patch copied
// from the first megabyte of the records (the copy instructions without their HBM traffic), 9 every tap reads the patch's first
// record, 10 every tap reads the table's first entry, 11 = 9 + 10 (the LDS reads without their bank conflicts).
template <int W, bool FAST, int ZD, int NW, int ABL>
__global__ void __launch_bounds__(64 * NW, (NW + 3) / 4)
k_rf_grid(const XgRec *__restrict__ recs, const XgCell *__restrict__ pk, const float *__restrict__ blobTable,
          float *__restrict__ tempV, float *__restrict__ tempW, int mv, float iDeltaSqrt, double blobRadius,
          const unsigned *__restrict__ tileList, const int *__restrict__ classOff, int *__restrict__ counter,
          const int *__restrict__ superList, const int *__restrict__ superCount, int superDim, int superCap,
          const float4 *__restrict__ superN, const float4 *__restrict__ superX, float4 reach, int tileBudget)
{
    using C = XgCfg<W, ZD, NW>;
    constexpr int PW = C::PW, PN = C::PN, NDMA = C::NDMA, NVOX = C::NVOX, KCAP = C::KCAP;
    constexpr float HZ = 0.5f * ZD - 0.5f;                   // half extent of a unit's voxel centres in z
    constexpr float HSPH = ZD == 8 ? 6.1f : 5.2f;            // ... never wider than the sphere around them
    __shared__ __align__(16) unsigned char lds[C::LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sBlob = reinterpret_cast<float *>(lds + C::LDS_BLOB);
    float *sAcc = reinterpret_cast<float *>(lds + C::LDS_ACC) + wv * 3 * NVOX;
    int *sKept = reinterpret_cast<int *>(lds + C::LDS_KEPT) + wv * 3 * KCAP;      // [0]: index | allHit << 31, [KCAP]: cell, [2 KCAP]: ox | oy << 16
    unsigned short *sQueue = reinterpret_cast<unsigned short *>(lds + C::LDS_QUEUE) + wv * NVOX;
    int *sTile = reinterpret_cast<int *>(lds + C::LDS_RING), *sReady = sTile + 8, *sTicket = sTile + 16, *sHop = sTile + 17;
    const unsigned char *sPatch = lds + C::LDS_PATCH + wv * C::PATCH_BYTES;
    const unsigned patchBase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)sPatch;
    const int sizeX = mv / 2, sizeY = mv, dim = mv + 1;
    const int SX = sizeX + 2 * XG_PAD, SY = sizeY + 2 * XG_PAD;

    for (int i = tid; i < XH_BLOB_TABLE + 4; i += 64 * NW) sBlob[i] = i < XH_BLOB_TABLE ? blobTable[i] : 0.f;
    const float fr = FAST ? 0.f : (float)blobRadius;
    const float frCull = FAST ? 0.5f : fr;       // --fast: the voxel nearest to the crossing lies up to half a voxel off the plane
    const float maxDistanceSqr = (sizeX + (FAST ? 0.0 : blobRadius)) * (sizeX + (FAST ? 0.0 : blobRadius));
    const float radiusSqr = blobRadius * blobRadius;
    const int lx = lane & 7, ly = lane >> 3;
    const float fmvh = (float)(mv / 2);
    const xg_v2f idel2 = {iDeltaSqrt, iDeltaSqrt}, half2 = {0.5f, 0.5f};
    const float limf = (float)XH_BLOB_TABLE;
    const unsigned long long below = (1ull << lane) - 1ull;
    // which pixel of a patch lane l of DMA instruction i fetches: LDS slot s = 64 i + l holds pixel (s / PW, s % PW)
    // (footprints reach rows and columns 0..PN-1 of a patch: the skew columns and the slots behind the last row re-read the last
    // pixel of their row / the last row, which costs no further cache line)
    unsigned dOff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) { const int slot = 64 * i + lane, row = slot / PW, col = slot - row * PW; dOff[i] = (unsigned)(min(row, PN - 1) * SX + min(col, PN - 1)) * (unsigned)sizeof(XgCell); }

    // ---- work distribution: a tile ring (tiles = 2 x 2 x 2 units, eight XCD classes dealt into NSUB interleaved streams, one
    // global grab per tile, eight tickets per tile drawn by the waves of the workgroup)
    constexpr int NSUB = 8, RS = 8;
    const int home = (blockIdx.x & 7) * NSUB + ((blockIdx.x >> 3) & (NSUB - 1));
    auto streamTiles = [&](int st) {
        const int c = st / NSUB, j = st % NSUB, nt = classOff[c + 1] - classOff[c];
        return nt > j ? (nt - j + NSUB - 1) / NSUB : 0;
    };
    auto produce = [&](int q) {
        while (q > 0 && atomicAdd(&sReady[(q - 1) % RS], 0) != q) __builtin_amdgcn_s_sleep(1);
        int tile = -1;
        // tileBudget > 0: a workgroup retires after that many tiles and the launch has more workgroups than CUs (bounded residency: the
        // kernels of another stream get CUs every few milliseconds instead of after the whole launch)
        if (!(tileBudget > 0 && q >= tileBudget))
        for (;;) {
            const int hop = atomicAdd(sHop, 0);
            if (hop >= 8 * NSUB) break;
            const int st = (home + hop) % (8 * NSUB);
            const int k = atomicAdd(&counter[st], 1);
            if (k < streamTiles(st)) { tile = (int)tileList[classOff[st / NSUB] + st % NSUB + k * NSUB]; break; }
            atomicMax(sHop, hop + 1);
        }
        sTile[q % RS] = tile;
        __threadfence_block();
        atomicExch(&sReady[q % RS], q + 1);
    };
    if (tid == 0) {
        *sTicket = 0; *sHop = 0;
        for (int i = 0; i < RS; ++i) sReady[i] = 0;
        produce(0);
        produce(1);
    }
    __syncthreads();

    for (;;) {
        int t = 0, tileP = 0;
        if (lane == 0) {
            t = atomicAdd(sTicket, 1);
            const int q = t >> 3;
            if ((t & 7) == 0) produce(q + 2);
            while (atomicAdd(&sReady[q % RS], 0) != q + 1) __builtin_amdgcn_s_sleep(1);
            tileP = atomicAdd(&sTile[q % RS], 0);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        tileP = __builtin_amdgcn_readfirstlane(tileP);
        if (tileP < 0) break;
        const unsigned packed = (unsigned)tileP;
        const int sub = t & 7;
        const int tx = packed & 0x3ff, ty = (packed >> 10) & 0x3ff, tz = (packed >> 20) & 0x3ff;
        const int x0 = tx * 16 + (sub & 1) * 8, y0 = ty * 16 + ((sub >> 1) & 1) * 8, z0 = tz * (2 * ZD) + (sub >> 2) * ZD;
        const int x = x0 + lx, y = y0 + ly;
        const float px = x - mv / 2, py = y - mv / 2;
        int sph = 0;      // bit zi: voxel (x, y, z0 + zi) lies in the volume and inside the sphere the reference keeps (RFA:640)
#pragma unroll
        for (int zi = 0; zi < ZD; ++zi) {
            const int z = z0 + zi;
            const float pz = z - mv / 2;
            const bool in = (x <= mv) && (y <= mv) && (z <= mv) && !((px * px + py * py + pz * pz) > maxDistanceSqr);
            sph |= (int)in << zi;
        }
        if (!__ballot(sph != 0)) continue;
#pragma unroll
        for (int i = 0; i < 3 * ZD; ++i) sAcc[i * 64 + lane] = 0.f;
        const float fx0 = (float)(x0 - mv / 2), fy0 = (float)(y0 - mv / 2), fz0 = (float)(z0 - mv / 2);
        const float ucx = x0 + 3.5f - mv / 2, ucy = y0 + 3.5f - mv / 2, ucz = z0 + HZ - mv / 2;
        const int sup = ((tz * (2 * ZD) / 16) * superDim + ty) * superDim + tx;
        const int nlist = superCount[sup];
        const int *lst = superList + (size_t)sup * superCap;
        const float4 *lstN = superN + (size_t)sup * superCap, *lstX = superX + (size_t)sup * superCap;

        int pos = 0;
        int sNext = 0;
        float4 nNext = make_float4(0.f, 0.f, 0.f, 0.f), xNext = nNext;
        constexpr int CR = KCAP < 64 ? KCAP : 64;      // list entries per cull round: every one of them may survive
        if (lane < CR && lane < nlist) { sNext = lst[lane]; nNext = lstN[lane]; xNext = lstX[lane]; }
        while (pos < nlist) {
            // ---- cull phase: lane <-> list entry
            int nk = 0;
            while (pos < nlist && nk <= KCAP - CR) {
                const int sIdx = sNext;
                const float4 n = nNext, xv = xNext;
                const int e = pos + lane, en = e + CR;
                if (lane < CR && en < nlist) { sNext = lst[en]; nNext = lstN[en]; xNext = lstX[en]; }
                bool keep = false;
                if (lane < CR && e < nlist) {
                    const float dn = n.x * ucx + n.y * ucy + n.z * ucz;
                    const float dx = xv.x * ucx + xv.y * ucy + xv.z * ucz;
                    // support function of the box of voxel centres (half extents 3.5, 3.5, HZ), never wider than its sphere
                    const float hn = fminf(HSPH, 3.5f * (fabsf(n.x) + fabsf(n.y)) + HZ * fabsf(n.z) + 0.02f);
                    const float hx = fminf(HSPH, 3.5f * (fabsf(xv.x) + fabsf(xv.y)) + HZ * fabsf(xv.z) + 0.02f);
                    keep = (fabsf(dn) <= frCull + hn) && (dx >= -(frCull + hx)) && (dx <= sizeX + frCull + hx);
                }
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    // What a visit needs before it can start, found here for 64 projections at a time instead of once per visit
                    // on the scalar side: where the PN x PN patch that covers the unit's footprints begins (the first footprint
                    // pixel of the corner with the smallest image coordinates: centre - image extent of the unit - r; kept inside
                    // the padded record: footprints never leave that, so an origin moved inwards still covers them), and whether
                    // every row of the unit crosses a face of the slab (u and t of the unit's centre row keep their distance from
                    // 0 and 1 by more than they vary over the unit: 3.5 rows in y, HZ in z).
                    const float4 r0 = recs[sIdx].r0, r1 = recs[sIdx].r1, h0 = recs[sIdx].h0, h1 = recs[sIdx].h1, h2 = recs[sIdx].h2;
                    const float cix = r0.x * ucx + r0.y * ucy + r0.z * ucz;
                    const float ciy = r1.x * ucx + r1.y * ucy + r1.z * ucz + fmvh;
                    const float ex = 3.5f * (fabsf(r0.x) + fabsf(r0.y)) + HZ * fabsf(r0.z) + 0.01f + fr;
                    const float ey = 3.5f * (fabsf(r1.x) + fabsf(r1.y)) + HZ * fabsf(r1.z) + 0.01f + fr;
                    const int ox = min(max((int)ceilf(cix - ex), -XG_PAD), SX - XG_PAD - PN);
                    const int oy = min(max((int)ceilf(ciy - ey), -XG_PAD), SY - XG_PAD - PN);
                    // cell index of the patch origin in the packed records (the host keeps a launch below 2^31 cells)
                    const unsigned cell = ((unsigned)__float_as_int(r0.w) * (unsigned)SY + (unsigned)(oy + XG_PAD)) * (unsigned)SX + (unsigned)(ox + XG_PAD);
                    const float yc = (float)y0 + 3.5f, zc = (float)z0 + HZ;
                    const float u1 = h0.x * yc + h0.y * zc + h0.z, t1 = h1.x * yc + h1.y * zc + h1.z;
                    const float eu = 3.5f * fabsf(h0.x) + HZ * fabsf(h0.y) + h2.x + 1e-4f, et = 3.5f * fabsf(h1.x) + HZ * fabsf(h1.y) + h2.y + 1e-4f;
                    const bool a1 = (fabsf(u1 - 0.5f) + eu < 0.5f) && (fabsf(t1 - 0.5f) + et < 0.5f);
                    const bool a2 = (fabsf(u1 + h0.w - 0.5f) + eu < 0.5f) && (fabsf(t1 + h1.w - 0.5f) + et < 0.5f);
                    // ... and whether the sparse pass can skip its bounds: every voxel of the unit inside the rows the traversal visits
                    // (AABB, RFA:724-741) and every image coordinate within reach of the image (the unit's image extent without the blob)
                    const int yy = __float_as_int(r1.w), zz = __float_as_int(recs[sIdx].r2.w);
                    const bool inBox = y0 >= (yy & 0xffff) && y0 + 7 <= (yy >> 16) && z0 >= (zz & 0xffff) && z0 + ZD - 1 <= (zz >> 16);
                    const float ex0 = ex - fr, ey0 = ey - fr;
                    const bool inReach = (cix - ex0 >= reach.x) && (cix + ex0 <= reach.y) && (ciy - ey0 >= reach.z) && (ciy + ey0 <= reach.w);
                    const int at = nk + __popcll(bal & below);
#ifdef XG_NO_INTERIOR
                    sKept[at] = sIdx | ((a1 || a2) ? (int)0x80000000 : 0) | ((inBox && inReach && mv < 0) ? 0x40000000 : 0);      // A/B builds
#else
                    sKept[at] = sIdx | ((a1 || a2) ? (int)0x80000000 : 0) | ((inBox && inReach) ? 0x40000000 : 0);
#endif
                    sKept[KCAP + at] = (int)cell;
                    sKept[2 * KCAP + at] = (ox & 0xffff) | (oy << 16);
                }
                nk += __popcll(bal);
                pos += CR;
            }
            if (nk == 0) continue;
            __builtin_amdgcn_wave_barrier();

            // ---- visits. Records travel through scalar registers, one visit ahead (unconditional loads: a select would
            // make the compiler wait for them on the spot); the index of the record after that is read from LDS meanwhile
            int kw = __builtin_amdgcn_readfirstlane(sKept[0]);
            int kid = kw & 0x3fffffff;
            int kidN = __builtin_amdgcn_readfirstlane(sKept[min(1, nk - 1)]) & 0x3fffffff;
            float4 R0 = recs[kid].r0, R1 = recs[kid].r1, R2 = recs[kid].r2;
            // the queue of the previous visit serves this one too when both projections have the same plane (the traverse spaces of
            // a launch are ordered by direction: particles assigned to one gallery direction differ in their in-plane angle only)
            // and neither needs more than the slab test: which voxels lie within the blob radius of the plane is then the same set
            int qn = 0;
            float pNx = 0.f, pNy = 0.f, pNz = 0.f;
            bool prevInterior = false;
            for (int k = 0; k < nk; ++k) {
                const float4 N0 = recs[kidN].r0, N1 = recs[kidN].r1, N2 = recs[kidN].r2;
                const int kidNN = sKept[min(k + 2, nk - 1)], kwN = sKept[min(k + 1, nk - 1)];
                const unsigned cell = (unsigned)__builtin_amdgcn_readfirstlane(sKept[KCAP + k]);
                const int oxy = __builtin_amdgcn_readfirstlane(sKept[2 * KCAP + k]);
                const int ox = (int)(short)(oxy & 0xffff), oy = oxy >> 16;
                const bool allHit = kw < 0, interior = (kw & 0x40000000) != 0;
                // the row-visit test's forms are only read where some row of the unit may miss (15 % of the visits)
                float4 H0 = make_float4(0.f, 0.f, 0.f, 0.f), H1 = H0, H2 = H0;
                if (!allHit) { H0 = recs[kid].h0; H1 = recs[kid].h1; H2 = recs[kid].h2; }
                if constexpr (FAST) {
                    // processVoxel (RFA:595-625) over the traversal of a zero-thickness slab (RFA:743-761): every row (y, z)
                    // that crosses the image plane gives its voxel nearest to the crossing the nearest pixel. lane <-> row.
                    for (int rr = lane; rr < 8 * ZD; rr += 64) {
                        const int ry = rr & 7, rz = rr >> 3, vy = y0 + ry, vz = z0 + rz;
                        const int yy = __float_as_int(R1.w), zz = __float_as_int(R2.w);
                        const float4 da = recs[kid].da, db = recs[kid].db, dc = recs[kid].dc;
                        const float ua[3] = {dc.x, da.x, da.y}, va[3] = {dc.y, da.z, da.w}, p0[3] = {dc.z, db.x, db.y};
                        float hitX;
                        const bool in = !(vy < (yy & 0xffff) || vy > (yy >> 16) || vz < (zz & 0xffff) || vz > (zz >> 16)) && vy <= mv && vz <= mv;
                        if (in && d_getX(hitX, (float)vy, (float)vz, ua, va, p0)) {
                            const int vx = (int)(hitX + 0.5f);
                            const float qx = vx - mv / 2, qy = vy - mv / 2, qz = vz - mv / 2;
                            if (vx >= x0 && vx < x0 + 8 && vx <= mv && !(qx * qx + qy * qy + qz * qz > maxDistanceSqr)) {
                                const float ix = R0.x * qx + R0.y * qy + R0.z * qz;
                                const float iy = R1.x * qx + R1.y * qy + R1.z * qz;
                                int imgX = (int)(ix + 0.5f);
                                imgX = imgX > sizeX - 1 ? sizeX - 1 : imgX;
                                imgX = imgX < 0 ? 0 : imgX;
                                int imgY = (int)(iy + 0.5f + mv / 2);
                                imgY = imgY > sizeY - 1 ? sizeY - 1 : imgY;
                                imgY = imgY < 0 ? 0 : imgY;
                                const XgCell qc = pk[((size_t)__float_as_int(R0.w) * SY + (imgY + XG_PAD)) * SX + (imgX + XG_PAD)];
                                const float4 q = make_float4(qc.v[0], qc.v[1], qc.v[2], 0.f);
                                const int ai = rz * 64 + ry * 8 + (vx - x0);
                                sAcc[ai] += q.z;
                                sAcc[NVOX + ai] += q.x;
                                sAcc[2 * NVOX + ai] += q.y;
                            }
                        }
                    }
                    R0 = N0; R1 = N1; R2 = N2;
                    kid = kidN;
                    kw = __builtin_amdgcn_readfirstlane(kwN);
                    kidN = __builtin_amdgcn_readfirstlane(kidNN) & 0x3fffffff;
                    continue;
                }
                // ---- the patch copy (origin found in the cull phase). The patch buffer is free: the previous dense pass has
                // consumed its reads.
                if constexpr (ABL != 3 && ABL != 6 && ABL != 7) xg_dma_patch<NDMA, C::PATCH_BYTES / 16>(pk + (ABL == 8 ? (cell & 0xffffu) : cell), dOff, patchBase, lane);
                // ---- sparse pass (RFA:631-653 and the reach of the footprint), two z at a time
                const int yy = __float_as_int(R1.w), zz = __float_as_int(R2.w);
                const bool yok = !(y < (yy & 0xffff) || y > (yy >> 16));
                const float ax = R0.x * px + R0.y * py, ay = R1.x * px + R1.y * py, az = R2.x * px + R2.y * py;
                const bool sameQueue = interior && prevInterior && ((R2.x == pNx && R2.y == pNy && R2.z == pNz) || tileBudget == -1);
                pNx = R2.x; pNy = R2.y; pNz = R2.z; prevInterior = interior;
                if (!sameQueue) qn = 0;
                if constexpr (ABL != 7) {
                if (sameQueue) {
                } else if (interior) {
                    // the slab test alone (the sphere bit stays: the unit may straddle the sphere the reference keeps)
#pragma unroll
                    for (int zp = 0; zp < ZD / 2; ++zp) {
                        const xg_v2f pz2 = {fz0 + (float)(2 * zp), fz0 + (float)(2 * zp + 1)};
                        const xg_v2f iz2 = (xg_v2f){az, az} + (xg_v2f){R2.z, R2.z} * pz2;
                        const xg_v2f zs2 = iz2 * iz2;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int zi = 2 * zp + h;
                            const bool pass = ((sph >> zi) & 1) && !((h ? zs2.y : zs2.x) > radiusSqr);
                            const unsigned long long pb = __ballot(pass);
                            if (pass) sQueue[qn + __popcll(pb & below)] = (unsigned short)(lane + 64 * zi);
                            qn += __popcll(pb);
                        }
                    }
                } else
#pragma unroll
                for (int zp = 0; zp < ZD / 2; ++zp) {
                    const xg_v2f pz2 = {fz0 + (float)(2 * zp), fz0 + (float)(2 * zp + 1)};
                    const xg_v2f ix2 = (xg_v2f){ax, ax} + (xg_v2f){R0.z, R0.z} * pz2;
                    xg_v2f iy2 = (xg_v2f){ay, ay} + (xg_v2f){R1.z, R1.z} * pz2;
                    const xg_v2f iz2 = (xg_v2f){az, az} + (xg_v2f){R2.z, R2.z} * pz2;
                    iy2 += (xg_v2f){fmvh, fmvh};
                    const xg_v2f zs2 = iz2 * iz2;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int zi = 2 * zp + h, z = z0 + zi;
                        const float ix = h ? ix2.y : ix2.x, iy = h ? iy2.y : iy2.x, zSqr = h ? zs2.y : zs2.x;
                        const bool zok = !(z < (zz & 0xffff) || z > (zz >> 16));
                        const bool pass = ((sph >> zi) & 1) && yok && zok && !(zSqr > radiusSqr) &&
                                          (ix >= reach.x) && (ix <= reach.y) && (iy >= reach.z) && (iy <= reach.w);
                        const unsigned long long pb = __ballot(pass);
                        if (pass) sQueue[qn + __popcll(pb & below)] = (unsigned short)(lane + 64 * zi);     // the voxel's accumulator: x + 8 y + 64 z
                        qn += __popcll(pb);
                    }
                }
                }
                if constexpr (ABL == 5) { if (lane == 0) atomicAdd(reinterpret_cast<int *>(tempV) + min(qn, NVOX), 1); }   // profiling: items per visit
                // ---- the patch of this visit has landed
                if constexpr (ABL != 1) xg_wait_vm0();
                // ---- dense pass
                const xg_v2f r01x = {R0.x, R1.x}, r01y = {R0.y, R1.y}, r01z = {R0.z, R1.z};
                const float uLo = 0.5f - H2.x, uHi = 0.5f + H2.x, tLo = 0.5f - H2.y, tHi = 0.5f + H2.y;
                if constexpr (ABL != 2 && ABL != 6 && ABL != 7) for (int b0 = 0; b0 < qn; b0 += 64) {
                    if (b0 + lane < qn) {
                        const int ai = sQueue[b0 + lane];
                        const float qx = fx0 + (float)(ai & 7), qy = fy0 + (float)((ai >> 3) & 7), qz = fz0 + (float)(ai >> 6);
                        // image coordinates of the voxel, (x, y) packed, with the reference's own operation order (RFA:643-647):
                        // a float coordinate near 256 has an ulp of 3e-5, which d2 * iDelta turns into 0.3 table entries
                        xg_v2f ixy = r01x * (xg_v2f){qx, qx} + r01y * (xg_v2f){qy, qy} + r01z * (xg_v2f){qz, qz};
                        ixy += (xg_v2f){0.f, fmvh};
                        const float iz = R2.x * qx + R2.y * qy + R2.z * qz;
                        float zSqr = iz * iz;
                        // the reference only visits rows that cross the top or bottom face of the slab (RFA:746-750): affine
                        // forms of u and t, exact evaluation where one of them lies within its band of 0 or 1
                        if (!allHit) {
                            const float vy = qy + fmvh, vz = qz + fmvh;
                            const xg_v2f ut1 = (xg_v2f){H0.x, H1.x} * (xg_v2f){vy, vy} + (xg_v2f){H0.y, H1.y} * (xg_v2f){vz, vz} + (xg_v2f){H0.z, H1.z};
                            const xg_v2f ut2 = ut1 + (xg_v2f){H0.w, H1.w};
                            const float cu1 = fabsf(ut1.x - 0.5f), ct1 = fabsf(ut1.y - 0.5f), cu2 = fabsf(ut2.x - 0.5f), ct2 = fabsf(ut2.y - 0.5f);
                            bool hit = (cu1 < uLo && ct1 < tLo) || (cu2 < uLo && ct2 < tLo);
                            const bool unsure = !((cu1 < uLo || cu1 > uHi) && (ct1 < tLo || ct1 > tHi) && (cu2 < uLo || cu2 > uHi) && (ct2 < tLo || ct2 > tHi));
                            if (__ballot(unsure)) {
                                if (unsure) {
                                    const float4 da = recs[kid].da, db = recs[kid].db;
                                    hit = d_hit(vy, vz, da.x, da.y, da.z, da.w, db.x, db.y) || d_hit(vy, vz, da.x, da.y, da.z, da.w, db.z, db.w);
                                }
                            }
                            if (!hit) zSqr = 3.0e38f;      // every tap fails the distance test
                        }
                        // first pixel of the W x W footprint, ceil(i - r) (RFA:655-658)
                        const float fbx = ceilf(ixy.x - fr), fby = ceilf(ixy.y - fr);
                        const int ry = (int)fby - oy, cx = (int)fbx - ox;       // 0..PN-W each
                        const __attribute__((address_space(3))) xg_v4f *tap =
                            (const __attribute__((address_space(3))) xg_v4f *)(uintptr_t)patchBase + (ry * PW + cx);
                        // (re, im) as one packed FMA per tap, the weight as a plain one; the fourth component of a record only keeps
                        // the read a ds_read_b128
                        xg_v2f accRI = {0.f, 0.f};
                        float accW = 0.f;
                        // the voxel's sums so far: requested now, needed at the very end
#if !XG_ACC_ATOMIC
                        const float oW = sAcc[ai], oR = sAcc[NVOX + ai], oI = sAcc[2 * NVOX + ai];
#endif
                        if constexpr (W == 4) {
                            // distances to the footprint's columns and rows as the reference forms them, i - (float)j (RFA:663,673)
                            const xg_v2f ix2 = {ixy.x, ixy.x}, iy2 = {ixy.y, ixy.y}, fbx2 = {fbx, fbx}, fby2 = {fby, fby};
                            const xg_v2f xa = ix2 - (fbx2 + (xg_v2f){0.f, 1.f}), xb = ix2 - (fbx2 + (xg_v2f){2.f, 3.f});
                            const xg_v2f ya = iy2 - (fby2 + (xg_v2f){0.f, 1.f}), yb = iy2 - (fby2 + (xg_v2f){2.f, 3.f});
                            const xg_v2f xs01 = xa * xa, xs23 = xb * xb;
                            const xg_v2f z2 = {zSqr, zSqr};
                            const xg_v2f yz01 = ya * ya + z2, yz23 = yb * yb + z2;
                            const float yz[4] = {yz01.x, yz01.y, yz23.x, yz23.y};
                            constexpr int NT = 16 / C::HALVES, NR = 4 / C::HALVES;     // taps and footprint rows per go
                            if constexpr (C::PIPE) {
                            // NR footprint rows at a time, the records and table entries of the next go requested before this go's
                            // multiply-adds (two register sets)
#ifndef XG_PIPE_DEPTH
#define XG_PIPE_DEPTH 1                                     // goes requested ahead of the multiply-adds (A/B: 2 = three register sets)
#endif
                            constexpr int PD = XG_PIPE_DEPTH, NS = PD + 1;
                            xg_v4f qq[NS][NT];
#if XG_TAPMASK
                            typedef xg_v2f xg_wt;           // a table entry in the low half of a register pair: the packed multiply-add's operand as it is
#else
                            typedef float xg_wt;
#endif
                            xg_wt ww[NS][NT];
#if XG_TAPMASK
                            // A tap beyond the blob (d > r^2, RFA:679) is switched off instead of being sent to the table's zero entry: its
                            // distance stays in a register until the multiply-adds, where v_cmpx takes its lane out of EXEC for the two of
                            // them (compare + select + 2 multiply-adds -> compare + 2 multiply-adds; what a dead lane read from the table
                            // -- its index is not clamped any more, the LDS returns zero or whatever lies there -- is never used)
                            float dd[NS][NT];
                            const unsigned long long lanesOn = __builtin_amdgcn_read_exec();
#endif
                            auto loadq = [&](int h, xg_v4f (&q_)[NT]) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) q_[t] = tap[(h * NR + t / 4) * PW + (t & 3)];
                            };
                            auto loadw = [&](int h, xg_wt (&w_)[NT], float *d_) {
                                int aux[NT];
#pragma unroll
                                for (int a = 0; a < NR; ++a) {
                                    const xg_v2f ya2 = {yz[h * NR + a], yz[h * NR + a]};
                                    const xg_v2f d01 = xs01 + ya2, d23 = xs23 + ya2;
                                    const xg_v2f t01 = d01 * idel2 + half2, t23 = d23 * idel2 + half2;
#if XG_TAPMASK
                                    d_[a * 4 + 0] = d01.x; d_[a * 4 + 1] = d01.y; d_[a * 4 + 2] = d23.x; d_[a * 4 + 3] = d23.y;
                                    aux[a * 4 + 0] = (int)t01.x; aux[a * 4 + 1] = (int)t01.y; aux[a * 4 + 2] = (int)t23.x; aux[a * 4 + 3] = (int)t23.y;
#else
                                    aux[a * 4 + 0] = (int)(d01.x > radiusSqr ? limf : t01.x);
                                    aux[a * 4 + 1] = (int)(d01.y > radiusSqr ? limf : t01.y);
                                    aux[a * 4 + 2] = (int)(d23.x > radiusSqr ? limf : t23.x);
                                    aux[a * 4 + 3] = (int)(d23.y > radiusSqr ? limf : t23.y);
#endif
                                }
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
#if XG_TAPMASK
                                    w_[t].x = sBlob[aux[t]];                    // (.y stays undefined on purpose: never read, op_sel_hi 0)
#else
                                    w_[t] = sBlob[aux[t]];
#endif
                                }
                            };
                            auto fmas = [&](xg_v4f (&q_)[NT], xg_wt (&w_)[NT], const float *d_) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
#if XG_TAPMASK
                                    const xg_v2f w2 = w_[t];                    // (the packed instruction reads the low half for both products: op_sel_hi 0)
                                    const xg_v2f qxy = {q_[t].x, q_[t].y};
                                    // !(d > r^2), the reference's test with the reference's operands: v_cmpx_nlt r^2, d
                                    asm volatile("v_cmpx_nlt_f32_e32 vcc, %[r2], %[d]\n\t"
                                                 "v_pk_fma_f32 %[ri], %[w2], %[qxy], %[ri] op_sel_hi:[0,1,1]\n\t"
                                                 "v_fmac_f32_e32 %[aw], %[w], %[qz]\n\t"
                                                 "s_mov_b64 exec, %[on]"
                                                 : [ri] "+v"(accRI), [aw] "+v"(accW)
                                                 : [r2] "s"(radiusSqr), [d] "v"(d_[t]), [w2] "v"(w2), [qxy] "v"(qxy), [w] "v"(w2.x), [qz] "v"(q_[t].z), [on] "s"(lanesOn)
                                                 : "vcc");
#else
                                    const xg_v2f w2 = {w_[t], w_[t]};
                                    accRI = __builtin_elementwise_fma(w2, (xg_v2f){q_[t].x, q_[t].y}, accRI);
                                    accW = __builtin_fmaf(w_[t], q_[t].z, accW);
#endif
                                    asm volatile("" :: "v"(q_[t].w));
                                }
                            };
#if XG_TAPMASK
#define XG_DD(i) dd[i]
#else
#define XG_DD(i) nullptr
#endif
#pragma unroll
                            for (int h = 0; h < PD && h < C::HALVES; ++h) {
                                loadq(h, qq[h % NS]);
                                if (h == 0) __builtin_amdgcn_sched_barrier(0);
                                loadw(h, ww[h % NS], XG_DD(h % NS));
                            }
#pragma unroll
                            for (int h = 0; h < C::HALVES; ++h) {
                                if (h + PD < C::HALVES) { loadq(h + PD, qq[(h + PD) % NS]); loadw(h + PD, ww[(h + PD) % NS], XG_DD((h + PD) % NS)); }
                                __builtin_amdgcn_sched_barrier(0);
                                fmas(qq[h % NS], ww[h % NS], XG_DD(h % NS));
                                __builtin_amdgcn_sched_barrier(0);
                            }
#undef XG_DD
                            } else
#pragma unroll
                            for (int h = 0; h < C::HALVES; ++h) {
                                // the records are requested before the index arithmetic of the weights, which they do not depend
                                // on: their latency passes under it (the scheduler, left alone, clusters them behind the table
                                // reads and the wave then waits for all of them)
                                xg_v4f q[NT];
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
                                    if constexpr (ABL == 9 || ABL == 11) {      // profiling: every lane reads the patch's first record
                                        asm volatile("" :: "v"(tap));
                                        q[t] = ((const __attribute__((address_space(3))) xg_v4f *)(uintptr_t)patchBase)[0];
                                    } else
                                    q[t] = tap[(h * NR + (t >> 2)) * PW + (t & 3)];
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                // table entry (int)(d2 * iDelta + 0.5) (RFA:682); a tap beyond the blob (d2 > r^2, RFA:679) reads a zero entry
                                int aux[NT];
#pragma unroll
                                for (int a = 0; a < NR; ++a) {
                                    const xg_v2f ya2 = {yz[h * NR + a], yz[h * NR + a]};
                                    const xg_v2f d01 = xs01 + ya2, d23 = xs23 + ya2;
                                    const xg_v2f t01 = d01 * idel2 + half2, t23 = d23 * idel2 + half2;
                                    aux[a * 4 + 0] = (int)(d01.x > radiusSqr ? limf : t01.x);
                                    aux[a * 4 + 1] = (int)(d01.y > radiusSqr ? limf : t01.y);
                                    aux[a * 4 + 2] = (int)(d23.x > radiusSqr ? limf : t23.x);
                                    aux[a * 4 + 3] = (int)(d23.y > radiusSqr ? limf : t23.y);
                                }
                                float wB[NT];
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
                                    if constexpr (ABL == 10 || ABL == 11) { asm volatile("" :: "v"(aux[t])); wB[t] = sBlob[0]; }      // profiling: one table entry
                                    else
                                    wB[t] = sBlob[aux[t]];
                                }
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
                                    const xg_v2f w2 = {wB[t], wB[t]};
                                    accRI = __builtin_elementwise_fma(w2, (xg_v2f){q[t].x, q[t].y}, accRI);
                                    accW = __builtin_fmaf(wB[t], q[t].z, accW);
                                    // the unused fourth component stays allocated up to here: handed out again earlier, its
                                    // register makes the index arithmetic wait for the record reads (write after write)
                                    asm volatile("" :: "v"(q[t].w));
                                }
                            }
                        } else {
                            // wider blobs: the same taps row by row
                            float xs[W];
#pragma unroll
                            for (int b = 0; b < W; ++b) { const float xD = ixy.x - (fbx + (float)b); xs[b] = xD * xD; }
#pragma unroll
                            for (int a = 0; a < W; ++a) {
                                const float yD = ixy.y - (fby + (float)a);
                                const float yzS = yD * yD + zSqr;
                                float wB[W];
                                xg_v4f q[W];
#pragma unroll
                                for (int b = 0; b < W; ++b) {
                                    const float d2 = xs[b] + yzS;
                                    wB[b] = sBlob[(int)(d2 > radiusSqr ? limf : d2 * iDeltaSqrt + 0.5f)];
                                    q[b] = tap[a * PW + b];
                                }
#pragma unroll
                                for (int b = 0; b < W; ++b) {
                                    const xg_v2f w2 = {wB[b], wB[b]};
                                    accRI = __builtin_elementwise_fma(w2, (xg_v2f){q[b].x, q[b].y}, accRI);
                                    accW = __builtin_fmaf(wB[b], q[b].z, accW);
                                }
                            }
                        }
#if XG_ACC_ATOMIC
                        // A/B build: three LDS additions without a return instead of three reads, three adds and three writes
                        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(sAcc + ai), accW, 0, 0, false);
                        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(sAcc + NVOX + ai), accRI.x, 0, 0, false);
                        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(sAcc + 2 * NVOX + ai), accRI.y, 0, 0, false);
#else
                        sAcc[ai] = oW + accW;
                        sAcc[NVOX + ai] = oR + accRI.x;
                        sAcc[2 * NVOX + ai] = oI + accRI.y;
#endif
                    }
                }
                __builtin_amdgcn_wave_barrier();
                R0 = N0; R1 = N1; R2 = N2;
                kid = kidN;
                kw = __builtin_amdgcn_readfirstlane(kwN);
                kidN = __builtin_amdgcn_readfirstlane(kidNN) & 0x3fffffff;
            }
        }
        // ---- write-back
#pragma unroll
        for (int zi = 0; zi < ZD; ++zi) {
            const float aW = sAcc[zi * 64 + lane], aR = sAcc[NVOX + zi * 64 + lane], aI = sAcc[2 * NVOX + zi * 64 + lane];
            if (((sph >> zi) & 1) && (aW != 0.f || aR != 0.f || aI != 0.f)) {
                const size_t vi = ((size_t)(z0 + zi) * dim + y) * dim + x;
                float2 *V = reinterpret_cast<float2 *>(tempV) + vi;
                float2 v = *V;
                v.x += aR;
                v.y += aI;
                *V = v;
                tempW[vi] += aW;
            }
        }
    }
}

#endif
