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

#ifndef XG_NW
#define XG_NW 12                // waves per workgroup (= per CU: the LDS budget admits one workgroup)
#endif
#define XG_NBUF 1               // patch buffers per wave
#define XG_KCAP 256             // surviving projections listed per cull phase
#define XG_PATCH_BYTES 4096     // 16 x 16 pixels x 16 B
#define XG_LDS_PATCH 0                                   // [XG_NW][2][4096]; first, so that LDS-DMA bases stay below 64 KB
#define XG_LDS_BLOB (XG_NW * XG_NBUF * XG_PATCH_BYTES)         // float[XH_BLOB_TABLE + 4]; entry XH_BLOB_TABLE is 0
#define XG_LDS_ACC (XG_LDS_BLOB + 4 * (XH_BLOB_TABLE + 4))   // [XG_NW][3][256] float
#define XG_LDS_KEPT (XG_LDS_ACC + XG_NW * 3 * 256 * 4)   // [XG_NW][XG_KCAP] int
#define XG_LDS_QUEUE (XG_LDS_KEPT + XG_NW * XG_KCAP * 4) // [XG_NW][256] uchar
#define XG_LDS_RING (XG_LDS_QUEUE + XG_NW * 256)         // tile ring: int[8] tiles, int[8] ready, ticket, hop
#define XG_LDS_TOTAL (XG_LDS_RING + 4 * 32)

struct XgRec { float4 r0, r1, r2, da, db; };   // tInv rows (.w: image index, minY | maxY << 16, minZ | maxZ << 16), getX operands

// one LDS-DMA instruction: every lane's 16 bytes at g land at ldsBase + 16 * lane (ldsBase wave-uniform, in an SGPR)
__device__ __forceinline__ void xg_dma16(const void *g, unsigned ldsBase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(ldsBase) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit byte offset per lane
__device__ __forceinline__ void xg_dma16s(const void *base, unsigned off, unsigned ldsBase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(ldsBase) : "memory");
}
typedef float xg_v2f __attribute__((ext_vector_type(2)));
typedef float xg_v4f __attribute__((ext_vector_type(4)));
// hit part of getX without the two IEEE divisions: u in (0, 1) is decided exactly from the signs and moduli of numerator
// and denominator; t needs the value of u, taken as num * (1 / den) -- whenever t comes within the error of that
// shortcut of 0 or 1 the lane asks for the exact evaluation (d_hit)
__device__ __forceinline__ bool xg_hit_fast(float y, float z, float a1, float a2, float b1, float y0, float z0, float den, float rden, bool &near)
{
    const float num = (z - z0) * a1 + (y0 - y) * a2;
    const bool uOk = ((num > 0.f && den > 0.f) || (num < 0.f && den < 0.f)) && (fabsf(num) < fabsf(den));
    const float yy = -y0 + y;
    const float ub = (num * rden) * b1;
    const float tn = yy - ub;
    const float tol = 6e-7f * (fabsf(ub) + fabsf(yy));
    const float atn = fabsf(tn), aa1 = fabsf(a1);
    near = near || (uOk && (atn < tol + 1e-30f || fabsf(atn - aa1) < tol || !(aa1 < 1e6f)));
    return uOk && ((tn > 0.f) == (a1 > 0.f)) && (atn < aa1);
}
__device__ __forceinline__ void xg_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// float4 (re*ctf*mod*w, im*ctf*mod*w, mod*w, 0) per pixel, XH_PAD zero cells on every side
__global__ void k_rf_pack_grid(const xh_cf *__restrict__ ffts, const float *__restrict__ ctfs, const float *__restrict__ mods,
                               const float *__restrict__ weights, float4 *__restrict__ pk, int n, int sizeX, int sizeY)
{
    const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;
    const size_t total = (size_t)n * SY * SX;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = gid % SX - XH_PAD;
    const int y = (gid / SX) % SY - XH_PAD;
    const size_t img = gid / ((size_t)SX * SY);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < sizeX && y >= 0 && y < sizeY) {
        const size_t o = img * sizeX * sizeY + (size_t)y * sizeX + x;
        const xh_cf f = ffts[o];
        const float w = weights ? weights[img] : 1.f;
        if (ctfs) {
            const float mw = mods[o] * w, c = ctfs[o];
            v = make_float4(f.x * mw * c, f.y * mw * c, mw, 0.f);
        } else v = make_float4(f.x * w, f.y * w, w, 0.f);
    }
    pk[gid] = v;
}

__global__ void __launch_bounds__(64 * XG_NW, (XG_NW + 3) / 4)
k_rf_grid(const XgRec *__restrict__ recs, const float4 *__restrict__ pk, const float *__restrict__ blobTable,
          float *__restrict__ tempV, float *__restrict__ tempW, int mv, float iDeltaSqrt, double blobRadius,
          const unsigned *__restrict__ tileList, const int *__restrict__ classOff, int *__restrict__ counter,
          const int *__restrict__ superList, const int *__restrict__ superCount, int superDim, int superCap,
          const float4 *__restrict__ superN, const float4 *__restrict__ superX, float4 reach, int dbg)
{
    __shared__ __align__(16) unsigned char lds[XG_LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sBlob = reinterpret_cast<float *>(lds + XG_LDS_BLOB);
    float *sAcc = reinterpret_cast<float *>(lds + XG_LDS_ACC) + wv * 3 * 256;
    int *sKept = reinterpret_cast<int *>(lds + XG_LDS_KEPT) + wv * XG_KCAP;
    unsigned char *sQueue = lds + XG_LDS_QUEUE + wv * 256;
    int *sTile = reinterpret_cast<int *>(lds + XG_LDS_RING), *sReady = sTile + 8, *sTicket = sTile + 16, *sHop = sTile + 17;
    const unsigned char *sPatch = lds + XG_LDS_PATCH + wv * XG_NBUF * XG_PATCH_BYTES;
    const unsigned patchBase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)sPatch;
    const int sizeX = mv / 2, sizeY = mv, dim = mv + 1;
    const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;

    for (int i = tid; i < XH_BLOB_TABLE + 4; i += 64 * XG_NW) sBlob[i] = i < XH_BLOB_TABLE ? blobTable[i] : 0.f;
    const float fr = (float)blobRadius;
    const float maxDistanceSqr = (sizeX + blobRadius) * (sizeX + blobRadius);
    const float radiusSqr = blobRadius * blobRadius;
    const int lx = lane & 7, ly = lane >> 3;
    const float fmvh = (float)(mv / 2);
    const xg_v2f idel2 = {iDeltaSqrt, iDeltaSqrt}, half2 = {0.5f, 0.5f};
    const float limf = (float)XH_BLOB_TABLE;
    const unsigned long long below = (1ull << lane) - 1ull;
    // which pixel of a patch lane l of DMA instruction i fetches: LDS slot 64 i + l holds row (64 i + l) / 16 and, rows
    // being rotated by their index against bank conflicts, column ((64 i + l) - row) mod 16
    unsigned dOff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int slot = 64 * i + lane, row = slot >> 4, col = (slot - row) & 15; dOff[i] = (unsigned)(row * SX + col) * 16u; }

    // ---- work distribution: the tile ring of the previous kernel (tiles = 2 x 2 x 2 units, eight XCD classes dealt into
    // NSUB interleaved streams, one global grab per tile, eight tickets per tile drawn by the waves of the workgroup)
    constexpr int NSUB = 8, RS = 8;
    const int home = (blockIdx.x & 7) * NSUB + ((blockIdx.x >> 3) & (NSUB - 1));
    auto streamTiles = [&](int st) {
        const int c = st / NSUB, j = st % NSUB, nt = classOff[c + 1] - classOff[c];
        return nt > j ? (nt - j + NSUB - 1) / NSUB : 0;
    };
    auto produce = [&](int q) {
        while (q > 0 && atomicAdd(&sReady[(q - 1) % RS], 0) != q) __builtin_amdgcn_s_sleep(1);
        int tile = -1;
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
        const int tx = packed & 0xff, ty = (packed >> 8) & 0xff, tz = (packed >> 16) & 0xff;
        const int x0 = tx * 16 + (sub & 1) * 8, y0 = ty * 16 + ((sub >> 1) & 1) * 8, z0 = tz * 8 + (sub >> 2) * 4;
        const int x = x0 + lx, y = y0 + ly;
        const float px = x - mv / 2, py = y - mv / 2;
        int sph = 0;      // bit zi: voxel (x, y, z0 + zi) lies in the volume and inside the sphere the reference keeps (RFA:640)
#pragma unroll
        for (int zi = 0; zi < 4; ++zi) {
            const int z = z0 + zi;
            const float pz = z - mv / 2;
            const bool in = (x <= mv) && (y <= mv) && (z <= mv) && !((px * px + py * py + pz * pz) > maxDistanceSqr);
            sph |= (int)in << zi;
        }
        if (!__ballot(sph != 0)) continue;
#pragma unroll
        for (int i = 0; i < 12; ++i) sAcc[i * 64 + lane] = 0.f;
        const float fx0 = (float)(x0 - mv / 2), fy0 = (float)(y0 - mv / 2), fz0 = (float)(z0 - mv / 2);
        const float ucx = x0 + 3.5f - mv / 2, ucy = y0 + 3.5f - mv / 2, ucz = z0 + 1.5f - mv / 2;
        const int sup = ((tz >> 1) * superDim + ty) * superDim + tx;
        const int nlist = superCount[sup];
        const int *lst = superList + (size_t)sup * superCap;
        const float4 *lstN = superN + (size_t)sup * superCap, *lstX = superX + (size_t)sup * superCap;

        int pos = 0;
        int sNext = 0;
        float4 nNext = make_float4(0.f, 0.f, 0.f, 0.f), xNext = nNext;
        if (lane < nlist) { sNext = lst[lane]; nNext = lstN[lane]; xNext = lstX[lane]; }
        while (pos < nlist) {
            // ---- cull phase: lane <-> list entry
            int nk = 0;
            while (pos < nlist && nk <= XG_KCAP - 64) {
                const int sIdx = sNext;
                const float4 n = nNext, xv = xNext;
                const int e = pos + lane, en = e + 64;
                if (en < nlist) { sNext = lst[en]; nNext = lstN[en]; xNext = lstX[en]; }
                bool keep = false;
                if (e < nlist) {
                    const float dn = n.x * ucx + n.y * ucy + n.z * ucz;
                    const float dx = xv.x * ucx + xv.y * ucy + xv.z * ucz;
                    // support function of the box of voxel centres (half extents 3.5, 3.5, 1.5), never wider than its sphere
                    const float hn = fminf(5.2f, 3.5f * (fabsf(n.x) + fabsf(n.y)) + 1.5f * fabsf(n.z) + 0.02f);
                    const float hx = fminf(5.2f, 3.5f * (fabsf(xv.x) + fabsf(xv.y)) + 1.5f * fabsf(xv.z) + 0.02f);
                    keep = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
                }
                const unsigned long long bal = __ballot(keep);
                if (keep) sKept[nk + __popcll(bal & below)] = sIdx;
                nk += __popcll(bal);
                pos += 64;
            }
            if (nk == 0) continue;
            __builtin_amdgcn_wave_barrier();

            // ---- visits. patchOf(rec): origin of the 16 x 16 patch = first footprint pixel of the block's corner with the
            // smallest image coordinates (the block's image extent is < 10.4 pixels, its footprints span < 16)
            auto issue = [&](const XgRec &R, int buf) {
                const float cix = R.r0.x * ucx + R.r0.y * ucy + R.r0.z * ucz;
                const float ciy = R.r1.x * ucx + R.r1.y * ucy + R.r1.z * ucz + (float)(mv / 2);
                const float ex = 3.5f * (fabsf(R.r0.x) + fabsf(R.r0.y)) + 1.5f * fabsf(R.r0.z) + 0.01f;
                const float ey = 3.5f * (fabsf(R.r1.x) + fabsf(R.r1.y)) + 1.5f * fabsf(R.r1.z) + 0.01f;
                // the patch stays inside the padded record: footprints never leave it, so an origin moved inwards
                // still covers them
                const int ox = min(max(__builtin_amdgcn_readfirstlane((int)ceilf(cix - ex - fr)), -XH_PAD), SX - XH_PAD - 16);
                const int oy = min(max(__builtin_amdgcn_readfirstlane((int)ceilf(ciy - ey - fr)), -XH_PAD), SY - XH_PAD - 16);
                if (dbg == 3) return make_int2(ox, oy);
                const size_t cell = ((size_t)__float_as_int(R.r0.w) * SY + (oy + XH_PAD)) * SX + (ox + XH_PAD);
                const float4 *base = pk + cell;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base);
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
                const void *sb = (const void *)(((uintptr_t)hi << 32) | lo);
#pragma unroll
                for (int i = 0; i < 4; ++i) xg_dma16s(sb, dOff[i], patchBase + buf * XG_PATCH_BYTES + i * 1024);
                return make_int2(ox, oy);
            };
            // records travel through scalar registers, one visit ahead (unconditional loads: a select would make the
            // compiler wait for them on the spot); the index of the record after that is read from LDS meanwhile
            XgRec R = recs[__builtin_amdgcn_readfirstlane(sKept[0])];
            int kidN = __builtin_amdgcn_readfirstlane(sKept[min(1, nk - 1)]);
            for (int k = 0; k < nk; ++k) {
                const XgRec Rn = recs[kidN];
                const int kidNN = sKept[min(k + 2, nk - 1)];
                const int2 org = issue(R, 0);     // the patch buffer is free: the previous dense pass has consumed its reads
                // ---- sparse pass (RFA:631-653 and the reach of the footprint)
                const int yy = __float_as_int(R.r1.w), zz = __float_as_int(R.r2.w);
                const bool yok = !(y < (yy & 0xffff) || y > (yy >> 16));
                const float ax = R.r0.x * px + R.r0.y * py, ay = R.r1.x * px + R.r1.y * py, az = R.r2.x * px + R.r2.y * py;
                int qn = 0;
#pragma unroll
                for (int zi = 0; zi < 4; ++zi) {
                    const int z = z0 + zi;
                    const float pz = z - mv / 2;
                    const float ix = ax + R.r0.z * pz;
                    float iy = ay + R.r1.z * pz;
                    const float iz = az + R.r2.z * pz;
                    iy += mv / 2;
                    const float zSqr = iz * iz;
                    const bool zok = !(z < (zz & 0xffff) || z > (zz >> 16));
                    const bool pass = ((sph >> zi) & 1) && yok && zok && !(zSqr > radiusSqr) &&
                                      (ix >= reach.x) && (ix <= reach.y) && (iy >= reach.z) && (iy <= reach.w);
                    const unsigned long long pb = __ballot(pass);
                    if (pass) sQueue[qn + __popcll(pb & below)] = (unsigned char)((zi << 6) | lane);
                    qn += __popcll(pb);
                }
                // ---- the patch of this visit has landed
                if (dbg != 1) xg_wait_vm0();
                // ---- dense pass
                const float hden = R.da.x * R.da.w - R.da.z * R.da.y;      // a1 * b2 - b1 * a2, as d_hit forms it
                const float hrden = __builtin_amdgcn_rcpf(hden);
                const xg_v2f r01x = {R.r0.x, R.r1.x}, r01y = {R.r0.y, R.r1.y}, r01z = {R.r0.z, R.r1.z};
                const unsigned patchAddr = patchBase;
                if (dbg != 2) for (int b0 = 0; b0 < qn; b0 += 64) {
                    if (b0 + lane < qn) {
                        const int id = sQueue[b0 + lane];
                        const float qx = fx0 + (float)(id & 7), qy = fy0 + (float)((id >> 3) & 7), qz = fz0 + (float)(id >> 6);
                        // image coordinates of the voxel, (x, y) packed, with the reference's own operation order (RFA:643-647):
                        // a float coordinate near 256 has an ulp of 3e-5, which d2 * iDelta turns into 0.3 table entries
                        xg_v2f ixy = r01x * (xg_v2f){qx, qx} + r01y * (xg_v2f){qy, qy} + r01z * (xg_v2f){qz, qz};
                        ixy += (xg_v2f){0.f, fmvh};
                        const float iz = R.r2.x * qx + R.r2.y * qy + R.r2.z * qz;
                        float zSqr = iz * iz;
                        // the reference only visits rows that cross the top or bottom face of the slab (RFA:746-750)
                        bool near = false;
                        bool hit1 = xg_hit_fast(qy + fmvh, qz + fmvh, R.da.x, R.da.y, R.da.z, R.db.x, R.db.y, hden, hrden, near);
                        bool hit2 = xg_hit_fast(qy + fmvh, qz + fmvh, R.da.x, R.da.y, R.da.z, R.db.z, R.db.w, hden, hrden, near);
                        if (__ballot(near)) {
                            if (near) {
                                hit1 = d_hit(qy + fmvh, qz + fmvh, R.da.x, R.da.y, R.da.z, R.da.w, R.db.x, R.db.y);
                                hit2 = d_hit(qy + fmvh, qz + fmvh, R.da.x, R.da.y, R.da.z, R.da.w, R.db.z, R.db.w);
                            }
                        }
                        if (!(hit1 || hit2)) zSqr = 3.0e38f;      // every tap fails the distance test
                        // first pixel of the 4 x 4 footprint, ceil(i - r) (RFA:655-658), and the offsets from it
                        const float fbx = ceilf(ixy.x - fr), fby = ceilf(ixy.y - fr);
                        // distances to the footprint's columns and rows as the reference forms them, i - (float)j (RFA:663,673)
                        const xg_v2f ix2 = {ixy.x, ixy.x}, iy2 = {ixy.y, ixy.y}, fbx2 = {fbx, fbx}, fby2 = {fby, fby};
                        const xg_v2f xa = ix2 - (fbx2 + (xg_v2f){0.f, 1.f}), xb = ix2 - (fbx2 + (xg_v2f){2.f, 3.f});
                        const xg_v2f ya = iy2 - (fby2 + (xg_v2f){0.f, 1.f}), yb = iy2 - (fby2 + (xg_v2f){2.f, 3.f});
                        const xg_v2f xs01 = xa * xa, xs23 = xb * xb;
                        const xg_v2f z2 = {zSqr, zSqr};
                        const xg_v2f yz01 = ya * ya + z2, yz23 = yb * yb + z2;
                        const float yz[4] = {yz01.x, yz01.y, yz23.x, yz23.y};
                        const int ry = (int)fby - org.y, cx = (int)fbx - org.x;       // 0..12 each
                        const unsigned rowBase = patchAddr + (unsigned)ry * 256u;     // low 8 bits clear
                        const int c16 = (cx + ry) << 4;
                        unsigned colAddr[7];
#pragma unroll
                        for (int s = 0; s < 7; ++s) colAddr[s] = ((unsigned)(c16 + 16 * s) & 0xf0u) | rowBase;
                        // table entry (int)(d2 * iDelta + 0.5) (RFA:682); a tap beyond the blob (d2 > r^2, RFA:679) reads a zero entry
                        int aux[16];
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const xg_v2f ya2 = {yz[a], yz[a]};
                            const xg_v2f d01 = xs01 + ya2, d23 = xs23 + ya2;
                            const xg_v2f t01 = d01 * idel2 + half2, t23 = d23 * idel2 + half2;
                            aux[a * 4 + 0] = (int)(d01.x > radiusSqr ? limf : t01.x);
                            aux[a * 4 + 1] = (int)(d01.y > radiusSqr ? limf : t01.y);
                            aux[a * 4 + 2] = (int)(d23.x > radiusSqr ? limf : t23.x);
                            aux[a * 4 + 3] = (int)(d23.y > radiusSqr ? limf : t23.y);
                        }
                        float wB[16];
                        xg_v4f q[16];
#pragma unroll
                        for (int t = 0; t < 16; ++t) wB[t] = sBlob[aux[t]];
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            q[t] = *(const __attribute__((address_space(3))) xg_v4f *)(uintptr_t)(colAddr[(t >> 2) + (t & 3)] + (t >> 2) * 256);
                        // (re, im) and (weight, 0) as two packed FMAs per tap; the fourth component is 0 in every record
                        xg_v2f accRI = {0.f, 0.f}, accWZ = {0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const xg_v2f w2 = {wB[t], wB[t]};
                            accRI = __builtin_elementwise_fma(w2, (xg_v2f){q[t].x, q[t].y}, accRI);
                            accWZ = __builtin_elementwise_fma(w2, (xg_v2f){q[t].z, q[t].w}, accWZ);
                        }
                        const float vW = accWZ.x + accWZ.y, vR = accRI.x, vI = accRI.y;
                        const int ai = id;
                        sAcc[ai] += vW;
                        sAcc[256 + ai] += vR;
                        sAcc[512 + ai] += vI;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                R = Rn;
                kidN = __builtin_amdgcn_readfirstlane(kidNN);
            }
        }
        // ---- write-back
#pragma unroll
        for (int zi = 0; zi < 4; ++zi) {
            const float aW = sAcc[zi * 64 + lane], aR = sAcc[256 + zi * 64 + lane], aI = sAcc[512 + zi * 64 + lane];
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
