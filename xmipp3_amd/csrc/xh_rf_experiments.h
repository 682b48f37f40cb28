// xh_rf_experiments.h -- earlier forms of the gridding kernel, kept for A/B measurements: the atomic scatter kernel
// (thread per row, three float atomics per voxel), the tile kernel (workgroup = 8^3 tile), the wave-per-sub-cube kernel
// (per-lane 4x4 gathers from global memory, bit-identical to the reference) and the LDS-patch tile kernel.
// Only compiled with -DXH_EXPERIMENTS (xh_rf_set_option "tile_variant" 0/1/2, "insert_variant" 3); the shipped library
// has one gridding kernel, k_rf_grid (xh_rf_grid.h).
#ifndef XH_RF_EXPERIMENTS_H
#define XH_RF_EXPERIMENTS_H

#define XH_TILE 16
// persistent blocks: block b serves spaces s == b%8 (mod 8) so that one projection's
// spectrum stays in one XCD's L2 (block b runs on XCD b%8).
template <bool HAS_CTF, bool FAST>
__global__ void __launch_bounds__(256)
k_rf_insert(const XhSpace *__restrict__ spaces, int nspaces, const xh_cf *__restrict__ ffts,
            const float *__restrict__ ctfs, const float *__restrict__ mods,
            const float *__restrict__ blobTable, float *__restrict__ tempV, float *__restrict__ tempW,
            int mv, float iDeltaSqrt, double blobRadius, int variant)
{
    __shared__ float sBlob[XH_BLOB_TABLE];
    if (!FAST) {
        for (int i = threadIdx.x; i < XH_BLOB_TABLE; i += blockDim.x) sBlob[i] = blobTable[i];
        __syncthreads();
    }
    const int tilesPerDim = (mv + 1 + XH_TILE - 1) / XH_TILE;
    const int tilesPerSpace = tilesPerDim * tilesPerDim;
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int spacesHere = (nspaces - xcd + 7) >> 3;
    const long long items = (long long)spacesHere * tilesPerSpace;
    const int ty = threadIdx.x & (XH_TILE - 1), tz = threadIdx.x >> 4;
    const int sizeX = mv / 2, sizeY = mv;
    const float maxDistanceSqr = (sizeX + (FAST ? 0.f : blobRadius)) * (sizeX + (FAST ? 0.f : blobRadius));
    const float radiusSqr = blobRadius * blobRadius;
    const int dim = mv + 1;
    for (long long it = rank; it < items; it += nb) {
        const int sIdx = xcd + 8 * (int)(it / tilesPerSpace);
        const int tile = (int)(it % tilesPerSpace);
        const XhSpace &S = spaces[sIdx];
        const int y = (tile % tilesPerDim) * XH_TILE + ty;
        const int z = (tile / tilesPerDim) * XH_TILE + tz;
        if (y < S.minY || y > S.maxY || z < S.minZ || z > S.maxZ) continue;
        const xh_cf *img = ffts + (size_t)S.img * sizeX * sizeY;
        const float *CTF = HAS_CTF ? ctfs + (size_t)S.img * sizeX * sizeY : nullptr;
        const float *MOD = HAS_CTF ? mods + (size_t)S.img * sizeX * sizeY : nullptr;
        const float dataWeight = S.weight;
        if (FAST) {
            float hitX;
            if (!d_getX(hitX, (float)y, (float)z, S.u, S.v, S.p0)) continue;
            const int x = (int)(hitX + 0.5f);
            // processVoxel RFA:595-625
            float px = x - mv / 2, py = y - mv / 2, pz = z - mv / 2;
            if (px * px + py * py + pz * pz > maxDistanceSqr) continue;
            const float ix = S.tInv[0] * px + S.tInv[1] * py + S.tInv[2] * pz;
            const float iy = S.tInv[3] * px + S.tInv[4] * py + S.tInv[5] * pz;
            int imgX = (int)(ix + 0.5f);
            imgX = imgX > sizeX - 1 ? sizeX - 1 : imgX;
            imgX = imgX < 0 ? 0 : imgX;
            int imgY = (int)(iy + 0.5f + mv / 2);
            imgY = imgY > sizeY - 1 ? sizeY - 1 : imgY;
            imgY = imgY < 0 ? 0 : imgY;
            float wCTF = 1.f, wMod = 1.f;
            if (HAS_CTF) { wCTF = CTF[(size_t)imgY * sizeX + imgX]; wMod = MOD[(size_t)imgY * sizeX + imgX]; }
            const float weight = 1.f * wMod * dataWeight;
            const xh_cf pix = img[(size_t)imgY * sizeX + imgX];
            const size_t vi = ((size_t)z * dim + y) * dim + x;
            atomicAdd(&tempV[2 * vi], pix.x * weight * wCTF);
            atomicAdd(&tempV[2 * vi + 1], pix.y * weight * wCTF);
            atomicAdd(&tempW[vi], weight);
            continue;
        }
        float x1, x2;
        const bool hit1 = d_getX(x1, (float)y, (float)z, S.u, S.v, S.p0);
        const bool hit2 = d_getX(x2, (float)y, (float)z, S.u, S.v, S.p4);
        if (!(hit1 || hit2)) continue;
        const float fmv = (float)mv;
        x1 = x1 > fmv ? fmv : x1; x1 = x1 < 0.f ? 0.f : x1;
        x2 = x2 > fmv ? fmv : x2; x2 = x2 < 0.f ? 0.f : x2;
        // std::min / std::max semantics of the reference (RFA:752-753)
        const float lower = (x2 < x1) ? x2 : x1, upper = (x1 < x2) ? x2 : x1;
        if (!(lower >= 0.f) || !(upper <= fmv)) continue;  // non-finite bounds: undefined in the reference
        const int xEnd = (int)ceilf(upper);
        for (int x = (int)floorf(lower); x <= xEnd; x++) {
            // processVoxelBlob RFA:627-700
            float px = x - mv / 2, py = y - mv / 2, pz = z - mv / 2;
            if ((px * px + py * py + pz * pz) > maxDistanceSqr) continue;
            const float ix = S.tInv[0] * px + S.tInv[1] * py + S.tInv[2] * pz;
            float iy = S.tInv[3] * px + S.tInv[4] * py + S.tInv[5] * pz;
            const float iz = S.tInv[6] * px + S.tInv[7] * py + S.tInv[8] * pz;
            iy += mv / 2;
            const float zSqr = iz * iz;
            if (zSqr > radiusSqr) continue;
            int minX = (int)ceil((double)ix - blobRadius);
            int maxX = (int)floor((double)ix + blobRadius);
            int minY = (int)ceil((double)iy - blobRadius);
            int maxY = (int)floor((double)iy + blobRadius);
            minX = max(minX, 0);
            minY = max(minY, 0);
            maxX = min(maxX, sizeX - 1);
            maxY = min(maxY, sizeY - 1);
            float accW = 0.f, accR = 0.f, accI = 0.f;
            if (variant == 2) { accW = 1.f; accR = ix; accI = iy; minY = maxY + 1; }   // experiment: atomics only
            for (int i = minY; i <= maxY; i++) {
                const float ySqr = (iy - i) * (iy - i);
                const float yzSqr = ySqr + zSqr;
                if (yzSqr > radiusSqr) continue;
                for (int j = minX; j <= maxX; j++) {
                    const float xD = ix - j;
                    const float distanceSqr = xD * xD + yzSqr;
                    if (distanceSqr > radiusSqr) continue;
                    const int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                    const float wBlob = sBlob[aux];
                    const xh_cf pix = img[(size_t)i * sizeX + j];
                    if (HAS_CTF) {
                        const float wCTF = CTF[(size_t)i * sizeX + j];
                        const float wModulator = MOD[(size_t)i * sizeX + j];
                        const float weight = wBlob * wModulator * dataWeight;
                        accW += weight;
                        accR += pix.x * weight * wCTF;
                        accI += pix.y * weight * wCTF;
                    } else {
                        const float weight = wBlob * dataWeight;
                        accW += weight;
                        accR += pix.x * weight;
                        accI += pix.y * weight;
                    }
                }
            }
            if (variant == 1) {   // experiment: everything but the atomics
                if (accW == 123456.789f) tempW[0] = accR + accI;
            } else if (accW != 0.f || accR != 0.f || accI != 0.f) {
                const size_t vi = ((size_t)z * dim + y) * dim + x;
                atomicAdd(&tempV[2 * vi], accR);
                atomicAdd(&tempV[2 * vi + 1], accI);
                atomicAdd(&tempW[vi], accW);
            }
        }
    }
}


// ---- gridding, output-stationary form ------------------------------------------------------
// HBM float atomics top out at ~19.5 G/s on MI355X (profiles/README.md): 3 per slab voxel made
// the scatter kernel above ~60 us per 256-px projection.  Here every workgroup OWNS one 8x8x8
// tile of the temp volume: it culls the launch's projections against the tile (plane distance
// + half-plane extent, data staged in LDS), lets each thread gather the Kaiser-Bessel taps of
// its single voxel from every surviving projection into registers, and finishes with one plain
// read-modify-write.  No atomics, run-to-run deterministic; a voxel receives exactly the
// contributions processVoxelBlob (RFA:627-700) would give it when the reference's traversal
// (RFA:743-761: AABB rows, hit1||hit2 via getX) visits it.

// a value every lane holds identically (read from one LDS address): park it in scalar registers.
// Only the CTF variant of the tile kernel does this: it is the one short of vector registers (float4
// records), while the plain variant would merely trade them for scalar spills.
template <bool ON> __device__ __forceinline__ float4 d_uniform(float4 v)
{
    if (!ON) return v;
    return make_float4(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.y))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.z))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.w))));
}

// ---- packed, padded projection records for the tile kernel -----------------------------------
// pk[n][sizeY+8][sizeX+8], 4 pad cells on every side (zero): a voxel's 4x4 footprint can then be
// fetched as four contiguous row segments without clamping. HAS_CTF: float4 (re, im, ctf, mod);
// otherwise float2 (re, im).
#define XH_PAD 4
template <bool HAS_CTF>
__global__ void k_rf_pack(const xh_cf *__restrict__ ffts, const float *__restrict__ ctfs, const float *__restrict__ mods,
                          void *__restrict__ pk, int n, int sizeX, int sizeY)
{
    const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;
    const size_t total = (size_t)n * SY * SX;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = gid % SX - XH_PAD;
    const int y = (gid / SX) % SY - XH_PAD;
    const size_t img = gid / ((size_t)SX * SY);
    const bool in = x >= 0 && x < sizeX && y >= 0 && y < sizeY;
    const size_t o = img * sizeX * sizeY + (size_t)y * sizeX + x;
    if (HAS_CTF) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) { const xh_cf f = ffts[o]; v = make_float4(f.x, f.y, ctfs[o], mods[o]); }
        reinterpret_cast<float4 *>(pk)[gid] = v;
    } else {
        float2 v = make_float2(0.f, 0.f);
        if (in) { const xh_cf f = ffts[o]; v = make_float2(f.x, f.y); }
        reinterpret_cast<float2 *>(pk)[gid] = v;
    }
}

#define XH_CHUNK 256      // projections culled per block-level pass (capacity of the LDS hit records)
#define XH_QCAP 128       // per-wave work queue capacity (64 pending + 64 new)
#define XH_GRAB 8         // tiles fetched per work-queue atomic
#define XH_CTF_ROWS2 0
#define XH_SEGCAP 16      // queue segments (one per projection) a dense pass can span
// Workgroup = one 8x8x8 tile, 8 waves; wave w owns the 4x4x4 sub-cube (w&1, (w>>1)&1, w>>2) and
// lane l the voxel (l&3, (l>>2)&3, l>>4) of it.
//  block level : cull the launch's projections against the tile, 256 at a time, with an ordered
//                compaction that also stages the survivors' constants in LDS
//  wave level  : scalar cull of each survivor against the sub-cube, then a SPARSE test pass
//                (voxel -> image coordinates, slab / reach tests) that only enqueues
//                (voxel, projection) items, and a DENSE pass that pops 64 items at a time so the
//                expensive part (row-visit test + Kaiser-Bessel taps) runs with every lane busy.
//  accumulation in wave-private LDS (ds_add_f32), one plain read-modify-write of the volume at the end.
// Tiles are assigned statically: block b (XCD b%8) walks the tiles of z-layer class b%8 (tz mod 8),
// raster order, stride gridDim/8 -- neighbouring tiles, which share most projections and adjacent
// image patches, meet in one L2, and no dequeue latency sits between tiles.
struct XhHitRec { float4 r0, r1, r2; };   // tInv rows; .w: image index, (minY | maxY<<16), (minZ | maxZ<<16)

// LDS-DMA used as a prefetcher: the dword lands in a scratch LDS row (never read); what matters is
// that the cache line is on its way to L2/L1 long before the dense pass gathers from it.
__device__ __forceinline__ void d_prefetch(const void *g, void *ldsWaveRow)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)ldsWaveRow, 4, 0, 0);
}
template <bool HAS_CTF, bool SMALLBLOB>
__global__ void __launch_bounds__(512, 4)
k_rf_insert_tiles(const XhSpace *__restrict__ spaces, const float4 *__restrict__ cullN,
                  const float4 *__restrict__ cullX, int nspaces, const void *__restrict__ pk,
                  const float *__restrict__ blobTable, float *__restrict__ tempV, float *__restrict__ tempW,
                  int mv, float iDeltaSqrt, double blobRadius, const unsigned *__restrict__ tileList,
                  const int *__restrict__ classOff, int *__restrict__ counter, int dbg,
                  const int *__restrict__ superList, const int *__restrict__ superCount, int superDim, int superCap)
{
    __shared__ float sBlob[XH_BLOB_TABLE];
    __shared__ XhHitRec sRec[XH_CHUNK];
    __shared__ int sHit[XH_CHUNK];
    __shared__ int sWaveCnt[8];
    __shared__ int sSegStart[8][XH_SEGCAP + 1];
    __shared__ unsigned long long sSegMask[8][XH_SEGCAP + 1];
    __shared__ float qIx[8][XH_QCAP], qIy[8][XH_QCAP], qZs[8][XH_QCAP];
    __shared__ int qMeta[8][XH_QCAP];
    const int tid = threadIdx.x;
    for (int i = tid; i < XH_BLOB_TABLE; i += 512) sBlob[i] = blobTable[i];
    __syncthreads();
    const int sizeX = mv / 2, sizeY = mv, dim = mv + 1;
    const float fr = (float)blobRadius;
    const float maxDistanceSqr = (sizeX + blobRadius) * (sizeX + blobRadius);
    const float radiusSqr = blobRadius * blobRadius;
    const float rho8 = 6.1f;                      // half diagonal of the 7^3 voxel-centre cube + slack
    const float rho4 = 2.65f;                     // same for a 4^3 sub-cube (1.5*sqrt(3) = 2.598)
    const int lane = tid & 63, wv = tid >> 6;
    const int ox = (wv & 1) * 4, oy = ((wv >> 1) & 1) * 4, oz = (wv >> 2) * 4;
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    float accW = 0.f, accR = 0.f, accI = 0.f;   // lane l accumulates voxel l of the wave's sub-cube, in registers
    int nseg = 0;                              // wave-uniform: projections ("segments") with items in the queue

    // dense pass over queue items [0, n): lane l handles item l
    // The items' sums are then handed to the lanes that own the voxels: every item writes (w, re, im) over its
    // own queue slot, and lane l, for every segment of the batch whose ballot mask has bit l set, picks up
    // slot start + popcount(mask below l). LDS float atomics (ds_add_f32) did this before at ~1 lane per
    // clock -- a quarter of the kernel's time; plain LDS writes/reads are 20x cheaper, the accumulation order
    // per voxel (projection order) is unchanged and the accumulators stay in registers.
    auto process = [&](int n, int ty0, int tz0) {
        if (dbg == 1) return;
        float vW = 0.f, vR = 0.f, vI = 0.f;
        if (lane < n) {
            const int meta = qMeta[wv][lane];
            const int vl = meta & 63, si = meta >> 6;
            const float ix = qIx[wv][lane], iy = qIy[wv][lane], zSqr = qZs[wv][lane];
            const XhSpace &S = spaces[si];
            const int y = ty0 + ((vl >> 2) & 3), z = tz0 + (vl >> 4);
            // the reference only visits rows that cross the top or bottom face of the slab (RFA:746-750)
            float xa, xb;
            const bool hit1 = d_getX(xa, (float)y, (float)z, S.u, S.v, S.p0);
            const bool hit2 = d_getX(xb, (float)y, (float)z, S.u, S.v, S.p4);
            if ((hit1 || hit2) && dbg != 3) {
                int minX = (int)ceil((double)ix - blobRadius);
                int maxX = (int)floor((double)ix + blobRadius);
                int minY = (int)ceil((double)iy - blobRadius);
                int maxY = (int)floor((double)iy + blobRadius);
                minX = max(minX, 0);
                minY = max(minY, 0);
                maxX = min(maxX, sizeX - 1);
                maxY = min(maxY, sizeY - 1);
                const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;
                const size_t imgOff = (size_t)S.img * SX * SY;
                const float dataWeight = S.weight;
                if (SMALLBLOB) {
                    // blob radius < 2: at most 4x4 candidate pixels, fetched as four contiguous row
                    // segments of the padded record (all loads issued before any arithmetic). Pixels
                    // outside the blob or the image get weight 0, which leaves the sums bit-identical
                    // to the reference's "continue" (RFA:660-698).
                    const int bY = (int)ceil((double)iy - blobRadius), bX = (int)ceil((double)ix - blobRadius);
                    float yz[4], xs[4];
                    bool rv[4], cv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int i = bY + a;
                        const float ySqr = (iy - i) * (iy - i);
                        yz[a] = ySqr + zSqr;
                        rv[a] = (i >= minY) && (i <= maxY) && !(yz[a] > radiusSqr);
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int j = bX + b;
                        const float xD = ix - j;
                        xs[b] = xD * xD;
                        cv[b] = (j >= minX) && (j <= maxX);
                    }
                    const size_t base = imgOff + (size_t)(bY + XH_PAD) * SX + (bX + XH_PAD);
                    // rows are fetched RB at a time, all loads of a group before its arithmetic. With the CTF
                    // planes a pixel is 16 bytes: four rows at once would need 64 live registers and spill
                    // (the kernel is capped at 128 VGPRs by its LDS-limited occupancy), two rows do not.
                    constexpr int RB = (HAS_CTF && XH_CTF_ROWS2) ? 2 : 4;
#pragma unroll
                    for (int a0 = 0; a0 < 4; a0 += RB) {
                        float pr[RB][4], pi_[RB][4], wc[RB][4], wm[RB][4];
#pragma unroll
                        for (int ar = 0; ar < RB; ++ar) {
                            const int a = a0 + ar;
                            // rows outside the blob are not fetched at all (exec-masked): ~1/4 fewer L1 lookups
#pragma unroll
                            for (int b = 0; b < 4; ++b) { pr[ar][b] = 0.f; pi_[ar][b] = 0.f; wc[ar][b] = 0.f; wm[ar][b] = 0.f; }
                            if (!rv[a] && dbg != 4) continue;
                            if (HAS_CTF) {
                                const float4 *row = reinterpret_cast<const float4 *>(pk) + base + (size_t)a * SX;
#pragma unroll
                                for (int b = 0; b < 4; ++b) { const float4 q = row[b]; pr[ar][b] = q.x; pi_[ar][b] = q.y; wc[ar][b] = q.z; wm[ar][b] = q.w; }
                            } else {
                                const float2 *row = reinterpret_cast<const float2 *>(pk) + base + (size_t)a * SX;
#pragma unroll
                                for (int b = 0; b < 4; ++b) { const float2 q = row[b]; pr[ar][b] = q.x; pi_[ar][b] = q.y; }
                            }
                        }
#pragma unroll
                        for (int ar = 0; ar < RB; ++ar) {
                            const int a = a0 + ar;
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float distanceSqr = xs[b] + yz[a];
                                const bool use = rv[a] && cv[b] && !(distanceSqr > radiusSqr);
                                const int aux = use ? (int)(distanceSqr * iDeltaSqrt + 0.5f) : 0;
                                const float wBlob = use ? sBlob[aux] : 0.f;
                                if (HAS_CTF) {
                                    const float weight = wBlob * wm[ar][b] * dataWeight;
                                    vW += weight;
                                    vR += pr[ar][b] * weight * wc[ar][b];
                                    vI += pi_[ar][b] * weight * wc[ar][b];
                                } else {
                                    const float weight = wBlob * dataWeight;
                                    vW += weight;
                                    vR += pr[ar][b] * weight;
                                    vI += pi_[ar][b] * weight;
                                }
                            }
                        }
                    }
                } else
                for (int i = minY; i <= maxY; i++) {
                    const float ySqr = (iy - i) * (iy - i);
                    const float yzSqr = ySqr + zSqr;
                    if (yzSqr > radiusSqr) continue;
                    for (int j = minX; j <= maxX; j++) {
                        const float xD = ix - j;
                        const float distanceSqr = xD * xD + yzSqr;
                        if (distanceSqr > radiusSqr) continue;
                        const int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                        const float wBlob = sBlob[aux];
                        const size_t o = imgOff + (size_t)(i + XH_PAD) * SX + (j + XH_PAD);
                        if (HAS_CTF) {
                            const float4 q = reinterpret_cast<const float4 *>(pk)[o];
                            const float weight = wBlob * q.w * dataWeight;
                            vW += weight;
                            vR += q.x * weight * q.z;
                            vI += q.y * weight * q.z;
                        } else {
                            const float2 q = reinterpret_cast<const float2 *>(pk)[o];
                            const float weight = wBlob * dataWeight;
                            vW += weight;
                            vR += q.x * weight;
                            vI += q.y * weight;
                        }
                    }
                }
            }
        }
        if (lane < n) { qIx[wv][lane] = vW; qIy[wv][lane] = vR; qZs[wv][lane] = vI; }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int sg = 0; sg < nseg; ++sg) {
            const unsigned long long mask = sSegMask[wv][sg];
            if ((mask >> lane) & 1ull) {
                const int pos = sSegStart[wv][sg] + __popcll(mask & below);
                if (pos >= 0 && pos < n) { accW += qIx[wv][pos]; accR += qIy[wv][pos]; accI += qZs[wv][pos]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // Work distribution: tiles are queued per XCD class (contiguous z-slab of equal expected work).
    // A block drains its own class first (L2 affinity), then steals from the others, in chunks of
    // XH_GRAB tiles per atomic; the next chunk is requested while the current one is processed, so the
    // dequeue latency never sits between tiles, and an anisotropic orientation distribution (all
    // half-planes in one hemisphere) cannot idle half of the chip.
    __shared__ int sGrab[2];
    int cls = blockIdx.x & 7, tried = 0;
    int pend = 0;                                  // thread 0: outstanding dequeue (index into class cls)
    if (tid == 0) pend = atomicAdd(&counter[cls], XH_GRAB);
    for (;;) {
      if (tid == 0) {
        int lo = 0, hiT = 0;
        while (tried < 8) {
            const int n = classOff[cls + 1] - classOff[cls];
            if (pend < n) { lo = classOff[cls] + pend; hiT = classOff[cls] + min(pend + XH_GRAB, n); break; }
            cls = (cls + 1) & 7;                    // class drained: steal from the next one
            if (++tried < 8) pend = atomicAdd(&counter[cls], XH_GRAB);
        }
        sGrab[0] = lo; sGrab[1] = hiT;
        if (tried < 8) pend = atomicAdd(&counter[cls], XH_GRAB);   // in flight while this chunk is processed
      }
      __syncthreads();
      const int tBeg = sGrab[0], tEnd = sGrab[1];
      if (tBeg >= tEnd) break;
      for (int ti = tBeg; ti < tEnd; ++ti) {
        const unsigned packed = tileList[ti];
        const int tx = packed & 0xff, ty = (packed >> 8) & 0xff, tz = (packed >> 16) & 0xff;
        const int x0 = tx * XH_TSZ + ox, y0 = ty * XH_TSZ + oy, z0 = tz * XH_TSZ + oz;
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        const bool inVol = (x <= mv) && (y <= mv) && (z <= mv);
        const float cx = tx * XH_TSZ + 3.5f - mv / 2, cy = ty * XH_TSZ + 3.5f - mv / 2, cz = tz * XH_TSZ + 3.5f - mv / 2;
        const float c4x = x0 + 1.5f - mv / 2, c4y = y0 + 1.5f - mv / 2, c4z = z0 + 1.5f - mv / 2;
        const float px = x - mv / 2, py = y - mv / 2, pz = z - mv / 2;
        const bool inSphere = inVol && !((px * px + py * py + pz * pz) > maxDistanceSqr);
        accW = 0.f; accR = 0.f; accI = 0.f;
        nseg = 0;
        int qn = 0;   // wave-uniform queue length
        // the tile only looks at the projections that reach its super-tile (k_rf_supercull), in launch order
        const int sup = superList ? ((tz >> XH_SUPERSHIFT) * superDim + (ty >> XH_SUPERSHIFT)) * superDim + (tx >> XH_SUPERSHIFT) : 0;
        const int nlist = superList ? superCount[sup] : nspaces;
        const int *lst = superList ? superList + (size_t)sup * superCap : nullptr;
        for (int s0 = 0; s0 < nlist; s0 += XH_CHUNK) {
            const int li = s0 + tid;
            int s = -1;
            bool hit = false;
            if (tid < XH_CHUNK && li < nlist) {
                s = lst ? lst[li] : li;
                const float4 n = cullN[s], r0 = cullX[s];
                const float dn = n.x * cx + n.y * cy + n.z * cz;
                const float dx = r0.x * cx + r0.y * cy + r0.z * cz;
                // box bound of the tile (centres within +-3.5 per axis), never wider than the sphere bound
                const float hn = fminf(rho8, 3.5f * n.w + 0.02f), hx = fminf(rho8, 3.5f * r0.w + 0.02f);
                hit = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) sWaveCnt[wv] = __popcll(bal);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < XH_CHUNK / 64; ++w) { const int c = sWaveCnt[w]; if (w < wv) base += c; total += c; }
            if (hit) {
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                const XhSpace &S = spaces[s];
                sHit[pos] = s;
                XhHitRec r;
                r.r0 = make_float4(S.tInv[0], S.tInv[1], S.tInv[2], __int_as_float(S.img));
                r.r1 = make_float4(S.tInv[3], S.tInv[4], S.tInv[5], __int_as_float(S.minY | (S.maxY << 16)));
                r.r2 = make_float4(S.tInv[6], S.tInv[7], S.tInv[8], __int_as_float(S.minZ | (S.maxZ << 16)));
                sRec[pos] = r;
            }
            __syncthreads();
            // wave-level cull against the 4^3 sub-cube, 64 survivors at a time (lane <-> survivor), then only
            // the sub-cube's own survivors are visited; the next survivor's record is fetched from LDS while
            // the current one is tested, so its latency no longer sits between iterations
            for (int hb = 0; hb < (dbg == 2 ? 0 : total); hb += 64) {
                bool keep = false;
                if (hb + lane < total) {
                    const float4 r2 = sRec[hb + lane].r2, r0 = sRec[hb + lane].r0;
                    const float dn = r2.x * c4x + r2.y * c4y + r2.z * c4z;
                    const float dx = r0.x * c4x + r0.y * c4y + r0.z * c4z;
                    // box bound of the sub-cube (centres within +-1.5 per axis), never wider than the sphere bound
                    const float hn = fminf(rho4, 1.5f * (fabsf(r2.x) + fabsf(r2.y) + fabsf(r2.z)) + 0.02f);
                    const float hx = fminf(rho4, 1.5f * (fabsf(r0.x) + fabsf(r0.y) + fabsf(r0.z)) + 0.02f);
                    keep = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
                }
                unsigned long long todo = __ballot(keep);
                if (!todo) continue;
                int h = hb + __builtin_ctzll(todo);
                todo &= todo - 1;
                // the records are wave-uniform: in the CTF variant they live in scalar registers (see d_uniform)
                float4 r0 = d_uniform<HAS_CTF>(sRec[h].r0), r1 = d_uniform<HAS_CTF>(sRec[h].r1), r2 = d_uniform<HAS_CTF>(sRec[h].r2);
                int hitId = __builtin_amdgcn_readfirstlane(sHit[h]);
                for (;;) {
                    const bool more = todo != 0;
                    float4 n0 = r0, n1 = r1, n2 = r2;
                    int nHit = hitId;
                    int hn = h;
                    if (more) {
                        hn = hb + __builtin_ctzll(todo);
                        todo &= todo - 1;
                        n0 = d_uniform<HAS_CTF>(sRec[hn].r0); n1 = d_uniform<HAS_CTF>(sRec[hn].r1); n2 = d_uniform<HAS_CTF>(sRec[hn].r2);
                        nHit = __builtin_amdgcn_readfirstlane(sHit[hn]);
                    }
                    const int yy = __float_as_int(r1.w), zz = __float_as_int(r2.w);
                    bool pass = inSphere && !(y < (yy & 0xffff) || y > (yy >> 16) || z < (zz & 0xffff) || z > (zz >> 16));
                    float ix = 0.f, iy = 0.f, zSqr = 0.f;
                    if (pass) {
                        ix = r0.x * px + r0.y * py + r0.z * pz;
                        iy = r1.x * px + r1.y * py + r1.z * pz;
                        const float iz = r2.x * px + r2.y * py + r2.z * pz;
                        iy += mv / 2;
                        zSqr = iz * iz;
                        pass = !(zSqr > radiusSqr);
                        // a voxel with no pixel within reach adds nothing: drop it before the costly part
                        pass = pass && ((double)ix + blobRadius >= 0.0) && ((double)ix - blobRadius <= (double)(sizeX - 1)) &&
                               ((double)iy + blobRadius >= 0.0) && ((double)iy - blobRadius <= (double)(sizeY - 1));
                    }
                    const unsigned long long pb = __ballot(pass);
                    const int np = __popcll(pb);
                    if (np != 0) {
                        if (pass) {
                            const int q = qn + __popcll(pb & ((1ull << lane) - 1ull));
                            qIx[wv][q] = ix; qIy[wv][q] = iy; qZs[wv][q] = zSqr;
                            qMeta[wv][q] = (hitId << 6) | lane;
                        }
                        if (lane == 0) { sSegStart[wv][nseg] = qn; sSegMask[wv][nseg] = pb; }
                        ++nseg;
                        qn += np;
                        if (qn >= 64 || nseg == XH_SEGCAP) {
                            const int take = min(qn, 64);
                            process(take, y0, z0);
                            // move the remainder down (source index >= 64 > destination); only the last segment
                            // can straddle the batch boundary: it stays, 64 slots further down
                            const int rem = qn - take;
                            float a = 0.f, b = 0.f, c = 0.f;
                            int m = 0;
                            if (lane < rem) { a = qIx[wv][64 + lane]; b = qIy[wv][64 + lane]; c = qZs[wv][64 + lane]; m = qMeta[wv][64 + lane]; }
                            if (lane < rem) { qIx[wv][lane] = a; qIy[wv][lane] = b; qZs[wv][lane] = c; qMeta[wv][lane] = m; }
                            if (rem > 0) {
                                const int st = sSegStart[wv][nseg - 1] - 64;
                                const unsigned long long mk = sSegMask[wv][nseg - 1];
                                __builtin_amdgcn_wave_barrier();
                                if (lane == 0) { sSegStart[wv][0] = st; sSegMask[wv][0] = mk; }
                                nseg = 1;
                            } else nseg = 0;
                            qn = rem;
                            // the prefetched record was dead weight across the dense pass (register pressure):
                            // fetch it again instead of keeping it live
                            if (more) { n0 = d_uniform<HAS_CTF>(sRec[hn].r0); n1 = d_uniform<HAS_CTF>(sRec[hn].r1); n2 = d_uniform<HAS_CTF>(sRec[hn].r2); nHit = __builtin_amdgcn_readfirstlane(sHit[hn]); }
                        }
                    }
                    if (!more) break;
                    h = hn;
                    r0 = n0; r1 = n1; r2 = n2; hitId = nHit;
                }
            }
            __syncthreads();
        }
        if (qn > 0) process(qn, y0, z0);
        const float aW = accW, aR = accR, aI = accI;
        if (inSphere && (aW != 0.f || aR != 0.f || aI != 0.f)) {
            const size_t vi = ((size_t)z * dim + y) * dim + x;
            float2 *V = reinterpret_cast<float2 *>(tempV) + vi;
            float2 v = *V;
            v.x += aR;
            v.y += aI;
            *V = v;
            tempW[vi] += aW;
        }
      }
      __syncthreads();   // sGrab is rewritten at the top of the loop
    }
}


// ---- gridding, output-stationary, wave-independent form ------------------------------------------------
// Same arithmetic as k_rf_insert_tiles (its dense pass is reused verbatim), different control: a WAVE owns a 4x4x4
// sub-cube from start to finish. It culls its super-tile's projection list against the sub-cube itself (lane <->
// projection, 64 at a time), fetches the survivors' records with scalar loads (the index is wave-uniform), runs the
// sparse and dense passes and writes its 64 voxels. Nothing is shared between the waves of a workgroup but the blob
// table: no block-level cull, no record staging, no barrier inside the work loop -- in the tile kernel every chunk of
// the list ended in a barrier at which seven waves waited for the slowest sub-cube. Work units (sub-cubes, eight per
// tile of the tile list, same XCD classes) are handed out per wave, the next grab in flight while a unit is processed.
struct XhRec { float4 r0, r1, r2; };     // XhHitRec layout, one per traverse space, in global memory
// what the dense pass needs of a traverse space, compact (three loads per item instead of ten dwords scattered over
// the 100-byte XhSpace): (u.y, u.z, v.y, v.z), (p0.y, p0.z, p4.y, p4.z) and (weight, image index)
struct XhDense { float4 a, b; };
#ifndef XH_CUBE_NW
#define XH_CUBE_NW 8      // waves per workgroup of the wave-independent kernel
#define XH_CUBE_WPS 4     // waves per SIMD the register allocation is sized for
#endif
template <bool HAS_CTF, bool SMALLBLOB>
__global__ void __launch_bounds__(64 * XH_CUBE_NW, XH_CUBE_WPS)
k_rf_insert_cubes(const XhSpace *__restrict__ spaces, const float4 *__restrict__ cullN, const float4 *__restrict__ cullX,
                  const XhRec *__restrict__ recs, int nspaces, const void *__restrict__ pk, const float *__restrict__ blobTable,
                  float *__restrict__ tempV, float *__restrict__ tempW, int mv, float iDeltaSqrt, double blobRadius,
                  const unsigned *__restrict__ tileList, const int *__restrict__ classOff, int *__restrict__ counter, int dbg,
                  const int *__restrict__ superList, const int *__restrict__ superCount, int superDim, int superCap,
                  const float4 *__restrict__ superN, const float4 *__restrict__ superX,
                  const XhDense *__restrict__ dense, const float2 *__restrict__ wimg, float4 reach)
{
    __shared__ float sBlob[XH_BLOB_TABLE];
    __shared__ int sSegStart[XH_CUBE_NW][XH_SEGCAP + 1];
    __shared__ unsigned long long sSegMask[XH_CUBE_NW][XH_SEGCAP + 1];
    __shared__ float qIx[XH_CUBE_NW][XH_QCAP], qIy[XH_CUBE_NW][XH_QCAP], qZs[XH_CUBE_NW][XH_QCAP];
    __shared__ int qMeta[XH_CUBE_NW][XH_QCAP];
    const int tid = threadIdx.x;
    for (int i = tid; i < XH_BLOB_TABLE; i += 64 * XH_CUBE_NW) sBlob[i] = blobTable[i];
    __syncthreads();
    const int sizeX = mv / 2, sizeY = mv, dim = mv + 1;
    const float fr = (float)blobRadius;
    const float maxDistanceSqr = (sizeX + blobRadius) * (sizeX + blobRadius);
    const float radiusSqr = blobRadius * blobRadius;
    const float rho4 = 2.65f;                     // half diagonal of the 3^3 voxel-centre cube + slack
    const int lane = tid & 63, wv = tid >> 6;
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    float accW = 0.f, accR = 0.f, accI = 0.f;   // lane l accumulates voxel l of the wave's sub-cube, in registers
    int nseg = 0;                              // wave-uniform: projections ("segments") with items in the queue

    auto process = [&](int n, int ty0, int tz0) {
        if (dbg == 1) return;
        float vW = 0.f, vR = 0.f, vI = 0.f;
        if (lane < n) {
            const int meta = qMeta[wv][lane];
            const int vl = meta & 63, si = meta >> 6;
            const float ix = qIx[wv][lane], iy = qIy[wv][lane], zSqr = qZs[wv][lane];
            const XhDense dn = dense[si];
            const float2 wi = wimg[si];
            const int y = ty0 + ((vl >> 2) & 3), z = tz0 + (vl >> 4);
            // the reference only visits rows that cross the top or bottom face of the slab (RFA:746-750)
            const bool hit1 = d_hit((float)y, (float)z, dn.a.x, dn.a.y, dn.a.z, dn.a.w, dn.b.x, dn.b.y);
            const bool hit2 = d_hit((float)y, (float)z, dn.a.x, dn.a.y, dn.a.z, dn.a.w, dn.b.z, dn.b.w);
            if ((hit1 || hit2) && dbg != 3) {
                int minX = (int)ceil((double)ix - blobRadius);
                int maxX = (int)floor((double)ix + blobRadius);
                int minY = (int)ceil((double)iy - blobRadius);
                int maxY = (int)floor((double)iy + blobRadius);
                minX = max(minX, 0);
                minY = max(minY, 0);
                maxX = min(maxX, sizeX - 1);
                maxY = min(maxY, sizeY - 1);
                const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;
                const size_t imgOff = (size_t)__float_as_int(wi.y) * SX * SY;
                const float dataWeight = wi.x;
                if (SMALLBLOB) {
                    // blob radius < 2: at most 4x4 candidate pixels, fetched as four contiguous row
                    // segments of the padded record (all loads issued before any arithmetic). Pixels
                    // outside the blob or the image get weight 0, which leaves the sums bit-identical
                    // to the reference's "continue" (RFA:660-698).
                    const int bY = (int)ceil((double)iy - blobRadius), bX = (int)ceil((double)ix - blobRadius);
                    float yz[4], xs[4];
                    bool rv[4], cv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int i = bY + a;
                        const float ySqr = (iy - i) * (iy - i);
                        yz[a] = ySqr + zSqr;
                        rv[a] = (i >= minY) && (i <= maxY) && !(yz[a] > radiusSqr);
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int j = bX + b;
                        const float xD = ix - j;
                        xs[b] = xD * xD;
                        cv[b] = (j >= minX) && (j <= maxX);
                    }
                    const size_t base = imgOff + (size_t)(bY + XH_PAD) * SX + (bX + XH_PAD);
                    // rows are fetched RB at a time, all loads of a group before its arithmetic. With the CTF
                    // planes a pixel is 16 bytes: four rows at once would need 64 live registers and spill
                    // (the kernel is capped at 128 VGPRs by its LDS-limited occupancy), two rows do not.
                    constexpr int RB = (HAS_CTF && XH_CTF_ROWS2) ? 2 : 4;
#pragma unroll
                    for (int a0 = 0; a0 < 4; a0 += RB) {
                        float pr[RB][4], pi_[RB][4], wc[RB][4], wm[RB][4];
#pragma unroll
                        for (int ar = 0; ar < RB; ++ar) {
                            const int a = a0 + ar;
                            // rows outside the blob are not fetched at all (exec-masked): ~1/4 fewer L1 lookups
#pragma unroll
                            for (int b = 0; b < 4; ++b) { pr[ar][b] = 0.f; pi_[ar][b] = 0.f; wc[ar][b] = 0.f; wm[ar][b] = 0.f; }
                            if (!rv[a] && dbg != 4) continue;
                            if (HAS_CTF) {
                                const float4 *row = reinterpret_cast<const float4 *>(pk) + base + (size_t)a * SX;
#pragma unroll
                                for (int b = 0; b < 4; ++b) { const float4 q = row[b]; pr[ar][b] = q.x; pi_[ar][b] = q.y; wc[ar][b] = q.z; wm[ar][b] = q.w; }
                            } else {
                                const float2 *row = reinterpret_cast<const float2 *>(pk) + base + (size_t)a * SX;
#pragma unroll
                                for (int b = 0; b < 4; ++b) { const float2 q = row[b]; pr[ar][b] = q.x; pi_[ar][b] = q.y; }
                            }
                        }
#pragma unroll
                        for (int ar = 0; ar < RB; ++ar) {
                            const int a = a0 + ar;
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float distanceSqr = xs[b] + yz[a];
                                const bool use = rv[a] && cv[b] && !(distanceSqr > radiusSqr);
                                const int aux = use ? (int)(distanceSqr * iDeltaSqrt + 0.5f) : 0;
                                const float wBlob = use ? sBlob[aux] : 0.f;
                                if (HAS_CTF) {
                                    const float weight = wBlob * wm[ar][b] * dataWeight;
                                    vW += weight;
                                    vR += pr[ar][b] * weight * wc[ar][b];
                                    vI += pi_[ar][b] * weight * wc[ar][b];
                                } else {
                                    const float weight = wBlob * dataWeight;
                                    vW += weight;
                                    vR += pr[ar][b] * weight;
                                    vI += pi_[ar][b] * weight;
                                }
                            }
                        }
                    }
                } else
                for (int i = minY; i <= maxY; i++) {
                    const float ySqr = (iy - i) * (iy - i);
                    const float yzSqr = ySqr + zSqr;
                    if (yzSqr > radiusSqr) continue;
                    for (int j = minX; j <= maxX; j++) {
                        const float xD = ix - j;
                        const float distanceSqr = xD * xD + yzSqr;
                        if (distanceSqr > radiusSqr) continue;
                        const int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                        const float wBlob = sBlob[aux];
                        const size_t o = imgOff + (size_t)(i + XH_PAD) * SX + (j + XH_PAD);
                        if (HAS_CTF) {
                            const float4 q = reinterpret_cast<const float4 *>(pk)[o];
                            const float weight = wBlob * q.w * dataWeight;
                            vW += weight;
                            vR += q.x * weight * q.z;
                            vI += q.y * weight * q.z;
                        } else {
                            const float2 q = reinterpret_cast<const float2 *>(pk)[o];
                            const float weight = wBlob * dataWeight;
                            vW += weight;
                            vR += q.x * weight;
                            vI += q.y * weight;
                        }
                    }
                }
            }
        }
        if (lane < n) { qIx[wv][lane] = vW; qIy[wv][lane] = vR; qZs[wv][lane] = vI; }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int sg = 0; sg < nseg; ++sg) {
            const unsigned long long mask = sSegMask[wv][sg];
            if ((mask >> lane) & 1ull) {
                const int pos = sSegStart[wv][sg] + __popcll(mask & below);
                if (pos >= 0 && pos < n) { accW += qIx[wv][pos]; accR += qIy[wv][pos]; accI += qZs[wv][pos]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };


    // work distribution: units = sub-cubes, eight per tile. Every XCD class of the tile list is dealt into NSUB
    // interleaved streams (tile t of the class belongs to stream t % NSUB), each with its own counter, and a wave takes
    // ONE unit per grab. Both matter (4096-projection launch, ms): one counter per class and 1 / 2 / 4 / 8 / 16 units
    // per grab: 75.5 / 50.3 / 46.7 / 53.7 / 71 -- same-address atomics serialise in the L2, and the larger the grab the
    // wider the band of tiles the chip works on at any moment, while the projections' patches are shared through the
    // L2. With 8 streams per class (64 waves per counter): 38.4 / 42.7 / 46.4 at 1 / 2 / 4 units per grab (4 or 16
    // streams: 39.4 / 39.1). A wave drains its block's stream first, then steals from the following ones.
    constexpr int NSUB = 8, RS = 8;
    // Within a workgroup the eight waves share a ring of tiles: one global grab per tile (not per sub-cube), and the
    // eight sub-cubes of a tile are worked on by the eight waves at the same time -- the tightest band the L2 can get --
    // without any barrier: a wave draws a ticket (LDS atomic) = (tile sequence, sub-cube); whoever draws sub-cube 0 of
    // sequence q first fetches the tile of sequence q + 2 into its ring slot; a wave that is early moves on to the next
    // tile's sub-cubes. Slot q % RS is rewritten for q + RS only after 48 more tickets have been drawn.
    __shared__ int sTile[RS], sReady[RS];
    __shared__ int sTicket, sHop;
    const int home = (blockIdx.x & 7) * NSUB + ((blockIdx.x >> 3) & (NSUB - 1));
    auto streamTiles = [&](int st) {
        const int c = st / NSUB, j = st % NSUB, nt = classOff[c + 1] - classOff[c];
        return nt > j ? (nt - j + NSUB - 1) / NSUB : 0;
    };
    auto produce = [&](int q) {          // one lane: the tile of block sequence q, -1 when every stream is drained
        // sequences are produced in order (wait for q - 1): after the first -1 every later sequence is -1 too, so a wave
        // may leave at the first -1 it meets without stranding a tile that a concurrent producer still found
        while (q > 0 && atomicAdd(&sReady[(q - 1) % RS], 0) != q) __builtin_amdgcn_s_sleep(1);
        int tile = -1;
        for (;;) {
            const int hop = atomicAdd(&sHop, 0);
            if (hop >= 8 * NSUB) break;
            const int st = (home + hop) % (8 * NSUB);
            const int k = atomicAdd(&counter[st], 1);
            if (k < streamTiles(st)) { tile = (int)tileList[classOff[st / NSUB] + st % NSUB + k * NSUB]; break; }
            atomicMax(&sHop, hop + 1);
        }
        sTile[q % RS] = tile;
        __threadfence_block();
        atomicExch(&sReady[q % RS], q + 1);
    };
    if (tid == 0) {
        sTicket = 0; sHop = 0;
        for (int i = 0; i < RS; ++i) sReady[i] = 0;
        produce(0);
        produce(1);
    }
    __syncthreads();
    for (;;) {
        int t = 0, tileP = 0;
        if (lane == 0) {
            t = atomicAdd(&sTicket, 1);
            const int q = t >> 3;
            if ((t & 7) == 0) produce(q + 2);
            while (atomicAdd(&sReady[q % RS], 0) != q + 1) __builtin_amdgcn_s_sleep(1);
            tileP = atomicAdd(&sTile[q % RS], 0);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        tileP = __builtin_amdgcn_readfirstlane(tileP);
        if (tileP < 0) break;
        {
            const unsigned packed = (unsigned)tileP;
            const int u = t;
            const int sub = u & 7;
            const int tx = packed & 0xff, ty = (packed >> 8) & 0xff, tz = (packed >> 16) & 0xff;
            const int x0 = tx * XH_TSZ + (sub & 1) * 4, y0 = ty * XH_TSZ + ((sub >> 1) & 1) * 4, z0 = tz * XH_TSZ + (sub >> 2) * 4;
            const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
            const bool inVol = (x <= mv) && (y <= mv) && (z <= mv);
            const float c4x = x0 + 1.5f - mv / 2, c4y = y0 + 1.5f - mv / 2, c4z = z0 + 1.5f - mv / 2;
            const float px = x - mv / 2, py = y - mv / 2, pz = z - mv / 2;
            const bool inSphere = inVol && !((px * px + py * py + pz * pz) > maxDistanceSqr);
            if (!__ballot(inSphere)) continue;
            accW = 0.f; accR = 0.f; accI = 0.f;
            nseg = 0;
            int qn = 0;   // wave-uniform queue length
            const int sup = superList ? ((tz >> XH_SUPERSHIFT) * superDim + (ty >> XH_SUPERSHIFT)) * superDim + (tx >> XH_SUPERSHIFT) : 0;
            const int nlist = dbg == 5 ? 0 : (superList ? superCount[sup] : nspaces);
            const int *lst = superList ? superList + (size_t)sup * superCap : nullptr;
            const float4 *lstN = superList ? superN + (size_t)sup * superCap : cullN;
            const float4 *lstX = superList ? superX + (size_t)sup * superCap : cullX;
            // the list entries of the next 64 projections are requested before the current ones are worked on
            int sNext = 0;
            float4 nNext = make_float4(0.f, 0.f, 0.f, 0.f), xNext = nNext;
            if (lane < nlist) { sNext = lst ? lst[lane] : lane; nNext = lstN[lane]; xNext = lstX[lane]; }
            for (int hb = 0; hb < nlist; hb += 64) {
                // cull against the sub-cube: lane <-> projection of the list
                bool keep = false;
                const int sIdx = sNext;
                const float4 n = nNext, xv = xNext;
                if (hb + 64 + lane < nlist) { sNext = lst ? lst[hb + 64 + lane] : hb + 64 + lane; nNext = lstN[hb + 64 + lane]; xNext = lstX[hb + 64 + lane]; }
                if (hb + lane < nlist) {
                    const float dn = n.x * c4x + n.y * c4y + n.z * c4z;
                    const float dx = xv.x * c4x + xv.y * c4y + xv.z * c4z;
                    const float hn = fminf(rho4, 1.5f * n.w + 0.02f), hx = fminf(rho4, 1.5f * xv.w + 0.02f);
                    keep = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
                }
                unsigned long long todo = __ballot(keep);
                if (!todo || dbg == 2) continue;
                // the records are wave-uniform: scalar loads, the next one in flight while the current one is tested
                int hitId = __builtin_amdgcn_readlane(sIdx, __builtin_ctzll(todo));
                todo &= todo - 1;
                float4 r0 = recs[hitId].r0, r1 = recs[hitId].r1, r2 = recs[hitId].r2;
                for (;;) {
                    const bool more = todo != 0;
                    float4 n0 = r0, n1 = r1, n2 = r2;
                    int nHit = hitId;
                    if (more) {
                        nHit = __builtin_amdgcn_readlane(sIdx, __builtin_ctzll(todo));
                        todo &= todo - 1;
                        n0 = recs[nHit].r0; n1 = recs[nHit].r1; n2 = recs[nHit].r2;
                    }
                    const int yy = __float_as_int(r1.w), zz = __float_as_int(r2.w);
                    bool pass = inSphere && !(y < (yy & 0xffff) || y > (yy >> 16) || z < (zz & 0xffff) || z > (zz >> 16));
                    float ix = 0.f, iy = 0.f, zSqr = 0.f;
                    if (pass) {
                        ix = r0.x * px + r0.y * py + r0.z * pz;
                        iy = r1.x * px + r1.y * py + r1.z * pz;
                        const float iz = r2.x * px + r2.y * py + r2.z * pz;
                        iy += mv / 2;
                        zSqr = iz * iz;
                        pass = !(zSqr > radiusSqr);
                        // a voxel with no pixel within reach adds nothing: drop it before the costly part. The four tests
                        //   (double)ix + r >= 0, (double)ix - r <= sizeX - 1, (double)iy + r >= 0, (double)iy - r <= sizeY - 1
                        // are monotone in the float ix / iy: the host found, with the same double expressions, the smallest
                        // and largest floats that pass (reach.x..w), so four float compares decide the same thing
                        pass = pass && (ix >= reach.x) && (ix <= reach.y) && (iy >= reach.z) && (iy <= reach.w);
                    }
                    const unsigned long long pb = __ballot(pass);
                    const int np = __popcll(pb);
                    if (np != 0) {
                        if (pass) {
                            const int q = qn + __popcll(pb & ((1ull << lane) - 1ull));
                            qIx[wv][q] = ix; qIy[wv][q] = iy; qZs[wv][q] = zSqr;
                            qMeta[wv][q] = (hitId << 6) | lane;
                        }
                        if (lane == 0) { sSegStart[wv][nseg] = qn; sSegMask[wv][nseg] = pb; }
                        ++nseg;
                        qn += np;
                        if (qn >= 64 || nseg == XH_SEGCAP) {
                            const int take = min(qn, 64);
                            process(take, y0, z0);
                            // move the remainder down; only the last segment can straddle the batch boundary
                            const int rem = qn - take;
                            float a = 0.f, b = 0.f, c = 0.f;
                            int m = 0;
                            if (lane < rem) { a = qIx[wv][64 + lane]; b = qIy[wv][64 + lane]; c = qZs[wv][64 + lane]; m = qMeta[wv][64 + lane]; }
                            if (lane < rem) { qIx[wv][lane] = a; qIy[wv][lane] = b; qZs[wv][lane] = c; qMeta[wv][lane] = m; }
                            if (rem > 0) {
                                const int st = sSegStart[wv][nseg - 1] - 64;
                                const unsigned long long mk = sSegMask[wv][nseg - 1];
                                __builtin_amdgcn_wave_barrier();
                                if (lane == 0) { sSegStart[wv][0] = st; sSegMask[wv][0] = mk; }
                                nseg = 1;
                            } else nseg = 0;
                            qn = rem;
                        }
                    }
                    if (!more) break;
                    r0 = n0; r1 = n1; r2 = n2; hitId = nHit;
                }
            }
            if (qn > 0) process(qn, y0, z0);
            const float aW = accW, aR = accR, aI = accI;
            if (inSphere && (aW != 0.f || aR != 0.f || aI != 0.f)) {
                const size_t vi = ((size_t)z * dim + y) * dim + x;
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


// ---- gridding, output-stationary, LDS-staged patches ---------------------------------------------
// The queue kernel above is bound by the L1's tag-lookup rate: every lane gathers its own 4x4 footprint
// (profiles/README.md: 26 L1 accesses per vector-memory instruction). Here a wave stages, per surviving
// projection, the bounding 12x12 patch of its 4x4x4 sub-cube's footprint into LDS with coalesced row
// loads (~10x fewer L1 accesses); every tap then comes from LDS. Lanes keep their voxel for the whole
// tile, so accumulation is in registers, in projection order: no LDS atomics, deterministic, and a
// single projection is bit-identical to processVoxelBlob (RFA:627-700). Blob radius < 2 only.
#define XH_PW 12          // patch width/height in pixels
template <bool HAS_CTF>
__global__ void __launch_bounds__(512, 4)
k_rf_insert_tiles_lds(const XhSpace *__restrict__ spaces, const float4 *__restrict__ cullN,
                      const float4 *__restrict__ cullX, int nspaces, const void *__restrict__ pk,
                      const float *__restrict__ blobTable, float *__restrict__ tempV, float *__restrict__ tempW,
                      int mv, float iDeltaSqrt, double blobRadius, const unsigned *__restrict__ tileList,
                      const int *__restrict__ classOff, int *__restrict__ counter)
{
    typedef typename std::conditional<HAS_CTF, float4, float2>::type Pix;
    __shared__ float sBlob[XH_BLOB_TABLE];
    __shared__ XhHitRec sRec[XH_CHUNK];
    __shared__ int sHit[XH_CHUNK];
    __shared__ int sWaveCnt[8];
    __shared__ int sGrab[2];
    __shared__ Pix sPatch[8][XH_PW * XH_PW];
    const int tid = threadIdx.x;
    for (int i = tid; i < XH_BLOB_TABLE; i += 512) sBlob[i] = blobTable[i];
    __syncthreads();
    const int sizeX = mv / 2, sizeY = mv, dim = mv + 1;
    const int SX = sizeX + 2 * XH_PAD, SY = sizeY + 2 * XH_PAD;
    const float fr = (float)blobRadius;
    const float maxDistanceSqr = (sizeX + blobRadius) * (sizeX + blobRadius);
    const float radiusSqr = blobRadius * blobRadius;
    const float rho8 = 6.1f, rho4 = 2.65f;
    const int lane = tid & 63, wv = tid >> 6;
    const int ox = (wv & 1) * 4, oy = ((wv >> 1) & 1) * 4, oz = (wv >> 2) * 4;
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    Pix *patch = sPatch[wv];
    const Pix *pkp = reinterpret_cast<const Pix *>(pk);

    int cls = blockIdx.x & 7, tried = 0;
    int pend = 0;
    if (tid == 0) pend = atomicAdd(&counter[cls], XH_GRAB);
    for (;;) {
      if (tid == 0) {
        int lo = 0, hiT = 0;
        while (tried < 8) {
            const int n = classOff[cls + 1] - classOff[cls];
            if (pend < n) { lo = classOff[cls] + pend; hiT = classOff[cls] + min(pend + XH_GRAB, n); break; }
            cls = (cls + 1) & 7;
            if (++tried < 8) pend = atomicAdd(&counter[cls], XH_GRAB);
        }
        sGrab[0] = lo; sGrab[1] = hiT;
        if (tried < 8) pend = atomicAdd(&counter[cls], XH_GRAB);
      }
      __syncthreads();
      const int tBeg = sGrab[0], tEnd = sGrab[1];
      if (tBeg >= tEnd) break;
      for (int ti = tBeg; ti < tEnd; ++ti) {
        const unsigned packed = tileList[ti];
        const int tx = packed & 0xff, ty = (packed >> 8) & 0xff, tz = (packed >> 16) & 0xff;
        const int x0 = tx * XH_TSZ + ox, y0 = ty * XH_TSZ + oy, z0 = tz * XH_TSZ + oz;
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        const bool inVol = (x <= mv) && (y <= mv) && (z <= mv);
        const float cx = tx * XH_TSZ + 3.5f - mv / 2, cy = ty * XH_TSZ + 3.5f - mv / 2, cz = tz * XH_TSZ + 3.5f - mv / 2;
        const float c4x = x0 + 1.5f - mv / 2, c4y = y0 + 1.5f - mv / 2, c4z = z0 + 1.5f - mv / 2;
        const float px = x - mv / 2, py = y - mv / 2, pz = z - mv / 2;
        const bool inSphere = inVol && !((px * px + py * py + pz * pz) > maxDistanceSqr);
        float accW = 0.f, accR = 0.f, accI = 0.f;
        for (int s0 = 0; s0 < nspaces; s0 += XH_CHUNK) {
            // ---- block level: cull XH_CHUNK projections against the tile, ordered compaction
            const int s = s0 + tid;
            bool hit = false;
            if (tid < XH_CHUNK && s < nspaces) {
                const float4 n = cullN[s], r0 = cullX[s];
                const float dn = n.x * cx + n.y * cy + n.z * cz;
                const float dx = r0.x * cx + r0.y * cy + r0.z * cz;
                // box bound of the tile (centres within +-3.5 per axis), never wider than the sphere bound
                const float hn = fminf(rho8, 3.5f * n.w + 0.02f), hx = fminf(rho8, 3.5f * r0.w + 0.02f);
                hit = (fabsf(dn) <= fr + hn) && (dx >= -(fr + hx)) && (dx <= sizeX + fr + hx);
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) sWaveCnt[wv] = __popcll(bal);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < XH_CHUNK / 64; ++w) { const int c = sWaveCnt[w]; if (w < wv) base += c; total += c; }
            if (hit) {
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                const XhSpace &S = spaces[s];
                sHit[pos] = s;
                XhHitRec r;
                r.r0 = make_float4(S.tInv[0], S.tInv[1], S.tInv[2], __int_as_float(S.img));
                r.r1 = make_float4(S.tInv[3], S.tInv[4], S.tInv[5], __int_as_float(S.minY | (S.maxY << 16)));
                r.r2 = make_float4(S.tInv[6], S.tInv[7], S.tInv[8], __int_as_float(S.minZ | (S.maxZ << 16)));
                sRec[pos] = r;
            }
            __syncthreads();
            // ---- wave level: lanes cull up to 64 survivors at a time against the 4^3 sub-cube
            for (int h0 = 0; h0 < total; h0 += 64) {
                bool sub = false;
                if (h0 + lane < total) {
                    const float4 r2 = sRec[h0 + lane].r2, r0 = sRec[h0 + lane].r0;
                    const float dn = r2.x * c4x + r2.y * c4y + r2.z * c4z;
                    const float dx = r0.x * c4x + r0.y * c4y + r0.z * c4z;
                    sub = (fabsf(dn) <= fr + rho4) && (dx >= -(fr + rho4)) && (dx <= sizeX + fr + rho4);
                }
                unsigned long long todo = __ballot(sub);
                while (todo) {
                    const int h = h0 + __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const float4 r0 = sRec[h].r0, r1 = sRec[h].r1, r2 = sRec[h].r2;
                    const int yy = __float_as_int(r1.w), zz = __float_as_int(r2.w);
                    bool pass = inSphere && !(y < (yy & 0xffff) || y > (yy >> 16) || z < (zz & 0xffff) || z > (zz >> 16));
                    float ix = 0.f, iy = 0.f, zSqr = 0.f;
                    if (pass) {
                        ix = r0.x * px + r0.y * py + r0.z * pz;
                        iy = r1.x * px + r1.y * py + r1.z * pz;
                        const float iz = r2.x * px + r2.y * py + r2.z * pz;
                        iy += mv / 2;
                        zSqr = iz * iz;
                        pass = !(zSqr > radiusSqr);
                        pass = pass && ((double)ix + blobRadius >= 0.0) && ((double)ix - blobRadius <= (double)(sizeX - 1)) &&
                               ((double)iy + blobRadius >= 0.0) && ((double)iy - blobRadius <= (double)(sizeY - 1));
                    }
                    if (!__any(pass)) continue;
                    // ---- stage the 12x12 patch that contains every candidate pixel of the sub-cube
                    const float a0 = fabsf(r0.x) + fabsf(r0.y) + fabsf(r0.z), a1 = fabsf(r1.x) + fabsf(r1.y) + fabsf(r1.z);
                    const float cix = r0.x * c4x + r0.y * c4y + r0.z * c4z;
                    const float ciy = r1.x * c4x + r1.y * c4y + r1.z * c4z + mv / 2;
                    const int bx0 = (int)floorf(cix - 1.5f * a0 - fr) - 1, by0 = (int)floorf(ciy - 1.5f * a1 - fr) - 1;
                    const size_t imgOff = (size_t)__float_as_int(r0.w) * SX * SY;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int c = lane + 64 * t;
                        if (c < XH_PW * XH_PW) {
                            const int qy = c / XH_PW, qx = c - qy * XH_PW;
                            const int j = bx0 + qx + XH_PAD, i = by0 + qy + XH_PAD;
                            Pix v;
                            memset(&v, 0, sizeof(v));
                            if (j >= 0 && j < SX && i >= 0 && i < SY) v = pkp[imgOff + (size_t)i * SX + j];
                            patch[c] = v;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (pass) {
                        const int si = sHit[h];
                        const XhSpace &S = spaces[si];
                        float xa, xb;
                        const bool hit1 = d_getX(xa, (float)y, (float)z, S.u, S.v, S.p0);
                        const bool hit2 = d_getX(xb, (float)y, (float)z, S.u, S.v, S.p4);
                        if (hit1 || hit2) {
                            int minX = (int)ceil((double)ix - blobRadius);
                            int maxX = (int)floor((double)ix + blobRadius);
                            int minY = (int)ceil((double)iy - blobRadius);
                            int maxY = (int)floor((double)iy + blobRadius);
                            const int bY = minY, bX = minX;
                            minX = max(minX, 0);
                            minY = max(minY, 0);
                            maxX = min(maxX, sizeX - 1);
                            maxY = min(maxY, sizeY - 1);
                            const float dataWeight = S.weight;
                            float yz[4], xs[4];
                            bool rv[4], cv[4];
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                const int i = bY + a;
                                const float ySqr = (iy - i) * (iy - i);
                                yz[a] = ySqr + zSqr;
                                rv[a] = (i >= minY) && (i <= maxY) && !(yz[a] > radiusSqr);
                            }
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const int j = bX + b;
                                const float xD = ix - j;
                                xs[b] = xD * xD;
                                cv[b] = (j >= minX) && (j <= maxX);
                            }
                            const Pix *pp = patch + (bY - by0) * XH_PW + (bX - bx0);
                            float vW = 0.f, vR = 0.f, vI = 0.f;
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const Pix q = pp[a * XH_PW + b];
                                    const float distanceSqr = xs[b] + yz[a];
                                    const bool use = rv[a] && cv[b] && !(distanceSqr > radiusSqr);
                                    const int aux = use ? (int)(distanceSqr * iDeltaSqrt + 0.5f) : 0;
                                    const float wBlob = use ? sBlob[aux] : 0.f;
                                    if constexpr (HAS_CTF) {
                                        const float weight = wBlob * q.w * dataWeight;
                                        vW += weight;
                                        vR += q.x * weight * q.z;
                                        vI += q.y * weight * q.z;
                                    } else {
                                        const float weight = wBlob * dataWeight;
                                        vW += weight;
                                        vR += q.x * weight;
                                        vI += q.y * weight;
                                    }
                                }
                            accW += vW;
                            accR += vR;
                            accI += vI;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
        }
        if (inSphere && (accW != 0.f || accR != 0.f || accI != 0.f)) {
            const size_t vi = ((size_t)z * dim + y) * dim + x;
            float2 *V = reinterpret_cast<float2 *>(tempV) + vi;
            float2 v = *V;
            v.x += accR;
            v.y += accI;
            *V = v;
            tempW[vi] += accW;
        }
      }
      __syncthreads();
    }
}

#endif
