// micro-benchmark: what the dense pass of the gridding kernel asks of the LDS of a gfx950 CU.  Twelve waves per CU (one workgroup of 768
// threads), every lane reads the sixteen taps of a 4 x 4 footprint at a lane-random position of its wave's patch, in the layouts
//   0  records of 16 bytes (re, im, w, -), row stride 21 records, ds_read_b128 per tap                          (the product)
//   1  records of 12 bytes (re, im, w), row stride 21 records, ds_read_b96 per tap
//   2  three planes of floats, row stride 21 floats, one ds_read_b128 per footprint row and plane (4-byte aligned only)
//   3  (re, im) records of 8 bytes + a plane of w: ds_read_b64 + ds_read_b32 per tap
//   4  sixteen ds_read_b32 of a 40 KB table at random entries (half of the lanes at one shared entry), the blob weights
//   5  0 + 4 together (what a dense batch issues)
//   6  1 + 4 together
//   7  2 + 4 together
//   8  5 with the taps beyond the blob (53 % of the lanes of every tap, lane-random) switched off for both reads
//   9  5 with those lanes reading one shared record / entry instead
//  10  0 with every lane on the patch's first record (no bank conflicts: what returning 1 KB per instruction costs)
//  11  10 as ds_read_b96 (16-byte aligned)      12  0 as ds_read_b96 of the 16-byte records      13  0 as ds_read_b64
// Addresses are formed once; the loop only reads and adds.  Prints cycles per batch (64 lanes x 16 taps) and CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o /tmp/ubench_lds && /tmp/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define LDSP(T, a) ((const __attribute__((address_space(3))) T *)(uintptr_t)(a))

template <int MODE>
__global__ void __launch_bounds__(768) k(float *out, int iters)
{
    __shared__ __align__(16) unsigned char lds[12 * 5120 + 40064];
    for (int i = threadIdx.x; i < (int)sizeof(lds) / 4; i += 768) reinterpret_cast<float *>(lds)[i] = (float)(i & 1023);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const unsigned patch = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)(lds + wv * 5120);
    const unsigned table = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)(lds + 12 * 5120);
    unsigned s = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    auto rnd = [&](unsigned m) { s = s * 1664525u + 1013904223u; return (s >> 10) % m; };
    const unsigned ry = rnd(12), cx = rnd(12);
    unsigned tIdx[16];
    for (int t = 0; t < 16; ++t) tIdx[t] = table + 4u * ((rnd(100) < 53) ? 10000u : rnd(10000));
    unsigned live = 0;
    for (int t = 0; t < 16; ++t) live |= (tIdx[t] != table + 40000u) << t;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 8) {
            v4f q[16];
            float w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                q[t] = (v4f){0.f, 0.f, 0.f, 0.f};
                if ((live >> t) & 1) q[t] = *LDSP(v4f, patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 16);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                w[t] = 0.f;
                if ((live >> t) & 1) w[t] = *LDSP(float, tIdx[t]);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x + w[t];
        }
        if (MODE == 9) {
            v4f q[16];
            float w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) q[t] = *LDSP(v4f, ((live >> t) & 1) ? patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 16 : patch);
#pragma unroll
            for (int t = 0; t < 16; ++t) w[t] = *LDSP(float, tIdx[t]);
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x + w[t];
        }
        if (MODE == 0 || MODE == 5) {
            v4f q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) q[t] = *LDSP(v4f, patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 16);
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x;
        }
        if (MODE == 10) {
            v4f q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { asm volatile("ds_read_b128 %0, %1" : "=v"(q[t]) : "v"(patch)); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x;
        }
        if (MODE == 11 || MODE == 12) {
            v3f q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned a = MODE == 11 ? patch : patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 16;
                asm volatile("ds_read_b96 %0, %1" : "=v"(q[t]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x;
        }
        if (MODE == 13) {
            v2f q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned a = patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 16;
                asm volatile("ds_read_b64 %0, %1" : "=v"(q[t]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x;
        }
        if (MODE == 1 || MODE == 6) {
            v3f q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned a = patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 12;
                asm volatile("ds_read_b96 %0, %1" : "=v"(q[t]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x;
        }
        if (MODE == 2 || MODE == 7) {
            v4f q[12];
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const unsigned a = patch + (t >> 2) * 1280 + ((ry + (t & 3)) * 21 + cx) * 4;
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[t]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 12; ++t) acc += q[t].x;
        }
        if (MODE == 3) {
            v2f q[16];
            float w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                q[t] = *LDSP(v2f, patch + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 8);
                w[t] = *LDSP(float, patch + 3072 + ((ry + (t >> 2)) * 21 + cx + (t & 3)) * 4);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += q[t].x + w[t];
        }
        if (MODE >= 4 && MODE <= 7) {
            float w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) w[t] = *LDSP(float, tIdx[t]);
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += w[t];
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <int MODE> static void run(float *d, const char *what)
{
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(768), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    // a CU runs 12 waves = 12 batches per iteration
    printf("mode %d  %-58s %.3f ms  %.1f cycles per batch and CU\n", MODE, what, ms, 2.4e9 * ms * 1e-3 / (iters * 12.0));
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 768 * 4);
    run<0>(d, "16 x ds_read_b128, 16-byte records");
    run<1>(d, "16 x ds_read_b96, 12-byte records");
    run<2>(d, "12 x ds_read_b128, planes, 4-byte aligned");
    run<3>(d, "16 x (ds_read_b64 + ds_read_b32)");
    run<4>(d, "16 x ds_read_b32 of the table");
    run<5>(d, "records of 16 bytes + table");
    run<6>(d, "records of 12 bytes + table");
    run<7>(d, "planes + table");
    run<8>(d, "records of 16 bytes + table, dead taps switched off");
    run<9>(d, "records of 16 bytes + table, dead taps on one record");
    run<10>(d, "16 x ds_read_b128, one record for all lanes");
    run<11>(d, "16 x ds_read_b96, one record for all lanes");
    run<12>(d, "16 x ds_read_b96 of 16-byte records");
    run<13>(d, "16 x ds_read_b64 of 16-byte records");
    return 0;
}
