// micro-benchmark: wave64 VALU issue rates on gfx950 (plain vs packed fp32), 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    v2f aa = {a, a}, bb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa), "v"(bb));
            }
        } else if (MODE == 2) {   // v_cvt_i32_f32 + v_cndmask mix (non-FMA plain ops)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %9\n v_min_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_floor_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (MODE == 3) {   // transcendental
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
// LDS random gather: ds_read_b32 from a 40 KB table at pseudo-random indices vs conflict-free
template <int MODE>
__global__ void __launch_bounds__(256) kl(float *out, int iters)
{
    __shared__ float tab[10240];
    for (int i = threadIdx.x; i < 10240; i += 256) tab[i] = i;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            s = s * 1664525u + 1013904223u;
            unsigned idx = MODE == 0 ? (s >> 8) % 10000u : ((s >> 8) % 156u) * 64u + (threadIdx.x & 63);
            acc += tab[idx];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    float *d; hipMalloc(&d, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;   // 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double inst = (double)blocks * 4 * iters * 64;   // wave-instructions
            const double perSimdPerSec = inst / (ms * 1e-3) / 1024.0;
            printf("waves/SIMD %d mode %d (%s): %.3f ms, %.2f G wave-inst/s/SIMD => %.2f cycles per wave-instruction at 2.4 GHz\n", wps, mode,
                   mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32" : mode == 2 ? "plain mix" : "v_exp_f32", ms, perSimdPerSec * 1e-9, 2.4e9 / perSimdPerSec);
        }
    }
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0;
            const int blocks = 256 * wps, it = 500;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(kl<0>, dim3(blocks), dim3(256), 0, 0, d, it);
                else hipLaunchKernelGGL(kl<1>, dim3(blocks), dim3(256), 0, 0, d, it);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double inst = (double)blocks * 4 * it * 16;
            printf("LDS gather waves/SIMD %d %s: %.3f ms, %.2f cycles per ds_read_b32 wave-instruction per CU\n", wps, mode == 0 ? "random" : "conflict-free", ms,
                   2.4e9 * ms * 1e-3 / (inst / 256.0));
        }
    return 0;
}
