// Do kernels from two streams share a CU?  A persistent kernel (one block per CU, LDS bytes and waves as given, spins for a fixed
// number of clock cycles) on stream 1, a light kernel (many small blocks, little work each) on stream 2 while it runs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_coresidency.hip -o /tmp/cores && /tmp/cores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_hog(long long cycles, float *out)
{
    extern __shared__ float lds[];
    const long long t0 = wall_clock64();
    float acc = 0.f;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += lds[(threadIdx.x + i) & 1023];
    }
    if (acc == 12345.f) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_light(const float *in, float *out, size_t n, int reps)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    for (int r = 0; r < reps; ++r) v = v * 1.0001f + 0.5f;
    out[i] = v;
}

// round 6: the same question with the real kernels' footprints.  k_rf_grid at eight waves per CU = 512 threads, 116 KB of LDS, 168
// registers; the matcher's S6 kernels = 256 threads, 37 KB of LDS, ~100 registers.  Registers are claimed by naming them in an asm clobber.
__global__ void __launch_bounds__(512) k_hog168(long long cycles, float *out)
{
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v167, 0" ::: "v167");
    const long long t0 = wall_clock64();
    float acc = 0.f;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += lds[(threadIdx.x + i) & 1023];
    }
    if (acc == 12345.f) out[0] = acc;
}
template <int VG>
__global__ void __launch_bounds__(256) k_mid(const float *in, float *out, size_t n, int reps)
{
    extern __shared__ float lds[];
    if (VG == 100) asm volatile("v_mov_b32 v99, 0" ::: "v99");
    if (VG == 170) asm volatile("v_mov_b32 v169, 0" ::: "v169");
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    lds[threadIdx.x] = in[i < n ? i : 0];
    __syncthreads();
    float v = lds[(threadIdx.x + 1) & 255];
    for (int r = 0; r < reps; ++r) v = v * 1.0001f + 0.5f;
    if (i < n) out[i] = v;
}

int main()
{
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    const size_t n = (size_t)64 << 20;
    float *a, *b, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, 1024);
    hipMemset(a, 0, n * 4);
    hipEvent_t e0, e1, h0, h1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&h0); hipEventCreate(&h1);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // wall_clock64 ticks at 100 MHz on gfx9: 2e6 ticks = 20 ms
    const long long spin = 2000000;
    struct Cfg { int threads; int ldsKB; };
    const Cfg cfgs[] = {{0, 0}, {768, 150}, {768, 100}, {512, 118}, {512, 64}, {256, 64}, {256, 0}, {768, 0}, {1024, 0}};
    for (const Cfg &c : cfgs) {
        hipFuncSetAttribute((const void *)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipDeviceSynchronize();
        if (c.threads) {
            hipEventRecord(h0, s1);
            hipLaunchKernelGGL(k_hog, dim3(cus), dim3(c.threads), (size_t)c.ldsKB * 1024 + 4096, s1, spin, o);
            hipEventRecord(h1, s1);
        }
        // give the hog a moment to occupy the device
        hipLaunchKernelGGL(k_light, dim3(1), dim3(256), 0, s2, a, b, (size_t)256, 1);
        hipEventRecord(e0, s2);
        hipLaunchKernelGGL(k_light, dim3((unsigned)(n / 256)), dim3(256), 0, s2, a, b, n, 64);
        hipEventRecord(e1, s2);
        hipDeviceSynchronize();
        float ml = 0, mh = 0;
        hipEventElapsedTime(&ml, e0, e1);
        if (c.threads) hipEventElapsedTime(&mh, h0, h1);
        printf("hog %4d threads %3d KB LDS per CU: hog %.2f ms, light kernel (64 M elements) %.3f ms\n", c.threads, c.ldsKB, mh, ml);
    }
    printf("\nround 6: hog = 512 threads, 116 KB of LDS, 168 registers (k_rf_grid at eight waves); second kernel 256 threads per block\n");
    hipFuncSetAttribute((const void *)k_hog168, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_mid<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute((const void *)k_mid<170>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute((const void *)k_mid<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    struct Mid { int vg; int ldsKB; };
    const Mid mids[] = {{0, 1}, {0, 37}, {100, 1}, {100, 37}, {100, 41}, {100, 47}, {170, 37}};
    for (int hog = 0; hog < 2; ++hog)
        for (const Mid &m : mids) {
            hipDeviceSynchronize();
            if (hog) {
                hipEventRecord(h0, s1);
                hipLaunchKernelGGL(k_hog168, dim3(cus), dim3(512), (size_t)116 * 1024, s1, spin, o);
                hipEventRecord(h1, s1);
            }
            hipLaunchKernelGGL(k_light, dim3(1), dim3(256), 0, s2, a, b, (size_t)256, 1);
            hipEventRecord(e0, s2);
            const size_t lds = (size_t)m.ldsKB * 1024;
            if (m.vg == 0) hipLaunchKernelGGL(k_mid<0>, dim3((unsigned)(n / 256)), dim3(256), lds, s2, a, b, n, 64);
            else if (m.vg == 100) hipLaunchKernelGGL(k_mid<100>, dim3((unsigned)(n / 256)), dim3(256), lds, s2, a, b, n, 64);
            else hipLaunchKernelGGL(k_mid<170>, dim3((unsigned)(n / 256)), dim3(256), lds, s2, a, b, n, 64);
            hipEventRecord(e1, s2);
            hipDeviceSynchronize();
            float ml = 0, mh = 0;
            hipEventElapsedTime(&ml, e0, e1);
            if (hog) hipEventElapsedTime(&mh, h0, h1);
            printf("%s second kernel %3d registers %2d KB LDS: %.3f ms%s\n", hog ? "beside the hog:" : "alone:         ", m.vg ? m.vg : 8, m.ldsKB, ml,
                   hog && ml > 15.f ? "   <- waited for the hog" : "");
        }
    return 0;
}
