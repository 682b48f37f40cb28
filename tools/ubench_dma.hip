// micro-benchmark: global_load_lds_dwordx4 (the gridding kernel's patch copy) from records 16 bytes apart (aligned) against records 12
// bytes apart (every lane's 16 bytes start on a 4-byte boundary only and take the next record's first word along).  Checks what lands
// in LDS and times 15 x 15-pixel patches at pseudo-random origins of 4096 packed images, 12 waves per CU as in the product.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dma.hip -o /tmp/ubench_dma && timeout 120 /tmp/ubench_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int STRIDE>
__global__ void __launch_bounds__(768) k(const unsigned char *__restrict__ recs, unsigned *__restrict__ bad, float *__restrict__ sink, int SX, int SY, int nimg, int iters)
{
    __shared__ __align__(16) unsigned char lds[12 * 5120];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsBase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)(lds + wv * 5120);
    unsigned off[5];
    for (int i = 0; i < 5; ++i) { const int slot = 64 * i + lane, row = min(slot / 21, 14), col = min(slot % 21, 14); off[i] = (unsigned)(row * SX + col) * STRIDE; }
    unsigned s = __builtin_amdgcn_readfirstlane((blockIdx.x * 12 + wv + 1) * 2654435761u);
    float acc = 0.f;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned img = __builtin_amdgcn_readfirstlane((s >> 8) % nimg), oy = __builtin_amdgcn_readfirstlane((s >> 3) % (SY - 15)), ox = __builtin_amdgcn_readfirstlane((s >> 13) % (SX - 15));
        const size_t bo = ((size_t)img * SY * SX + (size_t)oy * SX + ox) * STRIDE;
        const unsigned char *base = recs + (((size_t)__builtin_amdgcn_readfirstlane((unsigned)(bo >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)bo));
        for (int i = 0; i < 5; ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off[i]), "s"(base), "s"(ldsBase + 1024u * i) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // slot (row, col) must hold the first three floats of cell (oy + row, ox + col) of img: the cell's index, its negative, img
        const int slot = lane, row = slot / 21, col = slot % 21;
        const float *p = reinterpret_cast<const float *>(lds + wv * 5120 + 16 * slot);
        const float cell = (float)((oy + min(row, 14)) * (unsigned)SX + ox + min(col, 14));
        if (it < 64 && (p[0] != cell || p[1] != -cell || p[2] != (float)img)) ++nbad;
        acc += p[0];
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
    sink[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <int STRIDE> static void run(int SX, int SY, int nimg)
{
    const size_t cells = (size_t)nimg * SX * SY;
    std::vector<float> h(cells * (STRIDE / 4) + 4);
    for (size_t c = 0; c < cells; ++c) {
        const float v = (float)(c % ((size_t)SX * SY));
        h[c * (STRIDE / 4)] = v; h[c * (STRIDE / 4) + 1] = -v; h[c * (STRIDE / 4) + 2] = (float)(c / ((size_t)SX * SY));
        if (STRIDE == 16) h[c * 4 + 3] = 0.f;
    }
    unsigned char *d; unsigned *bad; float *sink;
    hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 4); hipMalloc(&sink, 256 * 768 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    const int iters = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<STRIDE>, dim3(256), dim3(768), 0, 0, d, bad, sink, SX, SY, nimg, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    const double patches = 256.0 * 12 * iters;
    printf("records %2d bytes apart: %u wrong slots, %.3f ms for %.0f patches, %.1f G patches/s, %.2f TB/s of requested bytes (15 x 15 x 16)\n", STRIDE, hb, ms, patches,
           patches / ms * 1e-6, patches * 15 * 15 * 16 / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(bad); hipFree(sink);
}

int main()
{
    run<16>(268, 524, 512);
    run<12>(268, 524, 512);
    return 0;
}
