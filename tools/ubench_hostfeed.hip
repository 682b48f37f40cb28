// Microbenchmark behind xmipp3_amd/host/fastio.h: what the host can deliver to the device.
//   (1) H2D from page-locked memory: one 1 GB copy, and the same as 4 MB pieces on one stream
//   (2) pread() of a /dev/shm file into page-locked memory with T threads (the page-cache copy the programs' readers do)
//   (3) the same into pageable memory (is the page-locked target the slow part?)
// build: hipcc --offload-arch=gfx950 -O2 -pthread tools/ubench_hostfeed.hip -o /tmp/ubench_hostfeed ; run: /tmp/ubench_hostfeed [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <sched.h>
#include <string>
#include <fstream>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// threads created after this call inherit the mask: the CPUs of one NUMA node (from /sys/devices/system/node/nodeN/cpulist)
static void bind_to_node(int node)
{
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s;
    std::getline(f, s);
    cpu_set_t set;
    CPU_ZERO(&set);
    size_t i = 0;
    while (i < s.size()) {
        int a = atoi(s.c_str() + i), b = a;
        while (i < s.size() && isdigit(s[i])) ++i;
        if (i < s.size() && s[i] == '-') { ++i; b = atoi(s.c_str() + i); while (i < s.size() && isdigit(s[i])) ++i; }
        for (int c = a; c <= b; ++c) CPU_SET(c, &set);
        if (i < s.size() && s[i] == ',') ++i;
    }
    if (sched_setaffinity(0, sizeof(set), &set) != 0) perror("sched_setaffinity");
    printf("bound to node %d (cpus %s)\n", node, s.c_str());
}

int main(int argc, char **argv)
{
    const size_t GB = argc > 1 ? (size_t)atoi(argv[1]) : 1, bytes = GB << 30;
    const int node = argc > 2 ? atoi(argv[2]) : -1;
    {
        char bdf[64] = {0};
        CK(hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0));
        for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
        std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
        std::string v; std::getline(f, v);
        printf("device 0 is %s, numa_node %s\n", bdf, v.c_str());
    }
    if (node >= 0) bind_to_node(node);
    char *pin, *dev;
    double t0 = now();
    CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault));
    printf("hipHostMalloc %zu GB: %.3f s\n", GB, now() - t0);
    CK(hipMalloc((void **)&dev, bytes));
    memset(pin, 1, bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        printf("H2D one copy: %.1f GB/s\n", bytes / (now() - t0) / 1e9);
    }
    for (size_t piece : {1u << 20, 4u << 20, 16u << 20, 64u << 20}) {
        t0 = now();
        for (size_t o = 0; o < bytes; o += piece) CK(hipMemcpyAsync(dev + o, pin + o, piece, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        printf("H2D %zu MB pieces, one stream: %.1f GB/s\n", piece >> 20, bytes / (now() - t0) / 1e9);
    }
    {   // pieces with a sync after every 8 (the copier thread's pattern)
        const size_t piece = 4u << 20;
        t0 = now();
        size_t k = 0;
        for (size_t o = 0; o < bytes; o += piece) { CK(hipMemcpyAsync(dev + o, pin + o, piece, hipMemcpyHostToDevice, s)); if (++k % 8 == 0) CK(hipStreamSynchronize(s)); }
        CK(hipStreamSynchronize(s));
        printf("H2D 4 MB pieces, sync every 8: %.1f GB/s\n", bytes / (now() - t0) / 1e9);
    }
    // the file
    const char *path = "/dev/shm/ubench_hostfeed.bin";
    {
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        for (size_t o = 0; o < bytes; o += 64 << 20) if (write(fd, pin + o, 64 << 20) < 0) { perror("write"); return 1; }
        close(fd);
    }
    char *pageable = (char *)malloc(bytes);
    memset(pageable, 2, bytes);
    for (char *dst : {pin, pageable})
        for (int T : {1, 4, 8, 16, 32, 64}) {
            int fd = open(path, O_RDONLY);
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    const size_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
                    for (size_t o = lo; o < hi; o += 256 << 10) { const size_t n = std::min<size_t>(256 << 10, hi - o); if (pread(fd, dst + o, n, (off_t)o) != (ssize_t)n) { perror("pread"); exit(1); } }
                });
            for (auto &x : th) x.join();
            printf("pread 256 KB x %d threads -> %s: %.1f GB/s\n", T, dst == pin ? "page-locked" : "pageable", bytes / (now() - t0) / 1e9);
            close(fd);
        }
    // both at once: 16 readers into page-locked memory while the device copy of the other half runs
    {
        int fd = open(path, O_RDONLY);
        const size_t half = bytes / 2;
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 16; ++t)
            th.emplace_back([&, t] {
                const size_t lo = half * t / 16, hi = half * (t + 1) / 16;
                for (size_t o = lo; o < hi; o += 256 << 10) { const size_t n = std::min<size_t>(256 << 10, hi - o); if (pread(fd, pin + o, n, (off_t)o) != (ssize_t)n) exit(1); }
            });
        CK(hipMemcpyAsync(dev + half, pin + half, half, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double tc = now() - t0;
        for (auto &x : th) x.join();
        printf("together: H2D of half %.1f GB/s while 16 readers fill the other half at %.1f GB/s\n", half / tc / 1e9, half / (now() - t0) / 1e9);
        close(fd);
    }
    {
        int fd = open(path, O_RDONLY);
        const size_t half = bytes / 2;
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 16; ++t)
            th.emplace_back([&, t] {
                const size_t lo = half * t / 16, hi = half * (t + 1) / 16;
                for (size_t o = lo; o < hi; o += 256 << 10) { const size_t n = std::min<size_t>(256 << 10, hi - o); if (pread(fd, pageable + o, n, (off_t)o) != (ssize_t)n) exit(1); }
            });
        CK(hipMemcpyAsync(dev + half, pin + half, half, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double tc = now() - t0;
        for (auto &x : th) x.join();
        printf("together, readers into PAGEABLE memory: H2D of half %.1f GB/s while 16 readers run at %.1f GB/s\n", half / tc / 1e9, half / (now() - t0) / 1e9);
        close(fd);
    }
    unlink(path);
    return 0;
}
