// The programs' loader alone (xmipp3_amd/host/fastio.h: BatchFeeder): batches of a Spider stack from the page cache into HBM, no compute beside it.
//   g++ -O2 -std=c++17 -pthread tools/ubench_feeder.cpp -o /tmp/ubench_feeder -Lxmipp3_amd -lxmipp_hip -Wl,-rpath,$PWD/xmipp3_amd -Wl,-rpath,/opt/rocm/lib
//   /tmp/ubench_feeder <stack.stk> <dim> <images in the stack> <batch> <batches> [readers]
#include "../xmipp3_amd/host/fastio.h"
using namespace mc;
int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: ubench_feeder stack dim nimages batch batches [readers]\n"); return 1; }
    try {
        const std::string stack = argv[1];
        const size_t dim = (size_t)atoi(argv[2]), nimg = (size_t)atoi(argv[3]), B = (size_t)atoi(argv[4]), nb = (size_t)atoi(argv[5]);
        const int readers = argc > 6 ? atoi(argv[6]) : 0;
        bindToDeviceNode(0);
        HostTiming tm;
        BatchFeeder f;
        f.create(0, dim, B, readers, &tm);
        auto locsOf = [&](size_t k) {
            std::vector<StackSource::Loc> l(B);
            for (size_t i = 0; i < B; ++i) l[i] = f.source.locate(std::to_string((k * B + i) % nimg + 1) + "@" + stack, dim);
            return l;
        };
        f.request(0, locsOf(0), nullptr);
        f.take(0);                                     // warm-up: files opened, pages touched
        const double t0 = nowSeconds();
        f.request(1, locsOf(1), nullptr);
        for (size_t k = 1; k <= nb; ++k) {
            f.take(k);
            if (k < nb) f.request(k + 1, locsOf(k + 1), nullptr);
        }
        const double dt = nowSeconds() - t0;
        const double gb = (double)nb * B * dim * dim * 4 / 1e9;
        printf("%zu batches of %zu images of %zu px with %d readers: %.3f s, %.1f GB/s into HBM, %.1f ms per batch (read-end to copy-end %.1f ms per batch)\n", nb, B, dim,
               readers, dt, gb / dt, 1e3 * dt / nb, 1e3 * tm.h2d / (nb + 1));
    } catch (const XmippError &e) { fprintf(stderr, "XmippError: %s\n", e.what()); return 2; }
    return 0;
}
