// Test probe for xmipp3_amd/host/fastio.h (no device needed): prints what FastTable / parseNeighbourRows / StackSource read,
// next to what the plain MetaDataVec reader of minicore.h reads, so that tests/test_host_fastio.py can compare them.
#include "../../xmipp3_amd/host/fastio.h"
using namespace mc;

int main(int argc, char **argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "table" && argc >= 4) {
            const int threads = atoi(argv[3]);
            FastTable t;
            t.read(argv[2], threads);
            MetaDataVec m;
            m.read(argv[2]);
            if (t.labels != m.labels) { printf("LABELS DIFFER\n"); return 2; }
            if (t.size() != m.size()) { printf("SIZES DIFFER %zu %zu\n", t.size(), m.size()); return 2; }
            for (size_t r = 0; r < t.size(); ++r)
                for (size_t c = 0; c < t.labels.size(); ++c)
                    if (std::string(t.cell((int)c, r)) != m.rows[r][c]) { printf("CELL %zu %zu DIFFERS '%s' '%s'\n", r, c, std::string(t.cell((int)c, r)).c_str(), m.rows[r][c].c_str()); return 2; }
            printf("rows %zu labels %zu\n", t.size(), t.labels.size());
            for (auto &l : t.labels) printf("%s\t", l.c_str());
            printf("\n");
            for (size_t r = 0; r < std::min<size_t>(t.size(), 5); ++r) {
                for (size_t c = 0; c < t.labels.size(); ++c) printf("%s\t", std::string(t.cell((int)c, r)).c_str());
                printf("\n");
            }
            return 0;
        }
        if (mode == "numbers" && argc >= 4) {          // column as doubles, %.17g, one per line
            FastTable t;
            t.read(argv[2], 2);
            const int c = t.col(argv[3]);
            for (size_t r = 0; r < t.size(); ++r) printf("%.17g\n", t.getDouble(c, r, -12345.0));
            return 0;
        }
        if (mode == "neigh" && argc >= 4) {
            const int threads = atoi(argv[3]);
            auto f = std::make_shared<MappedFile>(argv[2]);
            FastTable t;
            t.read(f, "neighbors", threads);
            NeighbourLists nl;
            parseNeighbourRows(t, t.col("neighbors"), nl, threads);
            printf("images %zu lists %zu ids %zu\n", nl.size(), nl.span.size(), nl.ids.size());
            for (size_t i = 0; i < nl.size(); ++i) {
                for (uint32_t j = 0; j < nl.count(i); ++j) printf("%d ", nl.begin(i)[j]);
                printf("\n");
            }
            return 0;
        }
        if (mode == "stack" && argc >= 5) {            // images named in argv[4...] as raw floats on stdout
            const size_t dim = (size_t)atoi(argv[2]);
            const int threads = atoi(argv[3]);
            StackSource src;
            std::vector<StackSource::Loc> locs;
            for (int i = 4; i < argc; ++i) locs.push_back(src.locate(argv[i], dim));
            std::vector<float> out(locs.size() * dim * dim);
            const size_t T = (size_t)std::max(1, threads);
            runOnSlots(T, [&](size_t t) {
                std::vector<unsigned char> scratch;
                for (size_t i = locs.size() * t / T; i < locs.size() * (t + 1) / T; ++i) StackSource::readFloats(locs[i], out.data() + i * dim * dim, dim, scratch);
            });
            fwrite(out.data(), 4, out.size(), stdout);
            return 0;
        }
        if (mode == "write" && argc >= 4) {            // MetaDataVec round trip: read argv[2], write argv[3]
            MetaDataVec m;
            m.read(argv[2]);
            m.write(argv[3]);
            return 0;
        }
        fprintf(stderr, "usage: fastio_probe table|numbers|neigh|stack|write ...\n");
        return 1;
    } catch (const XmippError &e) {
        fprintf(stderr, "XmippError %d: %s\n", e.code, e.what());
        return 3;
    }
}
