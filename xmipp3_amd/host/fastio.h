// fastio.h -- the host side of the two hot-path programs at the rate the device consumes particles.
//
// The reference feeds its kernels from one loader thread (loadImageThread / preloadBuffer,
// reconstruction/reconstruct_fourier_accel.cpp:300-388,969-995) and reads its inputs through MetaData rows and
// Image<T>::read, one object at a time (angular_projection_matching.cpp:991-1191 processSomeImages,
// data/sampling.cpp:1592-1659 readSamplingFile).  At 100-300 k particles per second per device that is what bounds a
// run, not the kernels, so this file holds:
//   * MappedFile / FastTable : an "# XMIPP_STAR_1" block read in place (mmap), rows split and tokenised by several
//                              threads, cells kept as (offset, length) into the mapping, numbers converted on demand
//   * parseNeighbourRows     : the `neighbors` block of a _sampling.xmd (5 KB of text per image at 1000 references)
//                              into CSR lists, identical consecutive rows shared
//   * StackSource            : "n@stack" names resolved to (descriptor, offset) once per stack file
//   * BatchFeeder            : page-locked double buffer; reader threads pread() images straight into it and enqueue
//                              their piece of the batch on a copy stream while the device works on the previous batch
// No numerics here: everything below moves bytes.
#ifndef XMIPP3_AMD_FASTIO_H
#define XMIPP3_AMD_FASTIO_H
#include "minicore.h"
#include "../../include/xmipp_hip.h"
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <chrono>
#include <exception>
#include <fcntl.h>
#include <future>
#include <memory>
#include <mutex>
#include <sched.h>
#include <string_view>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>

namespace mc {

inline void xhCheck(int rc) { if (rc != XH_OK) REPORT_ERROR(ERR_GPU, std::string("xmipp_hip: ") + xh_last_error()); }

inline double nowSeconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// f(g) for g < n, one host thread per g; the first exception is re-thrown on the caller's thread
template <class F> inline void runOnSlots(size_t n, F f)
{
    if (n == 1) { f((size_t)0); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(n);
    for (size_t g = 0; g < n; ++g)
        th.emplace_back([&, g] { try { f(g); } catch (...) { err[g] = std::current_exception(); } });
    for (auto &t : th) t.join();
    for (size_t g = 0; g < n; ++g) if (err[g]) std::rethrow_exception(err[g]);
}

inline int hostThreads(int wanted = 0)
{
    if (wanted > 0) return wanted;
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw ? hw / 2 : 4u));
}

// The calling thread (and every thread it starts afterwards) onto the CPUs of the NUMA node the device hangs off.  With the
// loader's threads and its page-locked memory spread over both sockets of the host the H2D copy and the readers beside it
// run at 21 + 20 GB/s; on one node, either one, at 56 + 56 (tools/ubench_hostfeed.hip).  XMIPP_HIP_NO_BIND=1 leaves the
// affinity alone (a host that is shared by several jobs may have given this process its CPUs already).
inline int bindToDeviceNode(int device)
{
    if (getenv("XMIPP_HIP_NO_BIND")) return -1;
    int node = -1;
    if (xh_device_numa_node(device, &node) != XH_OK || node < 0) return -1;
    FILE *f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return -1;
    char buf[4096] = {0};
    const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!ok) return -1;
    cpu_set_t allowed, set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return -1;
    int count = 0;
    for (const char *c = buf; *c;) {
        if (!isdigit((unsigned char)*c)) { ++c; continue; }
        int a = (int)strtol(c, (char **)&c, 10), b = a;
        if (*c == '-') b = (int)strtol(c + 1, (char **)&c, 10);
        for (int k = a; k <= b && k < CPU_SETSIZE; ++k) if (CPU_ISSET(k, &allowed)) { CPU_SET(k, &set); ++count; }
    }
    if (count == 0 || sched_setaffinity(0, sizeof(set), &set) != 0) return -1;     // (none of the node's CPUs is ours: stay where we are)
    return node;
}

// ------------------------------------------------------------------ a file read where it lies
struct MappedFile {
    const char *p = nullptr;
    size_t n = 0;
    std::string path;
    explicit MappedFile(const std::string &fn) : path(fn)
    {
        const int fd = ::open(fn.c_str(), O_RDONLY);
        if (fd < 0) REPORT_ERROR(ERR_IO_NOTEXIST, "MetaData::read: cannot open " + fn);
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); REPORT_ERROR(ERR_IO_NOREAD, "MetaData::read: cannot stat " + fn); }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); REPORT_ERROR(ERR_IO_NOREAD, "MetaData::read: cannot map " + fn); }
            p = (const char *)m;
            madvise(m, n, MADV_SEQUENTIAL);
        }
        ::close(fd);
    }
    ~MappedFile() { if (p) munmap((void *)p, n); }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

inline bool isBlank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// One data_ block of a metadata file.  Same reading rules as MetaDataVec::read (minicore.h): comment lines (# ;) and
// empty lines skipped, `_label [value]` lines up to the first row, quoted cells may hold blanks, short rows are padded
// with empty cells, an empty cell reads as "label absent".
class FastTable {
public:
    struct Cell { uint32_t off = 0, len = 0; };            // relative to the row's first byte
    std::shared_ptr<MappedFile> file;
    std::vector<std::string> labels;
    std::vector<const char *> rowStart;
    std::vector<Cell> cells;                               // [row][label]
    std::vector<std::string> single;                       // values of a block without loop_

    size_t size() const { return single.empty() ? rowStart.size() : 1; }
    bool containsLabel(const std::string &l) const { return col(l) >= 0; }
    int col(const std::string &l) const
    {
        auto it = std::find(labels.begin(), labels.end(), l);
        return it == labels.end() ? -1 : (int)(it - labels.begin());
    }
    std::string_view cell(int c, size_t row) const
    {
        if (c < 0) return {};
        if (!single.empty()) return single[c];
        const Cell &e = cells[row * labels.size() + (size_t)c];
        return std::string_view(rowStart[row] + e.off, e.len);
    }
    static double toDouble(std::string_view s, double def)
    {
        if (s.empty()) return def;
        if (s[0] == '+') s.remove_prefix(1);
        double v = def;
        auto r = std::from_chars(s.data(), s.data() + s.size(), v);
        return r.ec == std::errc() ? v : atof(std::string(s).c_str());
    }
    static long toLong(std::string_view s, long def)
    {
        if (s.empty()) return def;
        if (s[0] == '+') s.remove_prefix(1);
        long v = def;
        auto r = std::from_chars(s.data(), s.data() + s.size(), v);
        return r.ec == std::errc() ? v : atol(std::string(s).c_str());
    }
    double getDouble(int c, size_t row, double def) const { return toDouble(cell(c, row), def); }
    long getLong(int c, size_t row, long def) const { return toLong(cell(c, row), def); }
    // the MetaDataVec spellings (label looked up per call: fine outside the per-image loops)
    bool getValue(const std::string &l, std::string &v, size_t row) const { auto s = cell(col(l), row); if (s.empty()) return false; v.assign(s); return true; }
    bool getValue(const std::string &l, long &v, size_t row) const { auto s = cell(col(l), row); if (s.empty()) return false; v = toLong(s, v); return true; }
    double getDouble(const std::string &l, size_t row, double def) const { return getDouble(col(l), row, def); }

    // fn may be "block@file"; empty block => first block of the file
    void read(const std::string &fnFull, int threads = 0)
    {
        FileName fn(fnFull);
        read(std::make_shared<MappedFile>(fn.path), fn.hasNumber() ? "" : fn.prefix, threads);
    }
    void read(std::shared_ptr<MappedFile> f, const std::string &block, int threads = 0)
    {
        file = std::move(f);
        labels.clear(); rowStart.clear(); cells.clear(); single.clear();
        const char *p = file->p, *end = p + file->n;
        // the block's first line: "data_<name>" at the start of a line
        const char *b = nullptr;
        for (const char *q = p; q && q < end;) {
            const char *hit = (q == p && end - p >= 5 && memcmp(p, "data_", 5) == 0) ? p : nullptr;
            if (!hit) {
                const char *m = (const char *)memmem(q, (size_t)(end - q), "\ndata_", 6);
                if (!m) break;
                hit = m + 1;
            }
            const char *e = (const char *)memchr(hit, '\n', (size_t)(end - hit));
            if (!e) e = end;
            const char *t = e;
            while (t > hit + 5 && isBlank(t[-1])) --t;
            if (block.empty() || std::string_view(hit + 5, (size_t)(t - hit - 5)) == block) { b = e; break; }
            q = e;
        }
        if (!b) REPORT_ERROR(ERR_MD_NOOBJ, "MetaData::read: block '" + block + "' not found in " + file->path);
        const char *next = (const char *)memmem(b, (size_t)(end - b), "\ndata_", 6);
        const char *stop = next ? next + 1 : end;
        // header: loop_ and the _label lines
        bool loop = false;
        const char *q = b < end ? b + 1 : end;
        std::vector<std::string> values;
        while (q < stop) {
            const char *e = (const char *)memchr(q, '\n', (size_t)(stop - q));
            if (!e) e = stop;
            const char *t = q;
            while (t < e && isBlank(*t)) ++t;
            if (t == e || *t == '#' || *t == ';') { q = e + 1; continue; }
            if (e - t >= 5 && memcmp(t, "loop_", 5) == 0) { loop = true; q = e + 1; continue; }
            if (*t != '_') break;
            std::vector<std::string> tk = MetaDataVec::tokenize(std::string(t, (size_t)(e - t)));
            labels.push_back(tk[0].substr(1));
            values.push_back(tk.size() > 1 ? tk[1] : "");
            q = e + 1;
        }
        if (!loop) { single = values; return; }       // one row of `_label value` pairs
        if (q >= stop || labels.empty()) return;
        // rows: T pieces cut at line ends, each split and tokenised by its own thread
        const size_t bytes = (size_t)(stop - q), L = labels.size();
        const size_t T = (size_t)std::max(1, std::min(hostThreads(threads), (int)(bytes / (1 << 20)) + 1));
        std::vector<const char *> cut(T + 1, stop);
        cut[0] = q;
        for (size_t t = 1; t < T; ++t) {
            const char *c = q + bytes * t / T;
            c = (const char *)memchr(c, '\n', (size_t)(stop - c));
            cut[t] = c ? c + 1 : stop;
        }
        std::vector<std::vector<const char *>> rs(T);
        std::vector<std::vector<Cell>> cs(T);
        runOnSlots(T, [&](size_t t) {
            auto &R = rs[t];
            auto &Cc = cs[t];
            for (const char *s = cut[t]; s < cut[t + 1];) {
                const char *e = (const char *)memchr(s, '\n', (size_t)(cut[t + 1] - s));
                if (!e) e = cut[t + 1];
                const char *a = s;
                while (a < e && isBlank(*a)) ++a;
                if (a < e && *a != '#' && *a != ';') {
                    R.push_back(a);
                    const size_t base = Cc.size();
                    Cc.resize(base + L);
                    size_t k = 0;
                    const char *c = a;
                    while (c < e && k < L) {
                        while (c < e && isBlank(*c)) ++c;
                        if (c >= e) break;
                        if (*c == '\'' || *c == '"') {
                            const char *z = (const char *)memchr(c + 1, *c, (size_t)(e - c - 1));
                            if (!z) z = e;
                            Cc[base + k++] = Cell{(uint32_t)(c + 1 - a), (uint32_t)(z - c - 1)};
                            c = z < e ? z + 1 : e;
                        } else {
                            const char *z = c;
                            while (z < e && !isBlank(*z)) ++z;
                            Cc[base + k++] = Cell{(uint32_t)(c - a), (uint32_t)(z - c)};
                            c = z;
                        }
                    }
                }
                s = e + 1;
            }
        });
        size_t n = 0;
        for (auto &r : rs) n += r.size();
        rowStart.reserve(n);
        cells.reserve(n * L);
        for (size_t t = 0; t < T; ++t) {
            rowStart.insert(rowStart.end(), rs[t].begin(), rs[t].end());
            cells.insert(cells.end(), cs[t].begin(), cs[t].end());
            std::vector<const char *>().swap(rs[t]);
            std::vector<Cell>().swap(cs[t]);
        }
    }
    // rows whose `enabled` cell is <= 0 dropped (MetaData::removeDisabled)
    void removeDisabled()
    {
        const int c = col("enabled");
        if (c < 0 || !single.empty()) return;
        const size_t L = labels.size();
        size_t w = 0;
        for (size_t r = 0; r < rowStart.size(); ++r) {
            if (getLong(c, r, 0) <= 0) continue;
            if (w != r) { rowStart[w] = rowStart[r]; std::copy(cells.begin() + r * L, cells.begin() + (r + 1) * L, cells.begin() + w * L); }
            ++w;
        }
        rowStart.resize(w);
        cells.resize(w * L);
    }
};

// The `neighbors` column of a _sampling.xmd: one blank-separated list of reference numbers per image.  Lists equal to
// the previous image's (byte for byte) are stored once.  listOf[i] -> (first, count) into ids.
struct NeighbourLists {
    std::vector<uint32_t> listOf;                  // image -> list
    std::vector<std::pair<size_t, uint32_t>> span; // list -> (first, count)
    std::vector<int32_t> ids;
    size_t size() const { return listOf.size(); }
    const int32_t *begin(size_t image) const { return ids.data() + span[listOf[image]].first; }
    uint32_t count(size_t image) const { return span[listOf[image]].second; }
};
inline void parseNeighbourRows(const FastTable &t, int column, NeighbourLists &out, int threads = 0)
{
    const size_t n = t.size();
    out.listOf.assign(n, 0);
    out.span.clear();
    out.ids.clear();
    if (!n || column < 0) return;
    const size_t T = (size_t)std::max(1, std::min(hostThreads(threads), (int)(n / 256) + 1));
    struct Part { std::vector<uint32_t> listOf; std::vector<std::pair<size_t, uint32_t>> span; std::vector<int32_t> ids; };
    std::vector<Part> part(T);
    runOnSlots(T, [&](size_t k) {
        Part &P = part[k];
        std::string_view prev;
        bool havePrev = false;
        for (size_t i = n * k / T; i < n * (k + 1) / T; ++i) {
            const std::string_view s = t.cell(column, i);
            if (havePrev && s.size() == prev.size() && memcmp(s.data(), prev.data(), s.size()) == 0) { P.listOf.push_back((uint32_t)P.span.size() - 1); continue; }
            const size_t first = P.ids.size();
            const char *c = s.data(), *e = c + s.size();
            while (c < e) {
                while (c < e && (unsigned char)(*c - '0') > 9) ++c;
                if (c >= e) break;
                int32_t v = 0;
                while (c < e && (unsigned char)(*c - '0') <= 9) v = v * 10 + (*c++ - '0');
                P.ids.push_back(v);
            }
            P.span.emplace_back(first, (uint32_t)(P.ids.size() - first));
            P.listOf.push_back((uint32_t)P.span.size() - 1);
            prev = s;
            havePrev = true;
        }
    });
    // stitch the parts; a part whose first list equals the previous part's last shares it as well
    size_t w = 0;
    for (size_t k = 0; k < T; ++k) {
        Part &P = part[k];
        if (P.span.empty()) continue;
        uint32_t shift = (uint32_t)out.span.size();
        bool merged = false;
        if (!out.span.empty()) {
            const auto &a = out.span.back();
            const auto &b = P.span.front();
            merged = a.second == b.second && memcmp(out.ids.data() + a.first, P.ids.data() + b.first, (size_t)a.second * 4) == 0;
        }
        const size_t idBase = out.ids.size();
        const size_t skipIds = merged ? P.span.front().second : 0;
        out.ids.insert(out.ids.end(), P.ids.begin() + skipIds, P.ids.end());
        for (size_t s = merged ? 1 : 0; s < P.span.size(); ++s) out.span.emplace_back(idBase + P.span[s].first - skipIds, P.span[s].second);
        if (merged) shift -= 1;
        for (uint32_t l : P.listOf) out.listOf[w++] = shift + l;
    }
}

// ------------------------------------------------------------------ image stacks
// "n@stack.stk" / "n@stack.mrcs" / single-image files -> where the pixels lie.  One descriptor per file, kept open.
class StackSource {
public:
    struct Loc { int32_t fd = -1; int32_t mode = 2; uint64_t off = 0; };    // self-contained: readers never look at `files`
    struct File { std::string path; int fd = -1; ImageInfo info; };
    std::vector<File> files;
    ~StackSource() { for (File &f : files) if (f.fd >= 0) ::close(f.fd); }
    // not thread-safe: called by the one thread that assembles a batch.  dim: the expected image size (dim x dim x 1)
    Loc locate(std::string_view name, size_t dim)
    {
        size_t at = name.find('@');
        size_t idx = 0;
        std::string_view path = name;
        if (at != std::string_view::npos) {
            bool digits = at > 0;
            for (size_t i = 0; i < at; ++i) digits = digits && (unsigned char)(name[i] - '0') <= 9;
            if (digits) { for (size_t i = 0; i < at; ++i) idx = idx * 10 + (size_t)(name[i] - '0'); path = name.substr(at + 1); }
        }
        int fid = -1;
        if (last >= 0 && files[(size_t)last].path == path) fid = last;
        else {
            auto it = index.find(std::string(path));
            if (it != index.end()) fid = it->second;
            else {
                File f;
                f.path = std::string(path);
                f.info = readInfo(f.path);
                f.fd = ::open(f.path.c_str(), O_RDONLY);
                if (f.fd < 0) REPORT_ERROR(ERR_IO_NOTEXIST, "Image::read: cannot open " + f.path);
                files.push_back(f);
                fid = (int)files.size() - 1;
                index.emplace(f.path, fid);
            }
            last = fid;
        }
        const ImageInfo &I = files[(size_t)fid].info;
        if (I.x != dim || I.y != dim || I.z != 1) REPORT_ERROR(ERR_MULTIDIM_SIZE, "Image " + std::string(name) + " has a different size");
        const size_t per = I.x * I.y * I.z, bpp = I.bytesPerPixel();
        Loc l;
        l.fd = files[(size_t)fid].fd;
        l.mode = I.mode;
        if (I.mrc) l.off = I.headerBytes + (idx > 0 ? (idx - 1) * per * bpp : 0);
        else if (I.isStack) { if (idx == 0) idx = 1; l.off = I.headerBytes + (idx - 1) * (I.perImageHeader + per * 4) + I.perImageHeader; }
        else l.off = I.headerBytes;
        if (I.isStack && idx > I.n) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: image " + std::to_string(idx) + " beyond the end of " + files[(size_t)fid].path);
        return l;
    }
    // thread-safe (pread): the image as floats
    static void readFloats(const Loc &l, float *dst, size_t dim, std::vector<unsigned char> &scratch)
    {
        const size_t per = dim * dim, bytes = per * (l.mode == 0 ? 1 : (l.mode == 1 || l.mode == 6) ? 2 : 4);
        char *to = (char *)dst;
        if (l.mode != 2) { scratch.resize(bytes); to = (char *)scratch.data(); }
        for (size_t got = 0; got < bytes;) {
            const ssize_t r = pread(l.fd, to + got, bytes - got, (off_t)(l.off + got));
            if (r <= 0) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: short read of an image (file truncated?)");
            got += (size_t)r;
        }
        struct { int mode; } I{l.mode};
        if (I.mode == 0) { const signed char *p = (const signed char *)to; for (size_t i = 0; i < per; ++i) dst[i] = (float)p[i]; }
        else if (I.mode == 1) { const int16_t *p = (const int16_t *)to; for (size_t i = 0; i < per; ++i) dst[i] = (float)p[i]; }
        else if (I.mode == 6) { const uint16_t *p = (const uint16_t *)to; for (size_t i = 0; i < per; ++i) dst[i] = (float)p[i]; }
    }

private:
    std::unordered_map<std::string, int> index;
    int last = -1;
};

// wall-clock seconds of the host side of a run, printed under XMIPP_HIP_TIMING=1.  `load` runs on the loader's threads
// under the device's work (it is on the critical path only through `stall`).
struct HostTiming {
    double setup = 0;         // device contexts, page-locked buffers
    double parse = 0;         // metadata files -> tables, neighbour lists
    double bank = 0;          // APM: reading the gallery + xh_pm_create
    double loop = 0;          // the image loop, wall clock (all devices side by side)
    double stall = 0;         //   main thread(s) waiting for the loader: the part of `load` that was NOT hidden
    double device = 0;        //   main thread(s) inside the library's calls (enqueue + waits for results)
    double load = 0;          //   loader threads: pread of the batches into page-locked memory + their H2D, under the device's work
    double h2d = 0;           //     of which waiting for the copy stream after the last read
    double format = 0;        //   result rows -> text (a worker thread, under the device's work)
    double finish = 0;        // RFA: mirror/crop, reduction, finaliser, volume D2H
    double write = 0;         // output files
    double total = 0;
    size_t images = 0;
    void add(const HostTiming &o)
    {
        setup += o.setup; parse += o.parse; bank += o.bank; load += o.load; h2d += o.h2d; stall += o.stall; device += o.device;
        format += o.format; finish += o.finish; write += o.write; images += o.images;
    }
    // one line, also machine readable: tools/bench_cli.py reads the key=value pairs
    void print(const char *prog) const
    {
        fprintf(stderr, "timing %s: total=%.4f setup=%.4f parse=%.4f bank=%.4f loop=%.4f stall=%.4f device=%.4f load=%.4f h2d_wait=%.4f format=%.4f "
                        "finish=%.4f write=%.4f images=%zu images_per_s_loop=%.0f images_per_s_total=%.0f\n",
                prog, total, setup, parse, bank, loop, stall, device, load, h2d, format, finish, write, images, images / std::max(1e-9, loop),
                images / std::max(1e-9, total));
    }
};

// Two device buffers of `capacity` images and the threads that fill them.  A batch is cut into pieces of 4 MB (runs of consecutive
// images); piece number g of the run (counted across batches) owns slot g % R of ONE page-locked ring of R pieces.  `readers` threads take
// pieces off a counter and pread() them into their slots; ONE copier thread -- the only one to talk to the runtime -- waits for the
// next piece in order, takes every consecutive piece that is ready with it and sends the whole run as one copy (neighbours in the ring
// are neighbours on the device), alternating between two copy streams so that a copy is queued while the previous one flies; a slot is
// handed back when its copy has completed.  Large copies are what the link wants: 43 GB/s in 4 MB copies, 55 in 16 MB ones
// (tools/ubench_hostfeed.hip).  (Measured on the 256-core host: readers that enqueue their own copies on their own streams get slower
// with every reader added -- 75 k particles/s with 16, 30 k with 64: the runtime serialises them and the compute thread's launches
// with them.  Page-locking costs ~0.25 s per GB, so the batch itself is never page-locked.)
// request(k) starts batch k into device buffer k & 1; take(k) waits until it is in HBM.  The copy streams first wait (on the device)
// for everything the compute context had been given when request() was called: the buffer they overwrite was last read by batch k - 2.
class BatchFeeder {
public:
    StackSource source;
    size_t dim = 0, capacity = 0;
    HostTiming *timing = nullptr;

    ~BatchFeeder() { release(); }
    void release()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            if (!started) return;
            cvDone.wait(lk, [&] { return !active || copied + failed >= npieces; });
            stop = true;
            started = false;
        }
        cvWork.notify_all();
        cvCopy.notify_all();
        cvSlot.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
        if (ring) xh_host_free(copyCtx[0], ring);
        ring = nullptr;
        for (int s = 0; s < 2; ++s) { if (d[s]) xh_free(copyCtx[0], d[s]); d[s] = nullptr; }
        for (int c = 0; c < 2; ++c) { if (copyCtx[c]) xh_ctx_destroy(copyCtx[c]); copyCtx[c] = nullptr; }
    }
    void create(int device, size_t dim_, size_t capacity_, int readers, HostTiming *tm)
    {
        dim = dim_; capacity = capacity_; timing = tm;
        const size_t per = dim * dim;
        const char *ePiece = getenv("XMIPP_HIP_PIECE_MB"), *eRing = getenv("XMIPP_HIP_RING_MB");      // (A/B knobs of tools/cli_sweep.sh)
        const size_t pieceBytes = (size_t)std::max(1, ePiece ? atoi(ePiece) : 4) << 20;
        pieceImages = std::max<size_t>(1, pieceBytes / (per * sizeof(float)));
        const size_t batchPieces = (capacity + pieceImages - 1) / pieceImages;
        const size_t nr = std::max<size_t>(1, std::min<size_t>((size_t)hostThreads(readers), batchPieces));
        // the ring: room for every reader's piece in progress plus the copies in flight (256 MB unless the batch is smaller)
        const size_t ringBytes = (size_t)std::max(16, eRing ? atoi(eRing) : 256) << 20;
        R = std::max<size_t>(2 * nr, std::min<size_t>(std::max<size_t>(2 * nr, batchPieces), ringBytes / (pieceImages * per * sizeof(float))));
        for (int c = 0; c < 2; ++c) xhCheck(xh_ctx_create_private(device, &copyCtx[c]));
        for (int s = 0; s < 2; ++s) xhCheck(xh_malloc(copyCtx[0], capacity * per * sizeof(float), (void **)&d[s]));
        xhCheck(xh_host_alloc(copyCtx[0], R * pieceImages * per * sizeof(float), (void **)&ring));
        ready.assign(R, 0);
        stop = false; started = true; active = false;
        for (size_t r = 0; r < nr; ++r) threads.emplace_back([this] { readerLoop(); });
        threads.emplace_back([this] { copierLoop(); });
    }
    // where the images of batch k lie -> device buffer k & 1; `compute`: the context whose queued work must finish before the buffer is overwritten
    void request(size_t k, std::vector<StackSource::Loc> locs_, xh_ctx *compute)
    {
        if (locs_.size() > capacity) REPORT_ERROR(ERR_LOGIC_ERROR, "BatchFeeder: batch larger than the buffers");
        if (compute) for (int c = 0; c < 2; ++c) xhCheck(xh_ctx_wait_for(copyCtx[c], compute));
        {
            std::lock_guard<std::mutex> lk(m);
            if (active) REPORT_ERROR(ERR_LOGIC_ERROR, "BatchFeeder: request() before take()");
            locs = std::move(locs_);
            slot = (int)(k & 1);
            base += npieces;                       // the pieces of this batch continue the ring where the last batch stopped
            npieces = (locs.size() + pieceImages - 1) / pieceImages;
            next = 0; copied = 0; failed = 0; err = nullptr;
            active = true;
            t0 = nowSeconds();
        }
        cvWork.notify_all();
        cvCopy.notify_all();
    }
    float *take(size_t k)
    {
        const double w0 = nowSeconds();
        std::unique_lock<std::mutex> lk(m);
        cvDone.wait(lk, [&] { return copied + failed >= npieces; });
        active = false;
        if (timing) { timing->stall += nowSeconds() - w0; timing->load += tEnd - t0; timing->h2d += std::max(0.0, tEnd - tRead); }
        if (err) std::rethrow_exception(err);
        return d[k & 1];
    }

private:
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cvWork, cvCopy, cvDone, cvSlot;
    xh_ctx *copyCtx[2] = {nullptr, nullptr};
    float *d[2] = {nullptr, nullptr}, *ring = nullptr;
    std::vector<StackSource::Loc> locs;
    std::vector<char> ready;             // per ring slot: the piece in it has been read and waits for its copy
    size_t pieceImages = 1, R = 2;
    size_t base = 0;                     // ring sequence number of the current batch's piece 0
    size_t npieces = 0, next = 0;        // pieces of the current batch; the next one a reader takes
    size_t copied = 0, failed = 0;       // pieces of the current batch whose copy has completed (in order) / that could not be read or copied
    size_t retired = 0;                  // ring sequence number below which every slot is free again
    int slot = 0;
    bool stop = false, started = false, active = false;
    double t0 = 0, tEnd = 0, tRead = 0;
    std::exception_ptr err;

    void readerLoop()
    {
        std::vector<unsigned char> scratch;
        const size_t per = dim * dim;
        for (;;) {
            size_t piece, seq;
            {
                std::unique_lock<std::mutex> lk(m);
                cvWork.wait(lk, [&] { return stop || (active && next < npieces); });
                if (stop) return;
                piece = next++;
                seq = base + piece;
                cvSlot.wait(lk, [&] { return stop || seq < retired + R; });       // the slot's previous piece has left
                if (stop) return;
            }
            const size_t first = piece * pieceImages, count = std::min(pieceImages, locs.size() - first);
            float *dst = ring + (seq % R) * pieceImages * per;
            bool ok = true;
            try {
                for (size_t i = 0; i < count; ++i) StackSource::readFloats(locs[first + i], dst + i * per, dim, scratch);
            } catch (...) {
                ok = false;
                std::lock_guard<std::mutex> lk(m);
                if (!err) err = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> lk(m);
                ready[seq % R] = ok ? 1 : 2;                 // 2: nothing to copy, the slot is passed on as it is
                tRead = nowSeconds();                        // (the last reader to finish leaves the batch's read-end time)
            }
            cvCopy.notify_one();
        }
    }
    // one copy in flight per stream: (stream, ring sequence numbers [from, to) it carries)
    struct Flight { bool busy = false; size_t from = 0, to = 0, bad = 0; };
    void retire(Flight &f, int c, bool &broken)
    {
        if (!f.busy) return;
        if (xh_ctx_sync(copyCtx[c]) != XH_OK) broken = true;
        std::lock_guard<std::mutex> lk(m);
        if (broken && !err) err = std::make_exception_ptr(XmippError(ERR_GPU, std::string("xmipp_hip: ") + xh_last_error()));
        const size_t n = f.to - f.from;
        failed += broken ? n : f.bad;
        copied += broken ? 0 : n - f.bad;
        retired = f.to;
        tEnd = nowSeconds();
        f.busy = false;
        cvSlot.notify_all();
        cvDone.notify_all();
    }
    void copierLoop()
    {
        const size_t per = dim * dim;
        Flight fl[2];
        size_t nextSeq = 0;               // the next ring sequence number to send
        int c = 0;
        bool broken = false;
        for (;;) {
            size_t from, to, firstImage, bad = 0;
            int dstSlot;
            {
                std::unique_lock<std::mutex> lk(m);
                // the next piece in order is ready -- or there is a copy in flight to look after meanwhile
                cvCopy.wait(lk, [&] { return stop || (active && nextSeq < base + npieces && ready[nextSeq % R]) || fl[0].busy || fl[1].busy; });
                if (stop && !fl[0].busy && !fl[1].busy) return;
                const bool have = active && nextSeq >= base && nextSeq < base + npieces && ready[nextSeq % R];
                if (!have) {
                    lk.unlock();
                    retire(fl[c], c, broken);              // nothing to send: finish what flies, oldest first (stream c carries the older copy)
                    retire(fl[c ^ 1], c ^ 1, broken);
                    continue;
                }
                // every consecutive ready piece up to the end of the batch or the wrap of the ring goes in one copy
                from = nextSeq;
                to = from;
                while (to < base + npieces && ready[to % R] && (to == from || to % R != 0)) {
                    if (ready[to % R] == 2) ++bad;
                    ready[to % R] = 0;
                    ++to;
                }
                firstImage = (from - base) * pieceImages;
                dstSlot = slot;
            }
            retire(fl[c], c, broken);                       // this stream's previous copy (the other stream's is still flying)
            const size_t images = std::min((to - base) * pieceImages, locs.size()) - firstImage;
            if (!broken && bad == 0 &&
                xh_memcpy_h2d_async(copyCtx[c], d[dstSlot] + firstImage * per, ring + (from % R) * pieceImages * per, images * per * sizeof(float)) != XH_OK)
                broken = true;
            fl[c].busy = true; fl[c].from = from; fl[c].to = to; fl[c].bad = bad ? to - from : 0;
            nextSeq = to;
            c ^= 1;
            {
                // the batch's last run: nothing more will come for a while, finish both flights so that take() returns
                std::unique_lock<std::mutex> lk(m);
                const bool last = nextSeq >= base + npieces;
                lk.unlock();
                if (last) { retire(fl[c], c, broken); retire(fl[c ^ 1], c ^ 1, broken); }
            }
        }
    }
};

} // namespace mc
#endif
