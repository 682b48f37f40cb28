// minicore.h -- the sliver of xmippCore's API surface that the two hot-path programs touch.
//
// xmippCore (XmippProgram, MetaData*, Image<T>, FileName, SymList ...) is a separate upstream
// repository that is not part of the reference tree (I2PC/xmippCore @ v4).
// The host programs in this directory keep the reference's program contract -- same flags,
// same metadata labels, same file formats -- on top of this stand-in:
//   * XmippProgram : the declarative argument DSL used verbatim by defineParams()
//                    (reconstruction/angular_projection_matching.cpp:83-121,
//                     reconstruction/reconstruct_fourier_accel.cpp:55-82), tryRun() error contract
//   * MetaDataVec  : "# XMIPP_STAR_1" files, data_<block> / loop_ / block@file
//                    (format: resources/test/sampling/*.xmd)
//   * Image I/O    : Spider single/stack/volume and MRC (mode 2), n@stack addressing
//   * SymList      : cyclic groups
#ifndef XMIPP3_AMD_MINICORE_H
#define XMIPP3_AMD_MINICORE_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mc {

// ------------------------------------------------------------------ errors
enum ErrorType { ERR_ARG_INCORRECT = 2, ERR_ARG_MISSING = 3, ERR_IO_NOTEXIST = 20, ERR_IO_NOREAD = 21,
                 ERR_MD_NOOBJ = 30, ERR_MD_BADLABEL = 31, ERR_MULTIDIM_SIZE = 40, ERR_VALUE_INCORRECT = 50,
                 ERR_GPU = 60, ERR_NOT_IMPLEMENTED = 61, ERR_LOGIC_ERROR = 62 };
struct XmippError : public std::runtime_error {
    int code;
    XmippError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define REPORT_ERROR(code, msg) throw mc::XmippError(code, std::string(msg))

inline std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
inline bool fileExists(const std::string &fn) { std::ifstream f(fn); return f.good(); }

// ------------------------------------------------------------------ file names
// "n@stack.stk" (1-based n), "block@file.xmd"
struct FileName {
    std::string prefix, path;   // prefix = text before '@' ("" if none)
    FileName() {}
    FileName(const std::string &s) { size_t p = s.find('@'); if (p == std::string::npos) path = s; else { prefix = s.substr(0, p); path = s.substr(p + 1); } }
    std::string str() const { return prefix.empty() ? path : prefix + "@" + path; }
    bool hasNumber() const { return !prefix.empty() && std::all_of(prefix.begin(), prefix.end(), ::isdigit); }
    size_t number() const { return hasNumber() ? (size_t)std::stoul(prefix) : 0; }
    std::string extension() const { size_t p = path.rfind('.'); return p == std::string::npos ? "" : path.substr(p + 1); }
    std::string removeAllExtensions() const
    {
        size_t slash = path.rfind('/');
        size_t p = path.find('.', slash == std::string::npos ? 0 : slash);
        return p == std::string::npos ? path : path.substr(0, p);
    }
};

// ------------------------------------------------------------------ metadata
class MetaDataVec {
public:
    std::vector<std::string> labels;
    std::vector<std::vector<std::string>> rows;
    std::string comment;

    size_t size() const { return rows.size(); }
    bool containsLabel(const std::string &l) const { return std::find(labels.begin(), labels.end(), l) != labels.end(); }
    int col(const std::string &l) const
    {
        auto it = std::find(labels.begin(), labels.end(), l);
        return it == labels.end() ? -1 : (int)(it - labels.begin());
    }
    void addLabel(const std::string &l) { if (!containsLabel(l)) { labels.push_back(l); for (auto &r : rows) r.push_back(""); } }
    size_t addObject() { rows.emplace_back(labels.size()); return rows.size() - 1; }
    void setValue(const std::string &l, const std::string &v, size_t id) { addLabel(l); rows[id].resize(labels.size()); rows[id][col(l)] = v; }
    void setValue(const std::string &l, double v, size_t id) { char b[64]; snprintf(b, sizeof(b), "%.6f", v); setValue(l, std::string(b), id); }
    void setValue(const std::string &l, long v, size_t id) { setValue(l, std::to_string(v), id); }
    bool getValue(const std::string &l, std::string &v, size_t id) const { int c = col(l); if (c < 0 || rows[id][c].empty()) return false; v = rows[id][c]; return true; }
    bool getValue(const std::string &l, double &v, size_t id) const { std::string s; if (!getValue(l, s, id)) return false; v = atof(s.c_str()); return true; }
    bool getValue(const std::string &l, long &v, size_t id) const { std::string s; if (!getValue(l, s, id)) return false; v = atol(s.c_str()); return true; }
    double getDouble(const std::string &l, size_t id, double def) const { double v = def; getValue(l, v, id); return v; }

    static std::vector<std::string> tokenize(const std::string &line)
    {
        std::vector<std::string> t;
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && isspace((unsigned char)line[i])) ++i;
            if (i >= line.size()) break;
            if (line[i] == '\'' || line[i] == '"') {
                char q = line[i];
                size_t j = line.find(q, i + 1);
                if (j == std::string::npos) j = line.size();
                t.push_back(line.substr(i + 1, j - i - 1));
                i = j + 1;
            } else {
                size_t j = i;
                while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
                t.push_back(line.substr(i, j - i));
                i = j;
            }
        }
        return t;
    }

    // fn may be "block@file"; empty block => first block of the file
    void read(const std::string &fnFull)
    {
        FileName fn(fnFull);
        const std::string block = fn.hasNumber() ? "" : fn.prefix;
        std::ifstream f(fn.path);
        if (!f.good()) REPORT_ERROR(ERR_IO_NOTEXIST, "MetaData::read: cannot open " + fn.path);
        labels.clear(); rows.clear();
        std::string line;
        bool inBlock = false, found = false, loop = false, header = true;
        std::vector<std::string> single;
        while (std::getline(f, line)) {
            std::string t = trim(line);
            if (t.empty() || t[0] == '#' || t[0] == ';') continue;
            if (t.rfind("data_", 0) == 0) {
                if (inBlock) break;             // next block starts: done
                std::string name = t.substr(5);
                if (block.empty() || name == block) { inBlock = found = true; loop = false; header = true; }
                continue;
            }
            if (!inBlock) continue;
            if (t == "loop_") { loop = true; continue; }
            if (t[0] == '_' && header) {
                std::vector<std::string> tk = tokenize(t);
                labels.push_back(tk[0].substr(1));
                if (!loop) single.push_back(tk.size() > 1 ? tk[1] : "");
                continue;
            }
            header = false;
            if (loop) {
                std::vector<std::string> tk = tokenize(t);
                tk.resize(labels.size());
                rows.push_back(tk);
            }
        }
        if (!found) REPORT_ERROR(ERR_MD_NOOBJ, "MetaData::read: block '" + block + "' not found in " + fn.path);
        if (!loop && !labels.empty()) rows.push_back(single);
    }

    void write(const std::string &fnFull, bool append = false) const
    {
        FileName fn(fnFull);
        const std::string block = (fn.prefix.empty() || fn.hasNumber()) ? "noname" : fn.prefix;
        const bool exists = fileExists(fn.path);
        if (append && exists) {
            // MD_APPEND: a block of this name already in the file is replaced, the other blocks stay.  The name is compared the way
            // getBlocksInMetaDataFile reads it (trailing blanks / CR of files written elsewhere do not make a second block), and the
            // kept content goes through a temporary file that is renamed over the original: a crash in between loses nothing
            std::ifstream in(fn.path);
            std::string line, kept;
            bool skipping = false, dropped = false;
            while (std::getline(in, line)) {
                if (line.compare(0, 5, "data_") == 0) {
                    std::string b = line.substr(5);
                    while (!b.empty() && (b.back() == ' ' || b.back() == '\r' || b.back() == '\t')) b.pop_back();
                    skipping = (b == block);
                    dropped = dropped || skipping;
                }
                if (!skipping) kept += line + "\n";
            }
            in.close();
            if (dropped) {
                const std::string tmp = fn.path + ".tmp_md_append";
                { std::ofstream o(tmp, std::ios::trunc); o << kept; if (!o.good()) REPORT_ERROR(ERR_IO_NOREAD, "MetaData::write: cannot write " + tmp); }
                if (std::rename(tmp.c_str(), fn.path.c_str()) != 0) REPORT_ERROR(ERR_IO_NOREAD, "MetaData::write: cannot replace " + fn.path);
            }
        }
        std::ofstream f(fn.path, append ? std::ios::app : std::ios::trunc);
        if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "MetaData::write: cannot write " + fn.path);
        if (!(append && exists)) f << "# XMIPP_STAR_1 * \n# \n";
        if (!comment.empty()) f << "# " << comment << "\n";
        f << "data_" << block << "\nloop_\n";
        for (auto &l : labels) f << " _" << l << "\n";
        // rows through one text buffer (a million rows of eleven cells: stream formatting per cell was a second of the run)
        std::string buf;
        buf.reserve(1 << 22);
        const std::string zero = "0", none;
        for (auto &r : rows) {
            for (size_t c = 0; c < labels.size(); ++c) {
                const std::string &v = c < r.size() ? r[c] : none;
                if (v.find(' ') != std::string::npos && v.find('\'') == std::string::npos) { buf += " '"; buf += v; buf += '\''; }
                else {
                    const std::string &w = v.empty() ? zero : v;
                    buf += ' ';
                    if (w.size() < 12) buf.append(12 - w.size(), ' ');
                    buf += w;
                }
            }
            buf += " \n";
            if (buf.size() > (1u << 22) - 4096) { f.write(buf.data(), (std::streamsize)buf.size()); buf.clear(); }
        }
        f.write(buf.data(), (std::streamsize)buf.size());
        if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "MetaData::write: short write to " + fn.path);
    }
};

// getBlocksInMetaDataFile (xmippCore metadata_extension): the names of the data_ blocks of a metadata file, in file order
inline std::vector<std::string> getBlocksInMetaDataFile(const std::string &path)
{
    std::vector<std::string> out;
    std::ifstream f(FileName(path).path);
    if (!f.good()) REPORT_ERROR(ERR_IO_NOTEXIST, "MetaData::read: cannot open " + path);
    std::string line;
    while (std::getline(f, line))
        if (line.compare(0, 5, "data_") == 0) {
            std::string b = line.substr(5);
            while (!b.empty() && (b.back() == ' ' || b.back() == '\r' || b.back() == '\t')) b.pop_back();
            out.push_back(b);
        }
    return out;
}

// ------------------------------------------------------------------ images
struct ImageInfo { size_t x = 0, y = 0, z = 1, n = 1; bool isStack = false; size_t headerBytes = 0, perImageHeader = 0; bool mrc = false, swap = false;
                   int mode = 2;          // MRC data mode of the file: 0 int8, 1 int16, 2 float32, 6 uint16 (Spider files: 2)
                   size_t bytesPerPixel() const { return mode == 0 ? 1 : (mode == 1 || mode == 6) ? 2 : 4; } };

inline bool isMrcExt(const std::string &e) { return e == "mrc" || e == "mrcs" || e == "map" || e == "st"; }

inline ImageInfo readInfo(const std::string &path)
{
    FileName fn(path);
    std::ifstream f(fn.path, std::ios::binary);
    if (!f.good()) REPORT_ERROR(ERR_IO_NOTEXIST, "Image::read: cannot open " + fn.path);
    ImageInfo I;
    if (isMrcExt(fn.extension())) {
        int32_t h[256];
        f.read((char *)h, 1024);
        if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: short MRC header in " + fn.path);
        // MRC2014 data modes as xmippCore's reader takes them (rwMRC: 0 signed bytes, 1 int16, 2 float32, 6 uint16), cast to float
        if (h[3] != 0 && h[3] != 1 && h[3] != 2 && h[3] != 6)
            REPORT_ERROR(ERR_IO_NOREAD, "Image::read: MRC mode " + std::to_string(h[3]) + " is not supported (0 int8, 1 int16, 2 float32, 6 uint16): " + fn.path);
        I.mrc = true; I.x = h[0]; I.y = h[1]; I.mode = h[3];
        const bool stack = fn.extension() == "mrcs" || fn.extension() == "st";
        if (stack) { I.z = 1; I.n = h[2]; I.isStack = true; } else { I.z = h[2]; I.n = 1; }
        I.headerBytes = 1024 + (size_t)h[23];
        return I;
    }
    float h[64];
    f.read((char *)h, sizeof(h));
    if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: short Spider header in " + fn.path);
    // Spider header (1-based word index): 1 NZ, 2 NY, 5 IFORM, 12 NX, 13 LABREC, 22 LABBYT, 23 LENBYT, 24 ISTACK, 26 MAXIM
    I.z = (size_t)std::max(1.f, std::fabs(h[0])); I.y = (size_t)h[1]; I.x = (size_t)h[11];
    if (I.x == 0 || I.y == 0 || I.x > 65536 || I.y > 65536) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: not a (native-endian) Spider file: " + fn.path);
    size_t labbyt = (size_t)h[21];
    if (labbyt == 0) { size_t lenbyt = I.x * 4, labrec = (1024 + lenbyt - 1) / lenbyt; labbyt = labrec * lenbyt; }
    I.headerBytes = labbyt;
    if (h[23] > 0) { I.isStack = true; I.n = (size_t)h[25]; I.perImageHeader = labbyt; }
    return I;
}

// reads image `index` (1-based for stacks; 0 => the only image / whole volume) as float
// the bytes of one image as the file holds them (I.mode tells what they are): the movie program sends counts to the device
inline void readImageRaw(const std::string &name, std::vector<unsigned char> &raw, ImageInfo &I)
{
    FileName fn(name);
    // consecutive reads usually hit the same stack: keep its header and its stream (one per host thread; a file
    // that is rewritten between reads of the same process must not be read through this cache -- none is)
    struct Cached { std::string path; ImageInfo info; std::ifstream f; };
    static thread_local Cached cache;
    if (cache.path != fn.path || !cache.f.is_open()) {
        cache.info = readInfo(fn.path);
        cache.f.close();
        cache.f.clear();
        cache.f.open(fn.path, std::ios::binary);
        cache.path = fn.path;
    }
    I = cache.info;
    std::ifstream &f = cache.f;
    f.clear();
    size_t idx = fn.hasNumber() ? fn.number() : 0;
    const size_t per = I.x * I.y * I.z, bpp = I.bytesPerPixel();
    size_t off;
    if (I.mrc) off = I.headerBytes + (idx > 0 ? (idx - 1) * per * bpp : 0);
    else if (I.isStack) { if (idx == 0) idx = 1; off = I.headerBytes + (idx - 1) * (I.perImageHeader + per * 4) + I.perImageHeader; }
    else off = I.headerBytes;
    if (I.isStack && idx > I.n) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: image " + std::to_string(idx) + " beyond the end of " + fn.path);
    raw.resize(per * bpp);
    f.seekg((std::streamoff)off);
    f.read((char *)raw.data(), per * bpp);
    if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::read: short read in " + fn.path);
}

inline void readImage(const std::string &name, std::vector<float> &data, ImageInfo &I)
{
    static thread_local std::vector<unsigned char> raw;
    readImageRaw(name, raw, I);
    const size_t per = I.x * I.y * I.z;
    data.resize(per);
    if (I.mode == 2) memcpy(data.data(), raw.data(), per * 4);
    else if (I.mode == 0) { const signed char *p = (const signed char *)raw.data(); for (size_t i = 0; i < per; ++i) data[i] = (float)p[i]; }
    else if (I.mode == 1) { const int16_t *p = (const int16_t *)raw.data(); for (size_t i = 0; i < per; ++i) data[i] = (float)p[i]; }
    else { const uint16_t *p = (const uint16_t *)raw.data(); for (size_t i = 0; i < per; ++i) data[i] = (float)p[i]; }
}

inline void spiderHeader(std::vector<float> &h, size_t x, size_t y, size_t z, int iform, int istack, size_t maxim, size_t imgnum)
{
    const size_t lenbyt = x * 4, labrec = (1024 + lenbyt - 1) / lenbyt, labbyt = labrec * lenbyt;
    h.assign(labbyt / 4, 0.f);
    h[0] = (float)z; h[1] = (float)y; h[2] = (float)(labrec + y * z); h[4] = (float)iform; h[11] = (float)x;
    h[12] = (float)labrec; h[21] = (float)labbyt; h[22] = (float)lenbyt; h[23] = (float)istack;
    h[25] = (float)maxim; h[26] = (float)imgnum;
}

// writes a single image / volume (double data narrowed to float): Spider or MRC by extension
inline void writeVolume(const std::string &path, const double *data, size_t x, size_t y, size_t z)
{
    FileName fn(path);
    std::ofstream f(fn.path, std::ios::binary | std::ios::trunc);
    if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: cannot write " + fn.path);
    std::vector<float> buf(x * y * z);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (float)data[i];
    if (isMrcExt(fn.extension())) {
        int32_t h[256];
        memset(h, 0, sizeof(h));
        h[0] = (int32_t)x; h[1] = (int32_t)y; h[2] = (int32_t)z; h[3] = 2; h[7] = (int32_t)x; h[8] = (int32_t)y; h[9] = (int32_t)z;
        float *hf = (float *)h;
        hf[10] = (float)x; hf[11] = (float)y; hf[12] = (float)z; hf[13] = hf[14] = hf[15] = 90.f;
        h[16] = 1; h[17] = 2; h[18] = 3;
        memcpy(&h[52], "MAP ", 4);
        unsigned char stamp[4] = {0x44, 0x44, 0, 0};
        memcpy(&h[53], stamp, 4);
        f.write((char *)h, 1024);
    } else {
        std::vector<float> h;
        spiderHeader(h, x, y, z, z > 1 ? 3 : 1, 0, 0, 0);
        f.write((char *)h.data(), h.size() * 4);
    }
    f.write((char *)buf.data(), buf.size() * 4);
}

// writes a Spider stack of n images (float)
inline void writeStack(const std::string &path, const float *data, size_t x, size_t y, size_t n)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: cannot write " + path);
    std::vector<float> h;
    spiderHeader(h, x, y, 1, 1, 2, n, 0);
    f.write((char *)h.data(), h.size() * 4);
    for (size_t i = 0; i < n; ++i) {
        spiderHeader(h, x, y, 1, 1, 0, 0, i + 1);
        f.write((char *)h.data(), h.size() * 4);
        f.write((char *)(data + i * x * y), x * y * 4);
    }
}

// A stack written image by image (Image::write(..., index, true, WRITE_REPLACE) of the reference, e.g. ctf_correct_wiener2d.cpp's
// XmippMetadataProgram loop and movie_alignment_correlation_gpu.cpp:537-541): the file is created with room for n images, every
// write() puts one image into its slot, slots never written stay zero images.  Spider stack, or an MRC stack by extension.
class StackWriter {
    std::fstream f;
    size_t x = 0, y = 0, n = 0, hdr = 0, perHdr = 0;
    bool mrc = false;
    std::vector<bool> written;
    std::vector<float> h;

public:
    StackWriter(const std::string &path, size_t x_, size_t y_, size_t n_) : x(x_), y(y_), n(n_), written(n_, false)
    {
        FileName fn(path);
        mrc = isMrcExt(fn.extension());
        { std::ofstream c(fn.path, std::ios::binary | std::ios::trunc); if (!c.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: cannot write " + fn.path); }
        f.open(fn.path, std::ios::binary | std::ios::in | std::ios::out);
        if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: cannot write " + fn.path);
        if (mrc) {
            int32_t m[256];
            memset(m, 0, sizeof(m));
            m[0] = (int32_t)x; m[1] = (int32_t)y; m[2] = (int32_t)n; m[3] = 2; m[7] = (int32_t)x; m[8] = (int32_t)y; m[9] = (int32_t)n;
            float *hf = (float *)m;
            hf[10] = (float)x; hf[11] = (float)y; hf[12] = (float)n; hf[13] = hf[14] = hf[15] = 90.f;
            m[16] = 1; m[17] = 2; m[18] = 3;
            memcpy(&m[52], "MAP ", 4);
            unsigned char stamp[4] = {0x44, 0x44, 0, 0};
            memcpy(&m[53], stamp, 4);
            f.write((char *)m, 1024);
            hdr = 1024; perHdr = 0;
        } else {
            spiderHeader(h, x, y, 1, 1, 2, n, 0);
            f.write((char *)h.data(), h.size() * 4);
            hdr = perHdr = h.size() * 4;
        }
    }
    // image i (0-based) of the stack
    void write(size_t i, const float *data)
    {
        if (i >= n) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: image " + std::to_string(i + 1) + " beyond the stack");
        f.seekp((std::streamoff)(hdr + i * (perHdr + x * y * 4)));
        if (!mrc) { spiderHeader(h, x, y, 1, 1, 0, 0, i + 1); f.write((char *)h.data(), h.size() * 4); }
        f.write((const char *)data, x * y * 4);
        if (!f.good()) REPORT_ERROR(ERR_IO_NOREAD, "Image::write: short write");
        written[i] = true;
    }
    // zero images into the slots nobody wrote
    void finish()
    {
        std::vector<float> zero;
        for (size_t i = 0; i < n; ++i)
            if (!written[i]) { if (zero.empty()) zero.assign(x * y, 0.f); write(i, zero.data()); }
        f.flush();
    }
};

// ------------------------------------------------------------------ symmetries (cyclic groups)
class SymList {
    // rotation by 360/fold about axis (unit-normalised), Rodrigues
    static std::vector<double> rotAxis(double ang, double x, double y, double z)
    {
        const double n = std::sqrt(x * x + y * y + z * z);
        x /= n; y /= n; z /= n;
        const double c = std::cos(ang), s = std::sin(ang), t = 1 - c;
        return {t * x * x + c,     t * x * y - s * z, t * x * z + s * y,
                t * x * y + s * z, t * y * y + c,     t * y * z - s * x,
                t * x * z - s * y, t * y * z + s * x, t * z * z + c};
    }
    static std::vector<double> mul(const std::vector<double> &A, const std::vector<double> &B)
    {
        std::vector<double> C(9, 0.);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) C[i * 3 + j] += A[i * 3 + k] * B[k * 3 + j];
        return C;
    }
    static bool same(const std::vector<double> &A, const std::vector<double> &B)
    {
        for (int i = 0; i < 9; ++i)
            if (std::fabs(A[i] - B[i]) > 1e-6) return false;
        return true;
    }
    // closure of the group generated by `gens` (what SymList::computeSubgroup does in xmippCore); identity excluded
    void close(const std::vector<std::vector<double>> &gens)
    {
        const std::vector<double> I = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::vector<std::vector<double>> G = {I};
        bool grew = true;
        while (grew) {
            grew = false;
            const size_t n0 = G.size();
            for (size_t a = 0; a < n0; ++a)
                for (const auto &g : gens) {
                    const auto P = mul(G[a], g);
                    bool known = false;
                    for (const auto &h : G)
                        if (same(h, P)) { known = true; break; }
                    if (!known) { G.push_back(P); grew = true; }
                    if (G.size() > 240) REPORT_ERROR(ERR_VALUE_INCORRECT, "SymList: the generators do not close into a finite point group");
                }
        }
        R.assign(G.begin() + 1, G.end());
    }
    static std::vector<double> mirrorPlane(double x, double y, double z)
    {
        // reflection in the plane through the origin with normal (x, y, z): I - 2 n n^T (xmippCore builds it as
        // A diag(1,1,-1) A^-1 with A turning Z onto the normal, which is the same matrix)
        const double n = std::sqrt(x * x + y * y + z * z);
        x /= n; y /= n; z /= n;
        return {1 - 2 * x * x, -2 * x * y, -2 * x * z, -2 * x * y, 1 - 2 * y * y, -2 * y * z, -2 * x * z, -2 * y * z, 1 - 2 * z * z};
    }
    static int leadingInt(const std::string &s, size_t from, size_t &end)
    {
        end = from;
        while (end < s.size() && ::isdigit((unsigned char)s[end])) ++end;
        return end > from ? atoi(s.substr(from, end - from).c_str()) : -1;
    }
public:
    std::vector<std::vector<double>> R;   // 3x3 row-major, identity excluded (as SL.getMatrices)
    // Accepts what the reference's --sym takes (RFA:243-251): a point-group name or a symmetry file with
    // `rot_axis <fold> <x> <y> <z>`, `mirror_plane <x> <y> <z>` and `inversion` lines (xmippCore
    // SymList::readSymmetryFile). Only the SET of matrices matters to the callers (reconstruction sums over the
    // group, the sampling takes neighbourhoods over it). Built-in names: cN, dN (N-fold about Z and, for dN, a
    // 2-fold about X -- Scipion's "dihedral X" convention for Xmipp; the dead createSymFile of
    // sampling.cpp:1361 writes Y instead, which is the same group only for even N), t, o, i1..i4, and the
    // groups with improper elements ci, cs, cNv, cNh, sN, dNv, dNh, td, th, oh, ih (= i2h), i1h..i4h with
    // the generators of Sampling::createSymFile (sampling.cpp:1328-1420). For those R holds matrices of
    // determinant -1 as well; the left matrices L of xmippCore are not kept: the reference's reconstruction
    // reads R only (reconstruct_fourier_accel.cpp:252-254) and its sampling fixtures are met with L = I
    // (tests/test_sampling.py).
    void readSymmetryFile(const std::string &sym)
    {
        R.clear();
        std::ifstream f(sym);
        if (f.good()) {
            std::vector<std::vector<double>> gens;
            std::string line;
            while (std::getline(f, line)) {
                line = trim(line);
                if (line.empty() || line[0] == '#' || line[0] == ';') continue;
                std::istringstream is(line);
                std::string kw;
                is >> kw;
                double x = 0, y = 0, z = 0;
                if (kw == "rot_axis") {
                    int fold = 0;
                    if (!(is >> fold >> x >> y >> z) || fold < 2 || (x == 0 && y == 0 && z == 0))
                        REPORT_ERROR(ERR_VALUE_INCORRECT, "SymList: bad line in " + sym + ": " + line);
                    gens.push_back(rotAxis(2 * M_PI / fold, x, y, z));
                } else if (kw == "mirror_plane") {
                    if (!(is >> x >> y >> z) || (x == 0 && y == 0 && z == 0))
                        REPORT_ERROR(ERR_VALUE_INCORRECT, "SymList: bad line in " + sym + ": " + line);
                    gens.push_back(mirrorPlane(x, y, z));
                } else if (kw == "inversion")
                    gens.push_back({-1, 0, 0, 0, -1, 0, 0, 0, -1});
                else
                    REPORT_ERROR(ERR_NOT_IMPLEMENTED, "SymList: '" + kw + "' in " + sym + " (rot_axis, mirror_plane and inversion are supported)");
            }
            close(gens);
            return;
        }
        std::string s = sym;
        std::transform(s.begin(), s.end(), s.begin(), ::tolower);
        const std::vector<double> inversion = {-1, 0, 0, 0, -1, 0, 0, 0, -1};
        if (s == "ci") { close({inversion}); return; }
        if (s == "cs") { close({mirrorPlane(0, 0, 1)}); return; }
        if (s.size() >= 2 && (s[0] == 'c' || s[0] == 'd' || s[0] == 's') && ::isdigit((unsigned char)s[1])) {
            size_t end;
            const int n = leadingInt(s, 1, end);
            const std::string tail = s.substr(end);
            if (n < 1 || !(tail.empty() || ((tail == "v" || tail == "h") && s[0] != 's')) || (s[0] == 's' && (n % 2 || !tail.empty())))
                REPORT_ERROR(ERR_ARG_INCORRECT, "SymList: bad symmetry " + sym);
            std::vector<std::vector<double>> gens;
            if (s[0] == 's') {           // sampling.cpp:1347-1353: the N/2-fold axis and the inversion
                if (n / 2 > 1) gens.push_back(rotAxis(2 * M_PI / (n / 2), 0, 0, 1));
                gens.push_back(inversion);
            } else {
                if (n > 1) gens.push_back(rotAxis(2 * M_PI / n, 0, 0, 1));
                if (s[0] == 'd') gens.push_back(rotAxis(M_PI, 1, 0, 0));
                if (s[0] == 'c' && tail == "v") gens.push_back(mirrorPlane(0, 1, 0));
                if (s[0] == 'c' && tail == "h") gens.push_back(mirrorPlane(0, 0, 1));
                // dNv / dNh: with the 2-fold on X the mirror that makes removeRedundantPoints' wedges
                // (sampling.cpp:780-806) fundamental domains for every N is x = 0 / z = 0 (for even N the
                // same groups as createSymFile's, sampling.cpp:1362-1377; checked numerically in the tests)
                if (s[0] == 'd' && tail == "v") gens.push_back(mirrorPlane(1, 0, 0));
                if (s[0] == 'd' && tail == "h") gens.push_back(mirrorPlane(0, 0, 1));
            }
            close(gens);
            return;
        }
        // Cubic groups. Generators as written by Sampling::createSymFile (data/sampling.cpp:1377-1416; the
        // 6-digit axes there are sqrt(2/3), 1/sqrt(3), the golden ratio ...). i1, i3, i4 are i2 turned by
        // Euler(0, 90 | 31.7174745559 | -31.7174745559, 0): the only orientations for which the asymmetric
        // units of Sampling::removeRedundantPoints (sampling.cpp:985-1069) are fundamental domains
        // (checked numerically, tests/test_sampling.py).
        const double phi = (1 + std::sqrt(5.0)) / 2;
        if (s == "t") { close({rotAxis(2 * M_PI / 3, 0, 0, 1), rotAxis(M_PI, 0, std::sqrt(2.0 / 3.0), std::sqrt(1.0 / 3.0))}); return; }
        if (s == "td") { close({rotAxis(2 * M_PI / 3, 0, 0, 1), rotAxis(M_PI, 0, std::sqrt(2.0 / 3.0), std::sqrt(1.0 / 3.0)), mirrorPlane(std::sqrt(2.0), std::sqrt(6.0), 0)}); return; }
        if (s == "th") { close({rotAxis(2 * M_PI / 3, 0, 0, 1), rotAxis(M_PI, 0, -std::sqrt(2.0 / 3.0), -std::sqrt(1.0 / 3.0)), inversion}); return; }
        if (s == "o") { close({rotAxis(2 * M_PI / 3, 1, 1, 1), rotAxis(M_PI / 2, 0, 0, 1)}); return; }
        if (s == "oh") { close({rotAxis(2 * M_PI / 3, 1, 1, 1), rotAxis(M_PI / 2, 0, 0, 1), mirrorPlane(0, 1, 1)}); return; }
        const bool ih = s == "ih" || s == "i1h" || s == "i2h" || s == "i3h" || s == "i4h";
        if (s == "i" || s == "i1" || s == "i2" || s == "i3" || s == "i4" || ih) {
            std::vector<std::vector<double>> gens = {rotAxis(M_PI, 0, 0, 1), rotAxis(2 * M_PI / 5, -phi, -1, 0), rotAxis(2 * M_PI / 3, -1, -phi * phi, 0)};
            if (ih) gens.push_back(mirrorPlane(1, 0, 0));
            close(gens);
            const char o = s.size() > 1 && ::isdigit((unsigned char)s[1]) ? s[1] : '2';
            const double tilt = o == '1' ? 90. : o == '3' ? 31.7174745559 : o == '4' ? -31.7174745559 : 0.;
            if (tilt != 0.) {
                // Euler_angles2matrix(0, tilt, 0): rotation about Y
                const double b = tilt * M_PI / 180., cb = std::cos(b), sb = std::sin(b);
                const std::vector<double> A = {cb, 0, -sb, 0, 1, 0, sb, 0, cb}, At = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
                for (auto &g : R) g = mul(mul(A, g), At);
            }
            return;
        }
        REPORT_ERROR(ERR_NOT_IMPLEMENTED, "SymList: symmetry '" + sym + "' is neither a readable symmetry file nor one of cN, cNv, cNh, ci, cs, sN, "
                     "dN, dNv, dNh, t, td, th, o, oh, i1..i4, ih, i1h..i4h; write the group's rot_axis / mirror_plane / inversion lines to a file and pass that");
    }
    bool hasImproper() const
    {
        for (const auto &g : R) {
            const double det = g[0] * (g[4] * g[8] - g[5] * g[7]) - g[1] * (g[3] * g[8] - g[5] * g[6]) + g[2] * (g[3] * g[7] - g[4] * g[6]);
            if (det < 0) return true;
        }
        return false;
    }
    int symsNo() const { return (int)R.size(); }
};

// ------------------------------------------------------------------ XmippProgram
class XmippProgram {
    struct ParamDef { std::string name; std::vector<std::string> aliases; std::vector<std::string> argDefaults; std::vector<bool> argHasDefault; bool optional = false; std::string help; int group = -1; };
    std::vector<ParamDef> defs;
    std::map<std::string, std::vector<std::string>> given;
    std::vector<std::string> usage, examples;
    std::string progName;

    ParamDef *find(const std::string &n)
    {
        for (auto &d : defs) { if (d.name == n) return &d; for (auto &a : d.aliases) if (a == n) return &d; }
        return nullptr;
    }
protected:
    virtual void defineParams() = 0;
    virtual void readParams() = 0;
public:
    int verbose = 1;
    int errorCode = 0;
    virtual ~XmippProgram() {}
    virtual void run() = 0;
    void addUsageLine(const std::string &l) { usage.push_back(l); }
    void addSeeAlsoLine(const std::string &) {}
    void addExampleLine(const std::string &l, bool = true) { examples.push_back(l); }

    // subset of the xmippCore DSL: "  [--opt <a=1> <b>] : help", "  alias --other;", "==Section==", " : more help"
    void addParamsLine(const std::string &line)
    {
        std::string t = trim(line);
        if (t.empty() || t.rfind("==", 0) == 0) return;
        if (t[0] == ':') { if (!defs.empty()) defs.back().help += " " + trim(t.substr(1)); return; }
        if (t.rfind("alias", 0) == 0) {
            std::string a = trim(t.substr(5));
            if (!a.empty() && a.back() == ';') a.pop_back();
            if (!defs.empty()) defs.back().aliases.push_back(trim(a));
            return;
        }
        if (t.rfind("where", 0) == 0 || t.rfind("requires", 0) == 0) return;
        if (t.rfind("or ", 0) == 0 && !defs.empty()) {   // "or --other <arg>": alternative to the previous parameter
            const size_t prev = defs.size() - 1;
            addParamsLine(t.substr(3));
            if (defs.size() == prev + 2) {
                if (defs[prev].group < 0) defs[prev].group = (int)prev;
                defs.back().group = defs[prev].group;
                defs.back().optional = defs[prev].optional;
            }
            return;
        }
        ParamDef d;
        size_t colon = std::string::npos;
        {   // the help separator is the first ':' outside <...>
            int depth = 0;
            for (size_t i = 0; i < t.size(); ++i) { if (t[i] == '<') ++depth; else if (t[i] == '>') --depth; else if (t[i] == ':' && depth == 0) { colon = i; break; } }
        }
        std::string spec = trim(colon == std::string::npos ? t : t.substr(0, colon));
        if (colon != std::string::npos) d.help = trim(t.substr(colon + 1));
        if (!spec.empty() && spec[0] == '[') { d.optional = true; size_t e = spec.rfind(']'); spec = trim(spec.substr(1, e == std::string::npos ? std::string::npos : e - 1)); }
        if (spec.empty() || spec[0] != '-') return;
        size_t sp = spec.find_first_of(" \t");
        d.name = spec.substr(0, sp);
        std::string rest = sp == std::string::npos ? "" : spec.substr(sp);
        size_t p = 0;
        while ((p = rest.find('<', p)) != std::string::npos) {
            size_t e = rest.find('>', p);
            std::string a = rest.substr(p + 1, e - p - 1);
            size_t eq = a.find('=');
            if (eq == std::string::npos) { d.argDefaults.push_back(""); d.argHasDefault.push_back(false); }
            else {
                std::string v = trim(a.substr(eq + 1));
                if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
                d.argDefaults.push_back(v); d.argHasDefault.push_back(true);
            }
            p = e + 1;
        }
        defs.push_back(d);
    }

    void showUsage() const
    {
        std::cerr << "PROGRAM\n   " << progName << "\nUSAGE\n";
        for (auto &u : usage) std::cerr << "   " << u << "\n";
        std::cerr << "OPTIONS\n";
        for (auto &d : defs) {
            std::cerr << "   " << (d.group >= 0 && d.group != (int)(&d - defs.data()) ? "or " : "") << (d.optional ? "[" : "") << d.name;
            for (auto &a : d.aliases) std::cerr << ", " << a;
            for (size_t i = 0; i < d.argDefaults.size(); ++i) std::cerr << " <" << (d.argHasDefault[i] ? "=" + d.argDefaults[i] : "arg") << ">";
            std::cerr << (d.optional ? "]" : "") << " : " << d.help << "\n";
        }
        for (auto &e : examples) std::cerr << "   " << e << "\n";
    }

    void read(int argc, const char **argv)
    {
        progName = argc > 0 ? argv[0] : "xmipp_program";
        defs.clear(); given.clear();
        defineParams();
        addParamsLine("  [-v <verbose_level=1>] : Verbosity");
        addParamsLine("    alias --verbose;");
        try {
            for (int i = 1; i < argc; ++i) {
                std::string a = argv[i];
                if (a == "-h" || a == "--help") { showUsage(); errorCode = 0; throw XmippError(-1, ""); }
                bool isNeg = a.size() > 1 && a[0] == '-' && (isdigit((unsigned char)a[1]) || a[1] == '.');
                if (a.empty() || a[0] != '-' || isNeg) REPORT_ERROR(ERR_ARG_INCORRECT, "Unexpected argument '" + a + "'");
                ParamDef *d = find(a);
                if (!d) REPORT_ERROR(ERR_ARG_INCORRECT, "Unknown parameter '" + a + "'");
                std::vector<std::string> vals;
                while (i + 1 < argc) {
                    std::string n = argv[i + 1];
                    bool neg = n.size() > 1 && n[0] == '-' && (isdigit((unsigned char)n[1]) || n[1] == '.');
                    if (!n.empty() && n[0] == '-' && !neg) break;
                    if (vals.size() >= d->argDefaults.size()) break;
                    vals.push_back(n);
                    ++i;
                }
                given[d->name] = vals;
            }
            for (auto &d : defs) {
                if (d.optional || given.count(d.name)) continue;
                bool alt = false;
                if (d.group >= 0)
                    for (auto &o : defs) alt = alt || (o.group == d.group && given.count(o.name));
                if (!alt) REPORT_ERROR(ERR_ARG_MISSING, "Parameter " + d.name + " is mandatory");
            }
            long v = getIntParam("-v");
            verbose = (int)v;
            readParams();
        } catch (XmippError &e) {
            if (e.code == -1) { errorCode = -1; return; }
            std::cerr << "XMIPP_ERROR " << e.code << ": " << e.what() << std::endl;
            errorCode = e.code;
        }
    }
    void read(int argc, char **argv) { read(argc, (const char **)argv); }

    bool checkParam(const std::string &n) { ParamDef *d = find(n); return d && given.count(d->name); }
    std::string getParam(const std::string &n, int idx = 0)
    {
        ParamDef *d = find(n);
        if (!d) REPORT_ERROR(ERR_ARG_INCORRECT, "getParam: undefined parameter " + n);
        auto it = given.find(d->name);
        if (it != given.end() && idx < (int)it->second.size()) return it->second[idx];
        if (idx < (int)d->argDefaults.size() && d->argHasDefault[idx]) return d->argDefaults[idx];
        REPORT_ERROR(ERR_ARG_MISSING, "Parameter " + n + " needs a value");
    }
    long getIntParam(const std::string &n, int idx = 0) { return atol(getParam(n, idx).c_str()); }
    double getDoubleParam(const std::string &n, int idx = 0) { return atof(getParam(n, idx).c_str()); }

    // XmippProgram::tryRun: run(), map XmippError to "XMIPP_ERROR" on stderr + error code
    int tryRun()
    {
        if (errorCode == -1) return 0;       // --help
        if (errorCode != 0) return errorCode;
        try { run(); }
        catch (XmippError &e) { std::cerr << "XMIPP_ERROR " << e.code << ": " << e.what() << std::endl; errorCode = e.code; }
        catch (std::exception &e) { std::cerr << "XMIPP_ERROR 1: " << e.what() << std::endl; errorCode = 1; }
        return errorCode;
    }
};

inline void init_progress_bar(size_t) {}
inline void progress_bar(size_t) {}

}  // namespace mc
#endif
