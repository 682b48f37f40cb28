// programs.h -- C++ mirrors of the two Xmipp programs on the hot path, over the C ABI.
//
//   ProgAngularProjectionMatching  reconstruction/angular_projection_matching.{h,cpp}
//   ProgRecFourierAccel            reconstruction/reconstruct_fourier_accel.{h,cpp}
//
// Same class names, same virtual seams (defineParams / readParams / show / run, and for APM
// produceSideInfo / processAllImages / processSomeImages / writeOutputFiles), same parameter
// strings, same output labels.  All numerics happen behind include/xmipp_hip.h; there is no
// CPU path here.  Extra flags: --device <id>, --gpus <n>, --devices <list>, --batch <n>.
#ifndef XMIPP3_AMD_PROGRAMS_H
#define XMIPP3_AMD_PROGRAMS_H
#include "minicore.h"
#include "fastio.h"
#include "sampling_gen.h"
#include <chrono>
#include <exception>
#include <future>
#include <thread>

namespace mc {

// --device <id> / --gpus <n> / --devices <list>: the HIP devices one program drives, one host thread each
// (cf. the --device/--gpus flags of reconstruct_fourier_gpu, RFG:56-57). --devices may repeat an id.
inline std::vector<int> parseDevices(int device, int gpus, const std::string &list)
{
    std::vector<int> d;
    if (!list.empty()) {
        std::string tok;
        std::istringstream is(list);
        while (std::getline(is, tok, ',')) {
            tok = trim(tok);
            if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos)
                REPORT_ERROR(ERR_ARG_INCORRECT, "--devices expects a comma-separated list of device ids, got '" + list + "'");
            d.push_back(atoi(tok.c_str()));
        }
    } else {
        if (gpus < 1) REPORT_ERROR(ERR_ARG_INCORRECT, "--gpus must be at least 1");
        for (int g = 0; g < gpus; ++g) d.push_back(device + g);
    }
    if (d.empty()) REPORT_ERROR(ERR_ARG_INCORRECT, "no device selected");
    int count = 0;
    if (xh_device_count(&count) == XH_OK)
        for (int id : d)
            if (id < 0 || id >= count)
                REPORT_ERROR(ERR_ARG_INCORRECT, "device " + std::to_string(id) + " requested but this node has " + std::to_string(count));
    return d;
}

struct DeviceBuffer {
    xh_ctx *ctx = nullptr; void *p = nullptr; size_t bytes = 0;
    ~DeviceBuffer() { release(); }
    void release() { if (p) xh_free(ctx, p); p = nullptr; bytes = 0; }
    void reserve(xh_ctx *c, size_t b) { if (b <= bytes) return; release(); ctx = c; xhCheck(xh_malloc(c, b, &p)); bytes = b; }
    template <typename T> T *as() { return (T *)p; }
};

// CTFDescription::getValueAt for the pure CTF (data/ctf.h:452-496,1002-1029; data/ctf.cpp:645-679,1392-1402);
// host copy used only to build the gallery filter of --ctf (generateCTF, data/ctf.h:1219-1240)
inline double bessj0_host(double x)
{
    double ax = std::fabs(x);
    if (ax < 8.0) {
        double y = x * x;
        double a1 = 57568490574.0 + y * (-13362590354.0 + y * (651619640.7 + y * (-11214424.18 + y * (77392.33017 + y * (-184.9052456)))));
        double a2 = 57568490411.0 + y * (1029532985.0 + y * (9494680.718 + y * (59272.64853 + y * (267.8532712 + y * 1.0))));
        return a1 / a2;
    }
    double z = 8.0 / ax, y = z * z, xx = ax - 0.785398164;
    double a1 = 1.0 + y * (-0.1098628627e-2 + y * (0.2734510407e-4 + y * (-0.2073370639e-5 + y * 0.2093887211e-6)));
    double a2 = -0.1562499995e-1 + y * (0.1430488765e-3 + y * (-0.6911147651e-5 + y * (0.7621095161e-6 - y * 0.934935152e-7)));
    return std::sqrt(0.636619772 / ax) * (std::cos(xx) * a1 - z * std::sin(xx) * a2);
}
inline double ctfValueAt(const xh_ctf_params &c, double X, double Y)
{
    const double PI = 3.14159265358979323846;
    const double local_Cs = c.Cs * 1e7, local_Ca = c.Ca * 1e7, local_kV = c.kV * 1e3, local_ispr = c.ispr * 1e6;
    const double lambda = 12.2643247 / std::sqrt(local_kV * (1. + 0.978466e-6 * local_kV));
    const double K1 = PI * lambda, K2 = PI / 2 * local_Cs * lambda * lambda * lambda;
    const double K3 = std::pow(0.25 * PI * local_Ca * lambda * (c.espr / c.kV + 2 * local_ispr), 2) / std::log(2.0);
    const double K5 = PI * c.DeltaF * lambda, K6 = PI * PI * c.alpha * c.alpha, K7 = local_Cs * lambda * lambda;
    const double Ksin = std::sqrt(1 - c.Q0 * c.Q0), Kcos = c.Q0;
    const double ang = std::atan2(Y, X), u2 = X * X + Y * Y, u = std::sqrt(u2), u4 = u2 * u2;
    double deltaf = 0;
    if (!(std::fabs(X) < 1e-6 && std::fabs(Y) < 1e-6))
        deltaf = -(c.DeltafU + c.DeltafV) * 0.5 - (c.DeltafU - c.DeltafV) * 0.5 * std::cos(2 * (ang - c.azimuthal_angle * PI / 180.));
    double VPP = 0;
    if (std::round(c.VPP_radius * 1000) != 0) VPP = -c.phase_shift * (1 - std::exp(-u2 / (2 * c.VPP_radius * c.VPP_radius)));
    const double arg = VPP + K1 * deltaf * u2 + K2 * u4;
    const double xs = u * c.DeltaR;
    const double aux = K7 * u2 * u + deltaf * u;
    double E = std::exp(-K3 * u4) * bessj0_host(K5 * u2) * (xs == 0 ? 1.0 : std::sin(PI * xs) / (PI * xs)) * std::exp(-K6 * aux * aux) + c.envR0 +
               c.envR1 * u + c.envR2 * u2;
    if (E < 0) E = 0;
    return -c.K * (Ksin * std::sin(arg) - Kcos * std::cos(arg)) * E;
}

// Sampling::readSamplingFile (data/sampling.cpp:1592-1659): blocks extra / neighbors / projectionDirections.
// my_neighbors[image] is kept as shared lists (fastio.h: NeighbourLists): with every image searching the whole gallery
// the block is 5 KB of text per image, read in place by several threads.
struct Sampling {
    NeighbourLists my_neighbors;
    std::vector<size_t> no_redundant_sampling_points_index;
    std::vector<std::vector<double>> no_redundant_sampling_points_angles;
    size_t numberSamplesAsymmetricUnit = 0;
    void readSamplingFile(const std::string &base, int threads = 0)
    {
        auto f = std::make_shared<MappedFile>(base + "_sampling.xmd");
        FastTable md;
        md.read(f, "extra");
        if (md.size()) numberSamplesAsymmetricUnit = (size_t)std::max(0L, md.getLong(md.col("pointsAsymmetricUnit"), 0, 0));
        md.read(f, "neighbors", threads);
        parseNeighbourRows(md, md.col("neighbors"), my_neighbors, threads);
        md.read(f, "projectionDirections", threads);
        no_redundant_sampling_points_index.resize(md.size());
        no_redundant_sampling_points_angles.resize(md.size());
        const int cn = md.col("neighbor"), cr = md.col("angleRot"), ct = md.col("angleTilt"), cp = md.col("anglePsi");
        for (size_t i = 0; i < md.size(); ++i) {
            const long idx = md.getLong(cn, i, 0);
            no_redundant_sampling_points_index[i] = (size_t)idx;
            no_redundant_sampling_points_angles[i] = {md.getDouble(cr, i, 0), md.getDouble(ct, i, 0), md.getDouble(cp, i, 0)};
            numberSamplesAsymmetricUnit = std::max(numberSamplesAsymmetricUnit, (size_t)idx + 1);
        }
    }
};

// ============================================================================================
class ProgAngularProjectionMatching : public XmippProgram {
public:
    std::string fn_exp, fn_out, fn_ref, fn_ctf;
    double pad = 1, max_shift = -1, avail_memory = 1;
    int Ri = 1, Ro = -1, search5d_shift = 0, search5d_step = 2, numOrientations = 1, threads = 1;
    bool phase_flipped = false, do_scale = false, do_append = false;
    int device = 0, gpus = 1, batch = 4096, readers = 0;
    std::string deviceList;
    // side info
    FastTable DFexp;
    MetaDataVec DFo;
    Sampling mysampling;
    std::vector<int> convert_refno_to_stack_position;
    std::vector<int32_t> stackPositions;          // mysampling.my_neighbors.ids through convert_refno_to_stack_position
    std::vector<uint8_t> listIsWholeGallery;      // per neighbour list: every reference, in stack order
    HostTiming timing;
    std::mutex timingMutex;
    std::vector<int32_t> search5d_xoff, search5d_yoff;
    size_t dim = 0, total_nr_refs = 0;
    bool loop_forward_refs = true;
    // one slot per device, each with the whole reference bank (particles are independent)
    struct Slot {
        int device = 0;
        xh_ctx *ctx = nullptr;
        xh_pm *pm = nullptr;
        xh_rf *shifter = nullptr;   // only used for xh_rf_shift_images (previous shifts, APM:1228-1233)
        std::unique_ptr<BatchFeeder> feeder;   // loader threads + page-locked pieces + the two device batches (fastio.h)
    };
    std::vector<Slot> slots;
    int N = 0;

    ~ProgAngularProjectionMatching() override
    {
        for (Slot &s : slots) {
            s.feeder.reset();
            if (s.pm) xh_pm_destroy(s.pm);
            if (s.shifter) xh_rf_destroy(s.shifter);
            if (s.ctx) xh_ctx_destroy(s.ctx);
        }
    }

    void defineParams() override
    {
        // APM:83-121, verbatim parameter lines
        addUsageLine("Perform a discrete angular assignment using projection matching in real space.");
        addUsageLine("This program is relatively fast, using polar coordinates for the in-plane ");
        addUsageLine("angular searches and the 5-dimensional search of rotation angles and origin ");
        addUsageLine("offsets is broken in two: first the angles are search in a 3D-search; then, ");
        addUsageLine("for the optimal orientation the origin offsets are searched (2D).");
        addExampleLine("xmipp_angular_projection_matching -i experimental.doc -o assigned_angles.doc --ref reference.stk --search5d_step 2");
        addParamsLine("   -i <doc_file>                : Docfile with input images");
        addParamsLine("   -o <output_filename>         : Output filename");
        addParamsLine("   -r <stackFile>               : Reference projections");
        addParamsLine("     alias --ref;");
        addParamsLine("  [--search5d_shift <s5dshift=0>]: Search range (in +/- pix) for 5D shift search");
        addParamsLine("  [--search5d_step <s5dstep=2>]  : Step size for 5D shift search (in pix)");
        addParamsLine("  [--Ri <ri=1>]               : Inner radius to limit rotational search");
        addParamsLine("  [--Ro <ro=-1>]              : Outer radius to limit rotational search");
        addParamsLine("                        : ro = -1 -> dim/2-1");
        addParamsLine("  [-s <step=1> <n_steps=3>]    : scale step factor (1 means 0.01 in/de-crements) and number of steps around 1.");
        addParamsLine("                               : with default values: 1 0.01 | 0.02 | 0.03");
        addParamsLine("    alias --scale;");
        addParamsLine("==+Extra parameters==");
        addParamsLine("  [--mem <mem=1>]             : Available memory for reference library (Gb)");
        addParamsLine("  [--max_shift <max_shift=-1>]   : Max. change in origin offset (+/- pixels; neg= no limit)");
        addParamsLine("  [--ctf <filename>]            : CTF to apply to the reference projections, either a");
        addParamsLine("                     : CTF parameter file or a 2D image with the CTF amplitudes");
        addParamsLine("  [--pad <pad=1>]             : Padding factor (for CTF correction only)");
        addParamsLine("  [--phase_flipped]            : Use this if the experimental images have been phase flipped");
        addParamsLine("  [--thr <threads=1>]           : Number of concurrent threads of the reference. The search runs on the device whatever");
        addParamsLine("                               : the value; it only decides which of two EXACTLY equal correlation values is kept, as the");
        addParamsLine("                               : reference's split of a neighbour list over its threads and the merge of their results do");
        addParamsLine("                               : (up to 16)");
        addParamsLine("  [--number_orientations <numOrientations=1>]  : Number of possible orientations for each experimental image");
        addParamsLine("  [--append]                : Append (versus overwrite) data to the output file");
        addParamsLine("  [--device <id=0>]         : first HIP device");
        addParamsLine("  [--gpus <n=1>]            : number of consecutive HIP devices, one host thread each");
        addParamsLine("  [--devices <list=\"\">]    : explicit comma-separated device ids (overrides --device/--gpus)");
        addParamsLine("  [--batch <n=4096>]        : Particles per device batch");
        addParamsLine("  [--readers <n=0>]         : Host threads reading images into page-locked memory (0: half the cores, at most 16)");
    }

    void readParams() override
    {
        // APM:44-80
        fn_exp = getParam("-i");
        fn_out = getParam("-o");
        fn_ref = getParam("--ref");
        pad = std::max(1., getDoubleParam("--pad"));
        Ri = (int)getIntParam("--Ri");
        Ro = (int)getIntParam("--Ro");
        search5d_shift = (int)getIntParam("--search5d_shift");
        search5d_step = (int)getIntParam("--search5d_step");
        max_shift = getDoubleParam("--max_shift");
        numOrientations = (int)getIntParam("--number_orientations");
        avail_memory = getDoubleParam("--mem");
        if (checkParam("--ctf")) fn_ctf = getParam("--ctf");
        phase_flipped = checkParam("--phase_flipped");
        threads = (int)getIntParam("--thr");
        do_scale = checkParam("--scale");
        do_append = checkParam("--append");
        device = (int)getIntParam("--device");
        gpus = (int)getIntParam("--gpus");
        deviceList = getParam("--devices");
        batch = std::max(1, (int)getIntParam("--batch"));
        readers = std::max(0, (int)getIntParam("--readers"));
    }

    void show()
    {
        if (!verbose) return;
        std::cout << "  Input images            : " << fn_exp << " (" << DFexp.size() << ")\n"
                  << "  Reference projections   : " << fn_ref << " (" << total_nr_refs << ")\n"
                  << "  Output rootname         : " << fn_out << "\n"
                  << "  Inner radius rot-search : " << Ri << "\n  Outer radius rot-search : " << Ro << "\n"
                  << "  Max. shift              : " << max_shift << "\n"
                  << "  Devices                 : ";
        for (size_t g = 0; g < slots.size(); ++g) std::cout << (g ? "," : "") << slots[g].device;
        std::cout << " (" << xh_version() << ")\n";
    }

    virtual void produceSideInfo()
    {
        // APM:209-404
        if (numOrientations < 1 || numOrientations > 16)
            REPORT_ERROR(ERR_VALUE_INCORRECT, "--number_orientations must be in [1,16] on the device path");
        // 5-D search translations, origin included (APM:321-348)
        if (search5d_step == 0) {
            std::cout << "*   ERROR: search step should be different from 0\n*   search step set to 1 \n";
            search5d_step = 1;
        }
        search5d_xoff.clear();
        search5d_yoff.clear();
        {
            const int myfinal = search5d_shift + search5d_shift % search5d_step;
            for (int xoff = -myfinal; xoff <= myfinal; xoff += search5d_step)
                for (int yoff = -myfinal; yoff <= myfinal; yoff += search5d_step)
                    if (xoff * xoff + yoff * yoff <= search5d_shift * search5d_shift) {
                        search5d_xoff.push_back(xoff);
                        search5d_yoff.push_back(yoff);
                    }
        }
        if (do_scale) REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--scale is not supported (the reference loops forever at scale 1.0, APM:947-948)");
        const double tp0 = nowSeconds();
        DFexp.read(fn_exp, readers);
        if (DFexp.size() == 0) REPORT_ERROR(ERR_MD_NOOBJ, "Empty input metadata " + fn_exp);
        std::string fn_img;
        DFexp.getValue("image", fn_img, 0);
        ImageInfo info = readInfo(FileName(fn_img).path);
        dim = info.x;
        ImageInfo rinfo = readInfo(fn_ref);
        if (rinfo.x != dim || rinfo.y != info.y)
            REPORT_ERROR(ERR_MULTIDIM_SIZE, "Check that the reference volume and the experimental images are of the same size");
        if (max_shift < 0) max_shift = (double)(dim / 2);
        if (Ri < 1) Ri = 1;
        if (Ro < 0) Ro = (int)(dim / 2) - 1;
        mysampling.readSamplingFile(FileName(fn_ref).removeAllExtensions(), readers);
        total_nr_refs = mysampling.no_redundant_sampling_points_angles.size();
        if (rinfo.n < total_nr_refs) REPORT_ERROR(ERR_MULTIDIM_SIZE, "Reference stack holds fewer images than the sampling file lists");
        convert_refno_to_stack_position.assign(mysampling.numberSamplesAsymmetricUnit, -1);
        for (size_t i = 0; i < mysampling.no_redundant_sampling_points_index.size(); i++)
            convert_refno_to_stack_position[mysampling.no_redundant_sampling_points_index[i]] = (int)i;
        // the neighbour lists in stack positions, once per distinct list (the per-image loop of APM:1117 does it per image)
        {
            const NeighbourLists &nl = mysampling.my_neighbors;
            stackPositions.resize(nl.ids.size());
            for (size_t i = 0; i < nl.ids.size(); ++i) {
                const int32_t r = nl.ids[i];
                if (r < 0 || (size_t)r >= convert_refno_to_stack_position.size() || convert_refno_to_stack_position[(size_t)r] < 0)
                    REPORT_ERROR(ERR_VALUE_INCORRECT, "Wrong reference number " + std::to_string(r));
                stackPositions[i] = convert_refno_to_stack_position[(size_t)r];
            }
            listIsWholeGallery.assign(nl.span.size(), 0);
            for (size_t l = 0; l < nl.span.size(); ++l) {
                bool whole = nl.span[l].second == total_nr_refs;
                for (size_t j = 0; j < nl.span[l].second && whole; ++j) whole = stackPositions[nl.span[l].first + j] == (int32_t)j;
                listIsWholeGallery[l] = whole ? 1 : 0;
            }
        }
        timing.parse += nowSeconds() - tp0;
        loop_forward_refs = true;
        const double tb0 = nowSeconds();
        // reference library on the device, in stack order (getCurrentReference, APM:408-528)
        std::vector<float> refs(total_nr_refs * dim * dim);
        {
            StackSource src;
            std::vector<StackSource::Loc> locs(total_nr_refs);
            for (size_t r = 0; r < total_nr_refs; ++r) locs[r] = src.locate(std::to_string(r + 1) + "@" + fn_ref, dim);
            const size_t T = (size_t)std::max(1, std::min(hostThreads(readers), (int)(total_nr_refs / 32) + 1));
            runOnSlots(T, [&](size_t t) {
                std::vector<unsigned char> scratch;
                for (size_t r = total_nr_refs * t / T; r < total_nr_refs * (t + 1) / T; ++r) src.readFloats(locs[r], refs.data() + r * dim * dim, dim, scratch);
            });
        }
        // CTF filter of the gallery (APM:366-402): a 2-D image of amplitudes or a CTF parameter file
        std::vector<double> Mctf;
        const int paddim = (int)std::floor(pad * dim + 0.5);
        if (!fn_ctf.empty()) {
            const std::string ext = FileName(fn_ctf).extension();
            const bool isMd = ext == "xmd" || ext == "ctfparam" || ext == "doc" || ext == "sel";
            Mctf.resize((size_t)paddim * paddim);
            if (!isMd) {
                std::vector<float> im;
                ImageInfo ci;
                readImage(fn_ctf, im, ci);
                if ((int)ci.x != paddim) REPORT_ERROR(ERR_VALUE_INCORRECT, "Incompatible padding factor for this CTF filter");
                for (size_t i = 0; i < Mctf.size(); ++i) Mctf[i] = im[i];
            } else {
                MetaDataVec md;
                md.read(fn_ctf);
                xh_ctf_params c;
                xh_ctf_defaults(&c);
                c.Tm = md.getDouble("ctfSamplingRate", 0, 1); c.kV = md.getDouble("ctfVoltage", 0, 100);
                c.DeltafU = md.getDouble("ctfDefocusU", 0, 0); c.DeltafV = md.getDouble("ctfDefocusV", 0, c.DeltafU);
                c.azimuthal_angle = md.getDouble("ctfDefocusAngle", 0, 0); c.Cs = md.getDouble("ctfSphericalAberration", 0, 0);
                c.Ca = md.getDouble("ctfChromaticAberration", 0, 0); c.espr = md.getDouble("ctfEnergyLoss", 0, 0);
                c.ispr = md.getDouble("ctfLensStability", 0, 0); c.alpha = md.getDouble("ctfConvergenceCone", 0, 0);
                c.DeltaF = md.getDouble("ctfLongitudinalDisplacement", 0, 0); c.DeltaR = md.getDouble("ctfTransversalDisplacement", 0, 0);
                c.Q0 = md.getDouble("ctfQ0", 0, 0); c.K = md.getDouble("ctfK", 0, 1);
                if (std::fabs(c.DeltafV - c.DeltafU) > 1.) REPORT_ERROR(ERR_VALUE_INCORRECT, "ERROR!! Only non-astigmatic CTFs are allowed!");
                const double iTs = 1.0 / c.Tm;
                for (int i = 0; i < paddim; ++i) {
                    const double fy = (double)(i <= paddim / 2 ? i : i - paddim) / paddim * iTs;
                    for (int j = 0; j < paddim; ++j) {
                        const double fx = (double)(j <= paddim / 2 ? j : j - paddim) / paddim * iTs;
                        const double v = ctfValueAt(c, fx, fy);
                        Mctf[(size_t)i * paddim + j] = phase_flipped ? std::fabs(v) : v;
                    }
                }
            }
        }
        for (int d : parseDevices(device, gpus, deviceList)) {
            slots.emplace_back();
            slots.back().device = d;
        }
        runOnSlots(slots.size(), [&](size_t g) {
            Slot &s = slots[g];
            bindToDeviceNode(s.device);
            xhCheck(xh_ctx_create_private(s.device, &s.ctx));
            // the loader (page-locking its pieces takes tens of ms) is set up beside the reference bank
            s.feeder.reset(new BatchFeeder);
            auto feederReady = std::async(std::launch::async, [&] {
                s.feeder->create(s.device, dim, std::min((size_t)batch, DFexp.size()), std::max(1, hostThreads(readers) / (int)slots.size()), nullptr);
            });
            DeviceBuffer d_refs;
            d_refs.reserve(s.ctx, refs.size() * sizeof(float));
            xhCheck(xh_memcpy_h2d(s.ctx, d_refs.p, refs.data(), refs.size() * sizeof(float)));
            xhCheck(xh_pm_create(s.ctx, (int)dim, Ri, Ro, (int)total_nr_refs, d_refs.as<float>(), Mctf.empty() ? nullptr : Mctf.data(), paddim, &s.pm));
            // which of two exactly equal correlation values wins follows the reference's worker threads (APM:631,1063-1108)
            if (threads > 1) xhCheck(xh_pm_set_option(s.pm, "threads", (double)std::min(threads, 16)));
            feederReady.get();
        });
        int32_t nn;
        xhCheck(xh_pm_info(slots[0].pm, &nn, nullptr, nullptr));
        N = nn;
        timing.bank += nowSeconds() - tb0;
    }

    virtual void processAllImages()
    {
        std::vector<size_t> ids(DFexp.size());
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = i;
        processSomeImages(ids);
    }

    // The images go to the devices in contiguous ranges and the rows come back in the same order, so the output
    // does not depend on the number of devices: the reference visiting order alternates per image (APM:1112),
    // hence each range starts with the parity its first image would have had in a single sequential run.
    virtual void processSomeImages(const std::vector<size_t> &imagesToProcess)
    {
        const size_t G = slots.size(), count = imagesToProcess.size();
        std::vector<MetaDataVec> out(G);
        const bool forward0 = loop_forward_refs;
        const double tl0 = nowSeconds();
        runOnSlots(G, [&](size_t g) {
            const size_t lo = (g * count) / G, hi = ((g + 1) * count) / G;
            if (hi == lo) return;
            bindToDeviceNode(slots[g].device);          // this thread and the loader threads it starts: the device's side of the host
            std::vector<size_t> mine(imagesToProcess.begin() + lo, imagesToProcess.begin() + hi);
            processShard(slots[g], mine, (lo & 1) ? !forward0 : forward0, out[g]);
        });
        timing.loop += nowSeconds() - tl0;
        if (count & 1) loop_forward_refs = !loop_forward_refs;
        for (MetaDataVec &o : out) {
            if (o.rows.empty()) continue;
            if (DFo.labels.empty()) DFo.labels = o.labels;
            if (DFo.labels != o.labels) REPORT_ERROR(ERR_VALUE_INCORRECT, "internal: per-device result tables differ in labels");
            if (DFo.rows.empty()) DFo.rows.swap(o.rows);
            else for (auto &r : o.rows) DFo.rows.push_back(std::move(r));
        }
    }

    // what the host knows about one device batch before the device sees it: where the pixels lie, the previous shifts, the
    // neighbour lists in stack positions (CSR), assembled by a worker thread while the device is on the previous batch
    struct BatchInput {
        size_t b0 = 0, n = 0;
        std::vector<StackSource::Loc> locs;
        std::vector<float> prevShift;
        std::vector<int32_t> off, idsv;
        bool anyShift = false, wholeGallery = false;
    };
    // the matches of one batch as they come back from the device
    struct BatchResult {
        size_t b0 = 0, n = 0;
        std::vector<int32_t> refpos, psi;
        std::vector<uint8_t> flip;
        std::vector<double> f64;
        std::vector<float> prevShift;
    };

    void processShard(Slot &slot, const std::vector<size_t> &imagesToProcess, bool loop_forward_refs, MetaDataVec &DFo)
    {
        // APM:991-1192, batched and pipelined: reader threads + copy stream (fastio.h: BatchFeeder) fill batch k + 1, a worker
        // assembles the lists of batch k + 2 and another turns the matches of batch k - 1 into rows while the device is on
        // batch k. Row order and labels as APM:1149-1165.
        xh_ctx *ctx = slot.ctx;
        xh_pm *pm = slot.pm;
        xh_rf *&shifter = slot.shifter;
        const size_t per = dim * dim, total = imagesToProcess.size(), B = (size_t)batch, nb = (total + B - 1) / B;
        if (!total) return;
        DeviceBuffer d_shifted, d_i32a, d_i32b, d_u8, d_f64;
        HostTiming timing;                          // this device's share; added to the program's at the end
        BatchFeeder &feeder = *slot.feeder;
        feeder.timing = &timing;
        const int cImage = DFexp.col("image"), cSx = DFexp.col("shiftX"), cSy = DFexp.col("shiftY"), cScale = DFexp.col("scale"), cItem = DFexp.col("itemId");
        const size_t K = (size_t)numOrientations;
        DFo.labels = {"itemId", "image", "angleRot", "angleTilt", "anglePsi", "shiftX", "shiftY", "ref", "flip", "scale", "maxCC"};

        auto prepare = [&](size_t k) {
            BatchInput in;
            in.b0 = k * B;
            in.n = std::min(B, total - in.b0);
            in.locs.resize(in.n);
            in.prevShift.assign(2 * in.n, 0.f);
            in.off.assign(in.n + 1, 0);
            bool oneList = true;
            for (size_t i = 0; i < in.n; ++i) {
                const size_t id = imagesToProcess[in.b0 + i];
                in.locs[i] = feeder.source.locate(DFexp.cell(cImage, id), dim);
                in.prevShift[2 * i] = (float)DFexp.getDouble(cSx, id, 0.);
                in.prevShift[2 * i + 1] = (float)DFexp.getDouble(cSy, id, 0.);
                // getCurrentImage folds MDL_SCALE into the transformation it applies (APM:1222-1233); only shifts are
                // resampled here, so a scaled input row would silently give other matches than the reference
                if (std::fabs(DFexp.getDouble(cScale, id, 1.0) - 1.0) > 1e-9)
                    REPORT_ERROR(ERR_NOT_IMPLEMENTED, "input image " + std::to_string(id + 1) + " carries scale != 1: scaled inputs are not resampled by this build");
                in.anyShift = in.anyShift || in.prevShift[2 * i] != 0.f || in.prevShift[2 * i + 1] != 0.f;
                if (id >= mysampling.my_neighbors.size()) REPORT_ERROR(ERR_MD_NOOBJ, "No neighbour list for image " + std::to_string(id + 1));
                oneList = oneList && mysampling.my_neighbors.listOf[id] == mysampling.my_neighbors.listOf[imagesToProcess[in.b0]];
                in.off[i + 1] = in.off[i] + (int32_t)mysampling.my_neighbors.count(id);
            }
            // every image of the batch searching the whole gallery in stack order: the library's dense mode, no lists at all
            in.wholeGallery = oneList && listIsWholeGallery[mysampling.my_neighbors.listOf[imagesToProcess[in.b0]]];
            if (!in.wholeGallery) {
                in.idsv.resize((size_t)in.off[in.n]);
                for (size_t i = 0; i < in.n; ++i) {
                    const size_t id = imagesToProcess[in.b0 + i];
                    const int32_t *src = stackPositions.data() + (mysampling.my_neighbors.begin(id) - mysampling.my_neighbors.ids.data());
                    std::copy(src, src + mysampling.my_neighbors.count(id), in.idsv.begin() + in.off[i]);
                }
                if (in.idsv.empty()) in.idsv.push_back(0);
            }
            return in;
        };
        auto format = [&](BatchResult r) {
            const double t0 = nowSeconds();
            char b[64];
            auto real = [&b](double v) { snprintf(b, sizeof(b), "%.6f", v); return std::string(b); };
            for (size_t k = 0; k < r.n; ++k)
                for (size_t o = 0; o < K; ++o) {
                    const size_t e = k * K + o;
                    if (r.refpos[e] < 0) break;   // no (further) valid correlation: the reference writes no row (APM:1068-1090,1115)
                    const double *res = r.f64.data() + 3 * r.n * o;
                    const size_t id = imagesToProcess[r.b0 + k];
                    const std::vector<double> &ang = mysampling.no_redundant_sampling_points_angles[r.refpos[e]];
                    std::vector<std::string> row(11);
                    row[0] = std::to_string(DFexp.getLong(cItem, id, 0));
                    row[1].assign(DFexp.cell(cImage, id));
                    row[2] = real(ang[0]);
                    row[3] = real(ang[1]);
                    row[4] = real((double)r.psi[e] * (360. / N));
                    row[5] = real(res[k] + r.prevShift[2 * k]);
                    row[6] = real(res[r.n + k] + r.prevShift[2 * k + 1]);
                    row[7] = std::to_string((long)mysampling.no_redundant_sampling_points_index[r.refpos[e]]);
                    row[8] = std::to_string((long)r.flip[e]);
                    row[9] = real(DFexp.getDouble(cScale, id, 1.0));
                    row[10] = real(res[2 * r.n + k]);
                    DFo.rows.push_back(std::move(row));
                }
            timing.format += nowSeconds() - t0;
        };

        std::future<BatchInput> prepared = std::async(std::launch::async, prepare, (size_t)0);
        std::future<void> formatted;
        BatchInput cur = prepared.get();
        feeder.request(0, cur.locs, nullptr);
        if (nb > 1) prepared = std::async(std::launch::async, prepare, (size_t)1);
        for (size_t k = 0; k < nb; ++k) {
            const size_t n = cur.n;
            float *d_imgs = feeder.take(k);
            BatchInput next;
            if (k + 1 < nb) {
                next = prepared.get();
                feeder.request(k + 1, next.locs, ctx);     // (the copy stream waits for the work the device already has: batch k - 1)
                if (k + 2 < nb) prepared = std::async(std::launch::async, prepare, k + 2);
            }
            const double td0 = nowSeconds();
            if (cur.anyShift) {
                // getCurrentImage applies the previous shifts with BSPLINE3 + WRAP (APM:1228-1233)
                if (!shifter) {
                    xh_rf_params p{};
                    p.imgSize = (int)dim; p.padding_proj = p.padding_vol = 2; p.max_resolution = 0.5;
                    p.blob_radius = 1.9; p.blob_order = 0; p.blob_alpha = 15; p.min_ctf = 0.01; p.sampling = 1;
                    xhCheck(xh_rf_create(ctx, &p, &shifter));
                }
                d_shifted.reserve(ctx, B * per * sizeof(float));
                xhCheck(xh_rf_shift_images(shifter, d_imgs, cur.prevShift.data(), nullptr, (int)n, d_shifted.as<float>()));
                d_imgs = d_shifted.as<float>();
            }
            d_i32a.reserve(ctx, B * K * 4); d_i32b.reserve(ctx, B * K * 4); d_u8.reserve(ctx, B * K); d_f64.reserve(ctx, B * 8 * 3);
            // the visiting order of the references flips once per image (APM:1112)
            const int ntrans = search5d_xoff.size() > 1 ? (int)search5d_xoff.size() : 0;
            xhCheck(xh_pm_match_ex(pm, d_imgs, (int)n, cur.wholeGallery ? nullptr : cur.off.data(), cur.wholeGallery ? nullptr : cur.idsv.data(),
                                   loop_forward_refs ? 0 : 1, (int)K, ntrans, search5d_xoff.data(), search5d_yoff.data(), d_i32a.as<int32_t>(),
                                   d_i32b.as<int32_t>(), d_u8.as<uint8_t>()));
            if (n & 1) loop_forward_refs = !loop_forward_refs;
            BatchResult res;
            res.b0 = cur.b0; res.n = n;
            res.refpos.resize(n * K); res.psi.resize(n * K); res.flip.resize(n * K); res.f64.resize(3 * n * K);
            res.prevShift = std::move(cur.prevShift);
            // translational step per kept orientation (APM:1117-1124)
            double *d_sx = d_f64.as<double>(), *d_sy = d_sx + n, *d_cc = d_sy + n;
            if (K == 1) {
                xhCheck(xh_pm_translate(pm, d_imgs, (int)n, d_i32a.as<int32_t>(), d_i32b.as<int32_t>(), d_u8.as<uint8_t>(), max_shift, d_sx, d_sy, d_cc));
                xhCheck(xh_memcpy_d2h_async(ctx, res.refpos.data(), d_i32a.p, n * 4));
                xhCheck(xh_memcpy_d2h_async(ctx, res.psi.data(), d_i32b.p, n * 4));
                xhCheck(xh_memcpy_d2h_async(ctx, res.flip.data(), d_u8.p, n));
                xhCheck(xh_memcpy_d2h(ctx, res.f64.data(), d_f64.p, n * 8 * 3));
            } else {
                xhCheck(xh_memcpy_d2h(ctx, res.refpos.data(), d_i32a.p, n * K * 4));
                xhCheck(xh_memcpy_d2h(ctx, res.psi.data(), d_i32b.p, n * K * 4));
                xhCheck(xh_memcpy_d2h(ctx, res.flip.data(), d_u8.p, n * K));
                DeviceBuffer d_r1, d_p1, d_f1;
                d_r1.reserve(ctx, n * 4); d_p1.reserve(ctx, n * 4); d_f1.reserve(ctx, n);
                std::vector<int32_t> r1(n), p1(n);
                std::vector<uint8_t> f1(n);
                for (size_t o = 0; o < K; ++o) {
                    for (size_t i = 0; i < n; ++i) { r1[i] = res.refpos[i * K + o]; p1[i] = res.psi[i * K + o]; f1[i] = res.flip[i * K + o]; }
                    xhCheck(xh_memcpy_h2d(ctx, d_r1.p, r1.data(), n * 4));
                    xhCheck(xh_memcpy_h2d(ctx, d_p1.p, p1.data(), n * 4));
                    xhCheck(xh_memcpy_h2d(ctx, d_f1.p, f1.data(), n));
                    xhCheck(xh_pm_translate(pm, d_imgs, (int)n, d_r1.as<int32_t>(), d_p1.as<int32_t>(), d_f1.as<uint8_t>(), max_shift, d_sx, d_sy, d_cc));
                    xhCheck(xh_memcpy_d2h(ctx, res.f64.data() + 3 * n * o, d_f64.p, n * 8 * 3));
                }
            }
            timing.device += nowSeconds() - td0;
            timing.images += n;
            if (formatted.valid()) formatted.get();
            formatted = std::async(std::launch::async, format, std::move(res));
            cur = std::move(next);
        }
        if (formatted.valid()) formatted.get();
        feeder.timing = nullptr;
        std::lock_guard<std::mutex> lock(timingMutex);
        this->timing.add(timing);
    }

    virtual void writeOutputFiles() { DFo.write(fn_out, do_append); }

    void run() override
    {
        // XMIPP_HIP_TIMING=1: where the wall-clock seconds of the run went, on stderr (fastio.h: HostTiming)
        const bool show_timing = getenv("XMIPP_HIP_TIMING") != nullptr;
        const double t0 = nowSeconds();
        produceSideInfo();
        show();
        processAllImages();
        const double t2 = nowSeconds();
        writeOutputFiles();
        timing.write = nowSeconds() - t2;
        timing.total = nowSeconds() - t0;
        if (show_timing) timing.print("xmipp_angular_projection_matching");
        if (verbose) std::cout << "done!" << std::endl;
    }
};

// ============================================================================================
class ProgRecFourierAccel : public XmippProgram {
public:
    std::string fn_in, fn_out, fn_sym;
    bool do_weights = false, useFast = false, useCTF = false, isPhaseFlipped = false;
    double padding_factor_proj = 2, padding_factor_vol = 2, maxResolution = 0.5, minCTF = 0.01, Ts = 1;
    double blob_radius = 1.9, blob_alpha = 15;
    int blob_order = 0, bufferSize = 25, device = 0, gpus = 1, batch = 4096, readers = 0;
    std::string deviceList, fn_fsc;
    FastTable SF;
    HostTiming timing;
    std::mutex timingMutex;
    size_t imgSize = 0;
    std::vector<double> R_repository;   // nsym x 9
    // one slot per device: its own context (stream), gridding handle and temp spaces; slot 0 finishes
    struct Slot { int device = 0; xh_ctx *ctx = nullptr; xh_rf *rf = nullptr; xh_rf2 *rf2 = nullptr; std::unique_ptr<BatchFeeder> feeder; };
    std::vector<Slot> slots;
    // xmipp_reconstruct_fourier (ProgRecFourier, reconstruction/reconstruct_fourier.cpp): its own double-precision arithmetic on the
    // device (xh_rf2_*); NiterWeight = --iter (RF:44,96)
    bool rfArithmetic = false;
    int NiterWeight = 1;

    ~ProgRecFourierAccel() override
    {
        for (Slot &s : slots) {
            s.feeder.reset();
            if (s.rf2) xh_rf2_destroy(s.rf2);
            if (s.rf) xh_rf_destroy(s.rf);
            if (s.ctx) xh_ctx_destroy(s.ctx);
        }
    }
    void setIO(const std::string &in, const std::string &out) { fn_in = in; fn_out = out; }

    void defineParams() override
    {
        // RFA:55-82, verbatim parameter lines
        addUsageLine("Generate 3D reconstructions from projections using direct Fourier interpolation with arbitrary geometry.");
        addUsageLine("Kaisser-windows are used for interpolation in Fourier space.");
        addParamsLine("   -i <md_file>                : Metadata file with input projections");
        addParamsLine("  [-o <volume_file=\"rec_fourier.vol\">]  : Filename for output volume");
        addParamsLine("  [--sym <symfile=c1>]              : Enforce symmetry in projections");
        addParamsLine("                                    : (point group name or symmetry file; dNv / dNh with ODD N take their 2-fold on X,");
        addParamsLine("                                    :  where Sampling::createSymFile puts it on Y -- DESIGN history 8)");
        addParamsLine("  [--padding <proj=2.0> <vol=2.0>]  : Padding used for projections and volume");
        addParamsLine("  [--max_resolution <p=0.5>]     : Max resolution (Nyquist=0.5)");
        addParamsLine("  [--weight]                     : Use weights stored in the image metadata");
        addParamsLine("  [--blob <radius=1.9> <order=0> <alpha=15>] : Blob parameters");
        addParamsLine("                                 : radius in pixels, order of Bessel function in blob and parameter alpha");
        addParamsLine("  [--fast]                       : Do the blobing at the end of the computation.");
        addParamsLine("                                 : Gives slightly different results, but is faster.");
        addParamsLine("  [--useCTF]                     : Use CTF information if present");
        addParamsLine("  [--sampling <Ts=1>]            : sampling rate of the input images in Angstroms/pixel");
        addParamsLine("                                 : It is only used when correcting for the CTF");
        addParamsLine("  [--phaseFlipped]               : Give this flag if images have been already phase flipped");
        addParamsLine("  [--minCTF <ctf=0.01>]          : Minimum value of the CTF that will be inverted");
        addParamsLine("                                 : CTF values (in absolute value) below this one will not be corrected");
        addParamsLine("  [--bufferSize <size=25>]        : Number of projection loaded in memory (will be actually 2x as much.");
        addParamsLine("  [--thr <threads=1> <rows=1>]   : Accepted for compatibility with xmipp_reconstruct_fourier (RF:50); unused");
        addParamsLine("  [--iter <iterations=1>]        : xmipp_reconstruct_fourier: number of iterations for weight correction (RF:44)");
        addParamsLine("  [--prepare_fsc <fscfile>]      : Filename root for FSC files (RF:47): <root>_1_recons.vol, <root>_2_recons.vol");
        addParamsLine("  [--device <id=0>]              : first HIP device");
        addParamsLine("  [--gpus <n=1>]                 : number of consecutive HIP devices, one host thread each");
        addParamsLine("  [--devices <list=\"\">]         : explicit comma-separated device ids (overrides --device/--gpus)");
        addParamsLine("  [--batch <n=4096>]             : Projections per device batch");
        addParamsLine("  [--readers <n=0>]              : Host threads reading images into page-locked memory (0: half the cores, at most 16)");
        addExampleLine("   xmipp_reconstruct_fourier_accel  -i reconstruction.sel --sym c2 --weight");
    }

    void readParams() override
    {
        // RFA:85-103
        fn_in = getParam("-i");
        fn_out = getParam("-o");
        fn_sym = getParam("--sym");
        do_weights = checkParam("--weight");
        padding_factor_proj = getDoubleParam("--padding", 0);
        padding_factor_vol = getDoubleParam("--padding", 1);
        blob_radius = getDoubleParam("--blob", 0);
        blob_order = (int)getIntParam("--blob", 1);
        blob_alpha = getDoubleParam("--blob", 2);
        useFast = checkParam("--fast");
        maxResolution = getDoubleParam("--max_resolution");
        useCTF = checkParam("--useCTF");
        isPhaseFlipped = checkParam("--phaseFlipped");
        minCTF = getDoubleParam("--minCTF");
        Ts = getDoubleParam("--sampling");
        bufferSize = (int)getIntParam("--bufferSize");
        device = (int)getIntParam("--device");
        gpus = (int)getIntParam("--gpus");
        deviceList = getParam("--devices");
        batch = std::max(1, (int)getIntParam("--batch"));
        readers = std::max(0, (int)getIntParam("--readers"));
        if (checkParam("--prepare_fsc")) fn_fsc = getParam("--prepare_fsc");
        NiterWeight = (int)getIntParam("--iter");
        if (NiterWeight < 0) REPORT_ERROR(ERR_ARG_INCORRECT, "--iter must not be negative");
        if (rfArithmetic && useFast) REPORT_ERROR(ERR_ARG_INCORRECT, "--fast belongs to xmipp_reconstruct_fourier_accel");
        if (rfArithmetic && !fn_fsc.empty() && NiterWeight > 1)
            // correctWeight re-processes the images with processImages(0, last, !fn_fsc.empty(), true) (RF:1088): in the reference the replay
            // passes re-enter the FSC branch and finish, write and zero the half maps once more per weight iteration. Not reproduced.
            std::cerr << "xmipp_reconstruct_fourier: --prepare_fsc with --iter > 1: the half maps are finished once, before the weight iterations; "
                         "the reference's replay passes would finish and zero them again (reconstruct_fourier.cpp:1088), which this program does not do\n";
        // the accel program has no weight iterations (RFA:61-79): under its own name --iter is accepted and ignored
    }

    void show()
    {
        if (verbose <= 0) return;
        // RFA:106-137
        std::cout << " =====================================================================\n"
                  << " Direct 3D reconstruction method using Kaiser windows as interpolators\n"
                  << " =====================================================================\n"
                  << " Input selfile             : " << fn_in << "\n padding_factor_proj       : " << padding_factor_proj
                  << "\n padding_factor_vol        : " << padding_factor_vol << "\n Output volume             : " << fn_out << "\n";
        if (!fn_sym.empty()) std::cout << " Symmetry file for projections : " << fn_sym << "\n";
        if (!fn_fsc.empty()) std::cout << " File root for FSC files: " << fn_fsc << "\n";
        std::cout << (do_weights ? " Use weights stored in the image headers or doc file\n" : " Do NOT use weights\n");
        std::cout << "\n Interpolation Function\n   blrad                 : " << blob_radius << "\n   blord                 : " << blob_order
                  << "\n   blalpha               : " << blob_alpha << "\n max_resolution          : " << maxResolution
                  << "\n -----------------------------------------------------------------" << std::endl;
    }

    void produceSideinfo()
    {
        // RFA:175-257
        const double tp0 = nowSeconds();
        SF.read(fn_in, readers);
        SF.removeDisabled();
        timing.parse += nowSeconds() - tp0;
        if (SF.size() == 0) REPORT_ERROR(ERR_MD_NOOBJ, "No enabled images in " + fn_in);
        std::string fnImg;
        SF.getValue("image", fnImg, 0);
        ImageInfo info = readInfo(FileName(fnImg).path);
        if (info.x != info.y) REPORT_ERROR(ERR_MULTIDIM_SIZE, "This algorithm only works for squared images");
        imgSize = info.x;
        R_repository = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (!fn_sym.empty()) {
            SymList SL;
            SL.readSymmetryFile(fn_sym);
            for (int i = 0; i < SL.symsNo(); ++i) R_repository.insert(R_repository.end(), SL.R[i].begin(), SL.R[i].end());
        }
        xh_rf_params p{};
        p.imgSize = (int)imgSize; p.padding_proj = padding_factor_proj; p.padding_vol = padding_factor_vol;
        p.max_resolution = maxResolution; p.blob_radius = blob_radius; p.blob_order = blob_order; p.blob_alpha = blob_alpha;
        p.use_fast = useFast; p.phase_flipped = isPhaseFlipped; p.min_ctf = minCTF; p.sampling = Ts;
        const double tc0 = nowSeconds();
        for (int d : parseDevices(device, gpus, deviceList)) {
            slots.emplace_back();
            Slot &s = slots.back();
            s.device = d;
        }
        runOnSlots(slots.size(), [&](size_t g) {
            Slot &s = slots[g];
            bindToDeviceNode(s.device);
            xhCheck(xh_ctx_create_private(s.device, &s.ctx));
            // the loader (page-locking its pieces takes tens of ms) is set up beside the temp spaces
            s.feeder.reset(new BatchFeeder);
            auto feederReady = std::async(std::launch::async, [&] {
                s.feeder->create(s.device, imgSize, std::min((size_t)batch, SF.size()), std::max(1, hostThreads(readers) / (int)slots.size()), nullptr);
            });
            xhCheck(xh_rf_create(s.ctx, &p, &s.rf));       // (the double-precision program uses it for the shifts of readApplyGeo only)
            if (rfArithmetic) xhCheck(xh_rf2_create(s.ctx, &p, NiterWeight, &s.rf2));
            else xhCheck(xh_rf_reset(s.rf));
            feederReady.get();
        });
        timing.setup += nowSeconds() - tc0;
    }

    // Images first..last (inclusive) go to the devices in contiguous ranges (the reference's
    // job farm, mpi_reconstruct_fourier_accel.cpp:285-294, hands out blocks of mpi_job_size instead)
    void processImages(size_t first, size_t last)
    {
        if (last < first || last == (size_t)-1) return;
        const size_t count = last - first + 1, G = slots.size();
        const double tl0 = nowSeconds();
        runOnSlots(G, [&](size_t g) {
            const size_t lo = first + (g * count) / G, hi = first + ((g + 1) * count) / G;
            bindToDeviceNode(slots[g].device);          // this thread and the loader threads it starts: the device's side of the host
            if (hi > lo) processShard(slots[g], lo, hi - 1);
        });
        timing.loop += nowSeconds() - tl0;
    }

    // every device's partial sums -> slot 0 (mirrorAndCropTempSpaces on each, then one tree reduction)
    void gatherCropped()
    {
        std::vector<xh_rf *> h;
        for (Slot &s : slots) {
            xhCheck(xh_rf_mirror_and_crop(s.rf));
            h.push_back(s.rf);
        }
        if (h.size() > 1) xhCheck(xh_rf_reduce(h.data(), (int)h.size()));
    }

    void resetSpaces() { for (Slot &s : slots) xhCheck(xh_rf_reset(s.rf)); }

    // the host's view of one device batch, assembled by a worker thread while the device is on the previous one
    struct BatchInput {
        size_t n = 0;
        std::vector<StackSource::Loc> locs;
        std::vector<double> ang;
        std::vector<float> w, sh;
        std::vector<uint8_t> fl;
        std::vector<xh_ctf_params> ctfs;
        bool anyGeo = false;
    };

    void processShard(Slot &slot, size_t first, size_t last)
    {
        // loadImageThread/preloadBuffer + processBuffer (RFA:300-388,939-966), batched on the device: reader threads and a copy stream
        // bring batch k + 1 into HBM (fastio.h: BatchFeeder), a worker reads the metadata columns of batch k + 2, the device grids batch k;
        // nothing on this thread waits for the device inside the loop -- the copy stream waits, on the device, for the batch whose buffer it reuses
        xh_ctx *ctx = slot.ctx;
        xh_rf *rf = slot.rf;
        const size_t per = imgSize * imgSize, B = (size_t)batch, total = last - first + 1, nb = (total + B - 1) / B;
        const bool hasCTF = useCTF && (SF.containsLabel("ctfModel") || SF.containsLabel("ctfDefocusU"));
        if (hasCTF && !SF.containsLabel("ctfDefocusU"))
            REPORT_ERROR(ERR_NOT_IMPLEMENTED, "ctfModel files are not read by this build; put the CTF columns in the metadata");
        HostTiming timing;                          // this device's share; added to the program's at the end
        BatchFeeder &feeder = *slot.feeder;
        feeder.timing = &timing;
        DeviceBuffer d_shift;
        const int cImage = SF.col("image"), cRot = SF.col("angleRot"), cTilt = SF.col("angleTilt"), cPsi = SF.col("anglePsi"), cSx = SF.col("shiftX"),
                  cSy = SF.col("shiftY"), cFlip = SF.col("flip"), cWeight = SF.col("weight");
        const char *ctfLabels[16] = {"ctfSamplingRate", "ctfVoltage", "ctfDefocusU", "ctfDefocusV", "ctfDefocusAngle", "ctfSphericalAberration", "ctfChromaticAberration",
                                     "ctfEnergyLoss", "ctfLensStability", "ctfConvergenceCone", "ctfLongitudinalDisplacement", "ctfTransversalDisplacement", "ctfQ0",
                                     "ctfK", "ctfPhaseShift", "ctfVPPRadius"};
        int cCtf[16];
        for (int i = 0; i < 16; ++i) cCtf[i] = SF.col(ctfLabels[i]);

        auto prepare = [&](size_t k) {
            BatchInput in;
            const size_t b0 = first + k * B;
            const size_t n = in.n = std::min(B, last + 1 - b0);
            in.locs.resize(n);
            in.ang.resize(3 * n);
            in.w.assign(n, 1.f);
            in.sh.assign(2 * n, 0.f);
            in.fl.assign(n, 0);
            in.ctfs.resize(hasCTF ? n : 0);
            for (size_t i = 0; i < n; ++i) {
                const size_t id = b0 + i;
                in.locs[i] = feeder.source.locate(SF.cell(cImage, id), imgSize);
                in.ang[3 * i] = SF.getDouble(cRot, id, 0); in.ang[3 * i + 1] = SF.getDouble(cTilt, id, 0); in.ang[3 * i + 2] = SF.getDouble(cPsi, id, 0);
                in.sh[2 * i] = (float)SF.getDouble(cSx, id, 0); in.sh[2 * i + 1] = (float)SF.getDouble(cSy, id, 0);
                in.fl[i] = SF.getDouble(cFlip, id, 0) != 0 ? 1 : 0;
                in.anyGeo = in.anyGeo || in.sh[2 * i] != 0.f || in.sh[2 * i + 1] != 0.f || in.fl[i];
                if (do_weights) in.w[i] = (float)SF.getDouble(cWeight, id, 1.0);
                if (hasCTF) {
                    xh_ctf_params &c = in.ctfs[i];
                    xh_ctf_defaults(&c);     // data/ctf.cpp:365-388
                    c.Tm = SF.getDouble(cCtf[0], id, 1); c.kV = SF.getDouble(cCtf[1], id, 100);
                    c.DeltafU = SF.getDouble(cCtf[2], id, 0); c.DeltafV = SF.getDouble(cCtf[3], id, c.DeltafU);
                    c.azimuthal_angle = SF.getDouble(cCtf[4], id, 0); c.Cs = SF.getDouble(cCtf[5], id, 0);
                    c.Ca = SF.getDouble(cCtf[6], id, 0); c.espr = SF.getDouble(cCtf[7], id, 0);
                    c.ispr = SF.getDouble(cCtf[8], id, 0); c.alpha = SF.getDouble(cCtf[9], id, 0);
                    c.DeltaF = SF.getDouble(cCtf[10], id, 0); c.DeltaR = SF.getDouble(cCtf[11], id, 0);
                    c.Q0 = SF.getDouble(cCtf[12], id, 0); c.K = SF.getDouble(cCtf[13], id, 1);
                    c.phase_shift = SF.getDouble(cCtf[14], id, 0); c.VPP_radius = SF.getDouble(cCtf[15], id, 0);
                }
            }
            return in;
        };

        std::future<BatchInput> prepared = std::async(std::launch::async, prepare, (size_t)0);
        BatchInput cur = prepared.get();
        feeder.request(0, cur.locs, nullptr);
        if (nb > 1) prepared = std::async(std::launch::async, prepare, (size_t)1);
        for (size_t k = 0; k < nb; ++k) {
            const size_t n = cur.n;
            float *imgs = feeder.take(k);
            BatchInput next;
            if (k + 1 < nb) {
                next = prepared.get();
                feeder.request(k + 1, next.locs, ctx);     // (the copy stream waits for what the device already has: batch k - 1)
                if (k + 2 < nb) prepared = std::async(std::launch::async, prepare, k + 2);
            }
            const double td0 = nowSeconds();
            if (cur.anyGeo) {   // Projection::readApplyGeo with only_apply_shifts (RFA:311-323)
                d_shift.reserve(ctx, B * per * 4);
                xhCheck(xh_rf_shift_images(rf, imgs, cur.sh.data(), cur.fl.data(), (int)n, d_shift.as<float>()));
                imgs = d_shift.as<float>();
            }
            // processBufferGPU in one call (RFG:417-473): FFT, CTF factor and modulator evaluated while the gridding records
            // are packed (no CTF planes, same records bit for bit as the three separate steps), insertion
            if (rfArithmetic)
                xhCheck(xh_rf2_insert(slot.rf2, imgs, hasCTF ? cur.ctfs.data() : nullptr, cur.ang.data(), do_weights ? cur.w.data() : nullptr, (int)n,
                                      R_repository.data(), (int)(R_repository.size() / 9), 0));
            else
                xhCheck(xh_rf_insert_images(rf, imgs, hasCTF ? cur.ctfs.data() : nullptr, cur.ang.data(), do_weights ? cur.w.data() : nullptr, (int)n,
                                            R_repository.data(), (int)(R_repository.size() / 9)));
            timing.device += nowSeconds() - td0;
            timing.images += n;
            cur = std::move(next);
        }
        const double td1 = nowSeconds();
        xhCheck(xh_ctx_sync(ctx));
        timing.device += nowSeconds() - td1;
        feeder.timing = nullptr;
        std::lock_guard<std::mutex> lock(timingMutex);
        this->timing.add(timing);
    }

    void finishComputations(const std::string &out_name)
    {
        const double t0 = nowSeconds();
        std::vector<double> vol(imgSize * imgSize * imgSize);
        xhCheck(xh_rf_finish(slots[0].rf, vol.data()));
        const double t1 = nowSeconds();
        writeVolume(out_name, vol.data(), imgSize, imgSize, imgSize);
        timing.finish += t1 - t0;
        timing.write += nowSeconds() - t1;
    }

    // ---- xmipp_reconstruct_fourier (RF:124-180): every device's Fourier volume and weights summed into slot 0 through host memory
    void gatherDouble()
    {
        if (slots.size() < 2) return;
        const size_t nd = xh_rf2_state_doubles(slots[0].rf2);
        std::vector<double> host(nd);
        DeviceBuffer d0;
        d0.reserve(slots[0].ctx, nd * sizeof(double));
        for (size_t g = 1; g < slots.size(); ++g) {
            DeviceBuffer dg;
            dg.reserve(slots[g].ctx, nd * sizeof(double));
            xhCheck(xh_rf2_state_export(slots[g].rf2, dg.as<double>()));
            xhCheck(xh_memcpy_d2h(slots[g].ctx, host.data(), dg.p, nd * sizeof(double)));
            xhCheck(xh_memcpy_h2d(slots[0].ctx, d0.p, host.data(), nd * sizeof(double)));
            xhCheck(xh_rf2_state_import(slots[0].rf2, d0.as<double>(), 1));
            xhCheck(xh_ctx_sync(slots[0].ctx));
            xhCheck(xh_rf2_reset(slots[g].rf2));
        }
    }

    // correctWeight (RF:836-990,1056-1101): the re-processing passes replay orientations, weights and symmetry only
    void correctWeightDouble()
    {
        xh_rf2 *h = slots[0].rf2;
        xhCheck(xh_rf2_weights_step(h, 0));
        for (int it = 1; it < NiterWeight; ++it) {
            xhCheck(xh_rf2_weights_step(h, 1));
            for (size_t b0 = 0; b0 < SF.size(); b0 += (size_t)batch) {
                const size_t n = std::min((size_t)batch, SF.size() - b0);
                std::vector<double> ang(3 * n);
                std::vector<float> w(n, 1.f);
                for (size_t k = 0; k < n; ++k) {
                    const size_t id = b0 + k;
                    ang[3 * k] = SF.getDouble("angleRot", id, 0); ang[3 * k + 1] = SF.getDouble("angleTilt", id, 0); ang[3 * k + 2] = SF.getDouble("anglePsi", id, 0);
                    if (do_weights) w[k] = (float)SF.getDouble("weight", id, 1.0);
                }
                xhCheck(xh_rf2_insert(h, nullptr, nullptr, ang.data(), do_weights ? w.data() : nullptr, (int)n, R_repository.data(), (int)(R_repository.size() / 9), 1));
            }
            xhCheck(xh_rf2_weights_step(h, 2));
        }
        xhCheck(xh_rf2_weights_step(h, 3));
    }

    void finishDouble(const std::string &out_name)
    {
        std::vector<double> vol(imgSize * imgSize * imgSize);
        xhCheck(xh_rf2_finish(slots[0].rf2, vol.data()));
        writeVolume(out_name, vol.data(), imgSize, imgSize, imgSize);
    }

    void runDouble()
    {
        const size_t last = SF.size() - 1;
        xh_rf2 *h = slots[0].rf2;
        if (fn_fsc.empty()) {
            processImages(0, last);
            gatherDouble();
        } else {
            // RF:991-1045: the halves are finished as they stand -- finishComputations without correctWeight, i.e. PROCESS_WEIGHTS
            // multiplies by the raw weights; kept -- and their Fourier volumes and weights summed for the final volume
            const size_t FSCIndex = last / 2, nd = xh_rf2_state_doubles(h);
            DeviceBuffer half1;
            half1.reserve(slots[0].ctx, nd * sizeof(double));
            processImages(0, FSCIndex);
            gatherDouble();
            xhCheck(xh_rf2_state_export(h, half1.as<double>()));
            finishDouble(fn_fsc + "_1_recons.vol");
            xhCheck(xh_rf2_reset(h));
            if (FSCIndex + 1 <= last) processImages(FSCIndex + 1, last);
            gatherDouble();
            finishDouble(fn_fsc + "_2_recons.vol");
            xhCheck(xh_rf2_state_import(h, half1.as<double>(), 1));
        }
        correctWeightDouble();
        finishDouble(fn_out);
    }

    void run() override
    {
        // RFA:139-156.  XMIPP_HIP_TIMING=1: where the wall-clock seconds of the run went, on stderr (fastio.h: HostTiming)
        struct Report {
            HostTiming &t; double t0 = nowSeconds();
            ~Report() { t.total = nowSeconds() - t0; if (getenv("XMIPP_HIP_TIMING")) t.print("xmipp_reconstruct_fourier_accel"); }
        } report{timing};
        show();
        produceSideinfo();
        if (rfArithmetic) { runDouble(); return; }
        const size_t last = SF.size() - 1;
        if (fn_fsc.empty()) {
            processImages(0, last);
            const double tg0 = nowSeconds();
            gatherCropped();
            timing.finish += nowSeconds() - tg0;
            finishComputations(fn_out);
            return;
        }
        // --prepare_fsc (RF:846,991-1045): images 0..FSCIndex make <root>_1_recons.vol, the rest
        // <root>_2_recons.vol, each from zeroed spaces; the final volume is the sum of both halves' Fourier
        // volumes and weights. The reference parks the halves in <root>_{1,2}_{Fourier,Weights}.vol and deletes
        // them afterwards; here they stay in device memory.
        const size_t FSCIndex = last / 2;
        xh_ctx *ctx0 = slots[0].ctx;
        xh_rf *rf0 = slots[0].rf;
        const size_t bytes = sizeof(float) * xh_rf_cropped_floats(rf0);
        DeviceBuffer half1, half2;
        half1.reserve(ctx0, bytes);
        half2.reserve(ctx0, bytes);
        processImages(0, FSCIndex);
        gatherCropped();
        xhCheck(xh_rf_cropped_export(rf0, half1.as<float>()));
        finishComputations(fn_fsc + "_1_recons.vol");
        resetSpaces();
        if (FSCIndex + 1 <= last) processImages(FSCIndex + 1, last);
        gatherCropped();
        xhCheck(xh_rf_cropped_export(rf0, half2.as<float>()));
        finishComputations(fn_fsc + "_2_recons.vol");
        xhCheck(xh_rf_cropped_import(rf0, half1.as<float>(), 0));
        xhCheck(xh_rf_cropped_import(rf0, half2.as<float>(), 1));
        finishComputations(fn_out);
    }
};

// ============================================================================================
// xmipp_angular_project_library (reconstruction/angular_project_library.cpp): the gallery of reference
// projections + its sampling / neighbourhood files. Sampling on the host (sampling_gen.h), projections on
// the device (xh_fp_*, the FourierProjector). --method fourier <pad> <maxfreq> bspline only.
class ProgAngularProjectLibrary : public XmippProgram {
public:
    std::string input_volume, output_file, output_file_root, fn_sym, fn_sym_neigh, FnexperimentalImages, fn_groups;
    double sampling = 5, psi_sampling = 360, max_tilt_angle = 180, min_tilt_angle = 0, angular_distance = 0;
    double paddFactor = 1, maxFrequency = 0.25, perturb_projection_vector = 0;
    bool angular_distance_bool = false, compute_closer_sampling_point_bool = false, compute_neighbors_bool = false;
    bool remove_points_far_away_from_experimental_data_bool = false, only_winner = false, only_sampling = false;
    int device = 0, batch = 256;
    SamplingGen mysampling;

    void defineParams() override
    {
        // angular_project_library.cpp:98-148
        addUsageLine("Create a gallery of projections from a volume");
        addParamsLine("   -i <input_volume_file>       : Input Volume");
        addParamsLine("   -o <output_file_name>        : stack with output files");
        addParamsLine("  [--sym <symmetry=c1>]         : Symmetry to define sampling ");
        addParamsLine("  [--sampling_rate <Ts=5>]      : Distance in degrees between sampling points");
        addParamsLine("==+Extra parameters==");
        addParamsLine("  [--sym_neigh <symmetry>]      : symmetry used to define neighbors, by default same as sym");
        addParamsLine("  [--psi_sampling <psi=360>]    : sampling in psi, 360 -> no sampling in psi");
        addParamsLine("  [--max_tilt_angle <tmax=180>] : maximum tilt angle in degrees");
        addParamsLine("  [--min_tilt_angle <tmin=0>]   : minimum tilt angle in degrees");
        addParamsLine("  [--experimental_images <docfile=\"\">] : doc file with experimental data");
        addParamsLine("  [--angular_distance <ang=20>]     : Do not search a distance larger than...");
        addParamsLine("  [--closer_sampling_points]    : create doc file with closest sampling points");
        addParamsLine("  [--near_exp_data]             : remove points far away from experimental data");
        addParamsLine("  [--compute_neighbors]         : create doc file with sampling point neighbors");
        addParamsLine("  [--method <method=fourier> <pad=1> <maxfreq=0.25> <interp=bspline>] : Projection method");
        addParamsLine("  [--perturb <sigma=0.0>]       : gaussian noise projection unit vectors ");
        addParamsLine("  [--groups <selfile=\"\">]     : selfile with groups");
        addParamsLine("  [--only_winner]               : if set each experimental point will have a unique neighbor");
        addParamsLine("  [--device <id=0>]             : HIP device");
        addParamsLine("  [--batch <n=256>]             : projections per device batch");
        addParamsLine("  [--only_create_sampling]      : write the sampling / angle files and stop (needs no device)");
        addExampleLine("xmipp_angular_project_library -i in.vol -o out.stk --sym c6 --sampling_rate 2");
    }
    void readParams() override
    {
        // angular_project_library.cpp:47-95
        input_volume = getParam("-i");
        output_file = getParam("-o");
        output_file_root = FileName(output_file).removeAllExtensions();
        fn_sym = getParam("--sym");
        fn_sym_neigh = checkParam("--sym_neigh") ? getParam("--sym_neigh") : fn_sym;
        sampling = getDoubleParam("--sampling_rate");
        psi_sampling = getDoubleParam("--psi_sampling");
        max_tilt_angle = getDoubleParam("--max_tilt_angle");
        min_tilt_angle = getDoubleParam("--min_tilt_angle");
        angular_distance_bool = checkParam("--angular_distance");
        if (angular_distance_bool) {
            FnexperimentalImages = getParam("--experimental_images");
            angular_distance = getDoubleParam("--angular_distance");
        }
        compute_closer_sampling_point_bool = checkParam("--closer_sampling_points");
        if (compute_closer_sampling_point_bool) FnexperimentalImages = getParam("--experimental_images");
        const std::string method = getParam("--method");
        if (method != "fourier")
            REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--method " + method + ": only the Fourier (central slice) projector is on the device");
        paddFactor = getDoubleParam("--method", 1);
        maxFrequency = getDoubleParam("--method", 2);
        if (getParam("--method", 3) != "bspline")
            REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--method fourier: only the 'bspline' interpolation kernel is on the device");
        perturb_projection_vector = getDoubleParam("--perturb");
        compute_neighbors_bool = checkParam("--compute_neighbors");
        remove_points_far_away_from_experimental_data_bool = checkParam("--near_exp_data");
        if (remove_points_far_away_from_experimental_data_bool) FnexperimentalImages = getParam("--experimental_images");
        fn_groups = getParam("--groups");
        only_winner = checkParam("--only_winner");
        device = (int)getIntParam("--device");
        batch = std::max(1, (int)getIntParam("--batch"));
        only_sampling = checkParam("--only_create_sampling");
    }
    // the point-group names of xmippCore's SymList::isSymmetryGroup, as the families of SamplingGen::removeRedundantPoints
    static void parseGroup(const std::string &sym, std::string &family, int &order)
    {
        std::string s = sym;
        std::transform(s.begin(), s.end(), s.begin(), ::tolower);
        order = 1;
        if (s.size() >= 2 && (s[0] == 'c' || s[0] == 'd' || s[0] == 's') && ::isdigit((unsigned char)s[1])) {
            size_t end = 1;
            while (end < s.size() && ::isdigit((unsigned char)s[end])) ++end;
            order = atoi(s.substr(1, end - 1).c_str());
            const std::string tail = s.substr(end);
            const bool ok = order >= 1 && (tail.empty() || ((tail == "v" || tail == "h") && s[0] != 's')) && (s[0] != 's' || order % 2 == 0);
            if (ok) { family = std::string(1, s[0]) + tail; return; }
        }
        for (const char *g : {"ci", "cs", "t", "td", "th", "o", "oh", "i1", "i2", "i3", "i4", "i1h", "i2h", "i3h", "i4h"})
            if (s == g) { family = s; return; }
        if (s == "i") { family = "i2"; return; }
        if (s == "ih") { family = "i2h"; return; }
        REPORT_ERROR(ERR_NOT_IMPLEMENTED, "symmetry '" + sym + "': the sampling of the asymmetric unit is implemented for cN, cNv, cNh, ci, cs, sN, dN, dNv, "
                     "dNh, t, td, th, o, oh, i1..i4, ih and i1h..i4h (i5 and i5h are not implemented in the reference either, sampling.cpp:1072, 1216)");
    }
    void run() override
    {
        // angular_project_library.cpp:249-397
        if (perturb_projection_vector != 0) REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--perturb is not available (it is seeded with time() in the reference)");
        if (compute_closer_sampling_point_bool && FnexperimentalImages.empty()) REPORT_ERROR(ERR_ARG_MISSING, "--closer_sampling_points needs --experimental_images");
        if (!fn_groups.empty() && FnexperimentalImages.empty()) REPORT_ERROR(ERR_ARG_MISSING, "--groups needs --experimental_images");
        std::string fam, famN;
        int order, orderN;
        parseGroup(fn_sym, fam, order);
        parseGroup(fn_sym_neigh, famN, orderN);
        mysampling.setSampling(sampling);
        if (angular_distance_bool) mysampling.setNeighborhoodRadius(angular_distance);
        mysampling.computeSamplingPoints(false, max_tilt_angle, min_tilt_angle);
        mysampling.removeRedundantPoints(fam, order);
        SymList SLn;
        SLn.readSymmetryFile(fn_sym_neigh);
        mysampling.fillLRRepository(SLn);
        if (!FnexperimentalImages.empty()) {
            MetaDataVec DFi;
            DFi.read(FnexperimentalImages);
            mysampling.fillExpDataProjectionDirectionByLR(DFi);
            if (remove_points_far_away_from_experimental_data_bool) mysampling.removePointsFarAwayFromExperimentalData();
            if (compute_closer_sampling_point_bool) mysampling.findClosestSamplingPoint(DFi, output_file_root);
        }
        mysampling.createAsymUnitFile(output_file_root);
        if (compute_neighbors_bool) {
            mysampling.computeNeighbors(only_winner);
            mysampling.saveSamplingFile(output_file_root, false);
        }
        const size_t nDir = mysampling.no_redundant_sampling_points_angles.size();
        int numberStepsPsi = 1;
        if (psi_sampling < 360) numberStepsPsi = (int)(359.99999 / psi_sampling);
        // the final docfile (angular_project_library.cpp:357-393)
        {
            MetaDataVec out;
            size_t counter = 0;
            for (double mypsi = 0; mypsi < 360; mypsi += psi_sampling)
                for (size_t i = 0; i < nDir; ++i) {
                    const size_t id = out.addObject();
                    char name[32];
                    snprintf(name, sizeof(name), "%06zu@", ++counter);
                    out.setValue("image", std::string(name) + output_file, id);
                    out.setValue("enabled", (long)1, id);
                    out.setValue("angleRot", mysampling.no_redundant_sampling_points_angles[i][0], id);
                    out.setValue("angleTilt", mysampling.no_redundant_sampling_points_angles[i][1], id);
                    out.setValue("anglePsi", mysampling.no_redundant_sampling_points_angles[i][2] + mypsi, id);
                    out.setValue("X", mysampling.no_redundant_sampling_points_vector[i][0], id);
                    out.setValue("Y", mysampling.no_redundant_sampling_points_vector[i][1], id);
                    out.setValue("Z", mysampling.no_redundant_sampling_points_vector[i][2], id);
                    out.setValue("scale", 1.0, id);
                    out.setValue("ref", (long)i, id);
                }
            out.comment = "x,y,z refer to the coordinates of the unitary vector at direction given by the euler angles";
            if (out.size() > 0) out.write(output_file_root + ".doc");
            else std::cout << "There are no projections within the specified angular range and sampling" << std::endl;
        }
        std::remove((output_file_root + "_angles.doc").c_str());
        if (!fn_groups.empty()) {
            // createGroupSamplingFiles (angular_project_library.cpp:405-470): a sampling file per block of the groups file, with the
            // neighbours of that block's experimental images (the block of the same name in the experimental metadata)
            const std::string fn_exp = FileName(FnexperimentalImages).path;
            int igrp = 1;
            for (const std::string &block : getBlocksInMetaDataFile(fn_groups)) {
                char num[16];
                snprintf(num, sizeof(num), "%06d", igrp++);
                const std::string root = output_file_root + "_group" + num;
                std::cerr << "Writing group sampling file " << root << std::endl;
                MetaDataVec SFBlock;
                SFBlock.read(block + "@" + fn_exp);
                if (SFBlock.size() > 0) {
                    mysampling.fillExpDataProjectionDirectionByLR(SFBlock);
                    if (compute_closer_sampling_point_bool) mysampling.findClosestSamplingPoint(SFBlock, root);
                    if (compute_neighbors_bool) {
                        mysampling.computeNeighbors(only_winner);
                        mysampling.saveSamplingFile(root, false);
                    }
                }
            }
        }
        if (only_sampling) {
            // diagnostic: the rotations of the neighbourhood group (identity first), one 3x3 matrix per line
            std::ofstream f(output_file_root + "_symmetry.txt");
            f.precision(17);
            for (const auto &Rm : mysampling.R_repository) { for (double v : Rm) f << v << ' '; f << '\n'; }
        }
        if (only_sampling || nDir == 0) return;
        // projections (project_angle_vector, angular_project_library.cpp:194-245): psi-major, direction-minor
        std::vector<float> vol;
        ImageInfo vi;
        readImage(input_volume, vol, vi);
        if (vi.x != vi.y || vi.x != vi.z) REPORT_ERROR(ERR_MULTIDIM_SIZE, "the input volume must be cubic");
        const size_t D = vi.x;
        xh_ctx *ctx = nullptr;
        xhCheck(xh_ctx_create_private(device, &ctx));
        struct Guard { xh_ctx *c; xh_fp *f = nullptr; ~Guard() { if (f) xh_fp_destroy(f); if (c) xh_ctx_destroy(c); } } guard{ctx};
        DeviceBuffer d_vol, d_out;
        d_vol.reserve(ctx, vol.size() * sizeof(float));
        xhCheck(xh_memcpy_h2d(ctx, d_vol.p, vol.data(), vol.size() * sizeof(float)));
        xhCheck(xh_fp_create(ctx, d_vol.as<float>(), (int)D, paddFactor, maxFrequency, 3, &guard.f));
        d_vol.release();
        const size_t total = nDir * (size_t)numberStepsPsi;
        std::vector<float> gallery(total * D * D);
        std::vector<double> ang;
        d_out.reserve(ctx, (size_t)batch * D * D * sizeof(float));
        for (size_t n0 = 0; n0 < total; n0 += (size_t)batch) {
            const size_t m = std::min((size_t)batch, total - n0);
            ang.resize(3 * m);
            for (size_t k = 0; k < m; ++k) {
                const size_t n = n0 + k, psiIndex = n / nDir, index = n % nDir;
                ang[3 * k] = mysampling.no_redundant_sampling_points_angles[index][0];
                ang[3 * k + 1] = mysampling.no_redundant_sampling_points_angles[index][1];
                ang[3 * k + 2] = psiIndex * psi_sampling + mysampling.no_redundant_sampling_points_angles[index][2];
            }
            xhCheck(xh_fp_project(guard.f, ang.data(), (int)m, nullptr, d_out.as<float>()));
            xhCheck(xh_memcpy_d2h(ctx, gallery.data() + n0 * D * D, d_out.p, m * D * D * sizeof(float)));
        }
        d_out.release();
        writeStack(output_file, gallery.data(), D, D, total);
    }
};

// ============================================================================================
// xmipp_resolution_fsc (reconstruction/resolution_fsc.{h,cpp}): Fourier shell correlation of a map against a
// reference map, SURVEY.md 8f rank 2. The shell sums run on the device (xh_frc_dpr); --set_of_images
// (getFourierStatistics, xmippCore) is not available.
class ProgResolutionFsc : public XmippProgram {
public:
    std::string fn_ref, fn_img, fn_sel, fn_root, fn_out;
    double sam = 1, max_sam = -1, min_sam = -1;
    bool do_dpr = false, do_set_of_images = false, do_o = false, do_rfactor = false, apply_geo = true;
    int device = 0;

    void defineParams() override
    {
        // resolution_fsc.cpp:34-82, verbatim parameter lines
        addUsageLine("Calculate the resolution of one or more volumes or images with respect to a single reference.");
        addUsageLine("+ Three methods are employed:");
        addUsageLine("+ * Differential Phase Residual (DPR), Fourier Ring Correlation (FRC), Spectral Signal-to-Noise Ratio (SSNR)");
        addParamsLine("   -i <input_file>           : either an image or a volume");
        addParamsLine("   requires --ref;");
        addParamsLine("or --set_of_images <selfile> : selfile containing a set of 2D-images");
        addParamsLine("   [--oroot <root_file=\"\">] : Root of the output metadata. If not set, input file rootname is taken.");
        addParamsLine("   [-o <output_file=\"\">]   : Output file name.");
        addParamsLine("   [--ref <input_file>]      : filename for reference image/volume");
        addParamsLine("   [--sampling_rate <Ts=1>]  : Pixel size (Angstrom)");
        addParamsLine("  alias -s;");
        addParamsLine("   [--dont_apply_geo]        : for 2D-images: do not apply transformation stored in the header");
        addParamsLine("   [--do_dpr]                : compute dpr, by default only frc is computed");
        addParamsLine("   [--max_sam <max_sr=-1>]   : set fsc to 0 for frequencies above this one (Angstrom), -1 -> all fequencies");
        addParamsLine("                             : --max_sam = 10A -> frequencies higher than 0.1 A^-1 =0");
        addParamsLine("   [--do_rfactor]            : compute R-factor for input volumes");
        addParamsLine("   [--min_sam <min_sr=-1>]   : minimum frequency may use for calculating R-factor (Angstrom)");
        addParamsLine("                             : --min_sam = 10A -> frequencies smaller than 0.1 A^-1 =0");
        addParamsLine("   [--device <id=0>]         : HIP device");
        addExampleLine("xmipp_resolution_fsc --ref subset1.vol  -i subset2.vol --sampling_rate 5.6 ");
    }

    void readParams() override
    {
        // resolution_fsc.cpp:84-118
        sam = getDoubleParam("--sampling_rate");
        apply_geo = !checkParam("--dont_apply_geo");
        max_sam = getDoubleParam("--max_sam");
        do_dpr = checkParam("--do_dpr");
        do_set_of_images = checkParam("--set_of_images");
        min_sam = getDoubleParam("--min_sam");
        do_rfactor = checkParam("--do_rfactor");
        if (do_set_of_images) {
            fn_sel = getParam("--set_of_images");
            if (checkParam("-i") || checkParam("--ref")) REPORT_ERROR(ERR_ARG_INCORRECT, "--set_of_images should not be provided with -i or --ref");
        } else {
            if (!checkParam("-i")) REPORT_ERROR(ERR_ARG_MISSING, "-i is mandatory");
            if (!checkParam("--ref")) REPORT_ERROR(ERR_ARG_MISSING, "-i requires --ref");
            fn_ref = getParam("--ref");
            fn_img = getParam("-i");
        }
        do_o = checkParam("-o");
        if (do_o) fn_out = getParam("-o");
        else fn_root = getParam("--oroot");
        device = (int)getIntParam("--device");
    }

    // resolution_fsc.cpp:120-170
    void writeFiles(const std::string &fnRoot, const std::vector<double> &freq, std::vector<double> &frc, const std::vector<double> &frc_noise,
                    std::vector<double> &dpr, const std::vector<double> &error_l2, double rFactor)
    {
        MetaDataVec MD;
        const std::string fn_frc = do_o ? fn_out : fnRoot + ".frc";
        for (size_t i = 1; i < freq.size(); ++i) {
            const size_t id = MD.addObject();
            if (max_sam > 0 && (1. / freq[i]) < max_sam) { if (do_dpr) dpr[i] = 0.; frc[i] = 0.; }
            if (min_sam > 0 && (1. / freq[i]) > min_sam) { if (do_dpr) dpr[i] = 0.; frc[i] = 0.; }
            MD.setValue("resolutionFreqFourier", freq[i], id);
            MD.setValue("resolutionFRC", frc[i], id);
            if (do_dpr) MD.setValue("resolutionDPR", dpr[i], id);
            MD.setValue("resolutionErrorL2", error_l2[i], id);
            MD.setValue("resolutionFRCRandomNoise", frc_noise[i], id);
            MD.setValue("resolutionFreqReal", 1. / freq[i], id);
        }
        MD.write(fn_frc);
        // second block, row format (MD2.setColumnFormat(false); MD_APPEND)
        std::ofstream f(FileName(fn_frc).path, std::ios::app);
        char b[64];
        snprintf(b, sizeof(b), "%.6f", rFactor);
        f << "data_rfactor\n _resolutionRfactor " << b << "\n";
    }

    void run() override
    {
        if (do_set_of_images) REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--set_of_images (getFourierStatistics, resolution_fsc.cpp:205-214) is not available on the device path");
        // process_img, resolution_fsc.cpp:172-203
        std::vector<float> r32, i32;
        ImageInfo ri, ii;
        readImage(fn_ref, r32, ri);
        readImage(fn_img, i32, ii);
        if (ri.x != ii.x || ri.y != ii.y || ri.z != ii.z) REPORT_ERROR(ERR_MULTIDIM_SIZE, "MultidimArrays have different shapes!");
        std::vector<double> m1(r32.begin(), r32.end()), m2(i32.begin(), i32.end());
        double min_samp = sam / min_sam;
        if (min_sam < 0) min_samp = 0.;
        if (max_sam < 0) max_sam = 2 * sam;
        xh_ctx *ctx = nullptr;
        xhCheck(xh_ctx_create_private(device, &ctx));
        struct Guard { xh_ctx *c; ~Guard() { if (c) xh_ctx_destroy(c); } } guard{ctx};
        const size_t L = ri.x / 2 + 1;
        std::vector<double> freq(L), frc(L), frc_noise(L), dpr(L, 0.), error_l2(L);
        double rFactor = -1.;
        {
            DeviceBuffer d1, d2;
            d1.reserve(ctx, m1.size() * sizeof(double));
            d2.reserve(ctx, m2.size() * sizeof(double));
            xhCheck(xh_memcpy_h2d(ctx, d1.p, m1.data(), m1.size() * sizeof(double)));
            xhCheck(xh_memcpy_h2d(ctx, d2.p, m2.data(), m2.size() * sizeof(double)));
            xhCheck(xh_frc_dpr(ctx, d1.as<double>(), d2.as<double>(), (int)ri.z, (int)ri.y, (int)ri.x, sam, do_dpr, do_rfactor, min_samp,
                               sam / max_sam, freq.data(), frc.data(), frc_noise.data(), dpr.data(), error_l2.data(), &rFactor));
        }
        writeFiles(fn_root.empty() ? fn_img : fn_root, freq, frc, frc_noise, dpr, error_l2, rFactor);
    }
};

}  // namespace mc
#endif
