// movie_programs.h -- host side of FlexAlign: the command line and the run() of AProgMovieAlignmentCorrelation
// (reconstruction/movie_alignment_correlation_base.cpp:36-150,519-586) over xh_fa_*, with the steps of the CUDA program
// (reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp): global alignment, local (patch) alignment unless
// --skipLocalAlignment, the B-spline warp of every summed frame (localFromGlobal when only the global alignment ran), the sums
// and the metadata blocks referenceFrame / localAlignment / frameShifts; --bin bins every frame while it is loaded (Fourier cropping,
// xh_movie_bin_frame).  Not here: the reference's search for FFT-friendly patch and correlation sizes (it benchmarks cuFFT on the
// installed GPU; the requested sizes are used).
#ifndef XMIPP3_AMD_MOVIE_PROGRAMS_H
#define XMIPP3_AMD_MOVIE_PROGRAMS_H
#include "ctf_programs.h"
#include <condition_variable>
#include <mutex>
#include <thread>
#include <deque>
#include <future>
#include <memory>

namespace mc {

class ProgMovieAlignmentCorrelation : public XmippProgram {
public:
    std::string fnMovie, fnOut, fnInitialAvg, fnDark, fnGain, fnAligned, fnAvg;
    float binning = 1, Ts = 1, maxShift = 50, maxResForCorrelation = 30;
    int nfirst = -1, nlast = -1, nfirstSum = -1, nlastSum = -1, device = 0;
    int cpX = 6, cpY = 6, cpT = 5, patchesX = 0, patchesY = 0, patchesAvg = 3, minLocalRes = 500;
    bool skipLocalAlignment = false;
    std::string sizeSearch = "off";

    // BSplineHelper::getShift (bspline_helper.cpp:104-148)
    void splineShift(const std::vector<double> &cX, const std::vector<double> &cY, int X, int Y, int N, int x, int y, int n, double &sx, double &sy) const
    {
        auto b3 = [](double v) { v = std::fabs(v); if (v < 1) return (v * v * (v - 2) * 3 + 4) / 6; if (v < 2) { v -= 2; return v * v * v / -6; } return 0.0; };
        const double hX = cpX == 3 ? X : X / (double)(cpX - 3), hY = cpY == 3 ? Y : Y / (double)(cpY - 3), hT = cpT == 3 ? N : N / (double)(cpT - 3);
        const double xPos = x / hX, yPos = y / hY, tPos = n / hT;
        sx = sy = 0;
        for (int it = std::max(-1, (int)tPos - 1); it <= std::min((int)tPos + 2, cpT - 2); ++it)
            for (int iy = std::max(-1, (int)yPos - 1); iy <= std::min((int)yPos + 2, cpY - 2); ++iy)
                for (int ix = std::max(-1, (int)xPos - 1); ix <= std::min((int)xPos + 2, cpX - 2); ++ix) {
                    const double tmp = b3(xPos - ix) * b3(yPos - iy) * b3(tPos - it);
                    if (std::fabs((float)tmp) > 0.0001) {
                        const size_t o = (size_t)(it + 1) * (cpX * cpY) + (size_t)(iy + 1) * cpX + (ix + 1);
                        sx += cX[o] * tmp; sy += cY[o] * tmp;
                    }
                }
    }

    // --sizeSearch smooth.  What CudaFFT::findOptimal / cuFFTAdvisor (an external dependency, not in the reference tree) look for: sizes near
    // the requested one whose prime factors are 2, 3, 5 and 7 -- below it when cropping, above it otherwise, within sigPercChange per cent
    // (10 for the crop, 20 for patches; movie_alignment_correlation_gpu.cpp:80,95,112) -- of which they time the candidates on the card and
    // keep the fastest.  The stand-in keeps the candidate nearest to the request (the largest below / the smallest above): no timing, the
    // same answer on every machine.  Even sizes only (the correlation code halves them).
    static size_t smoothSize(size_t n, int percent, bool crop)
    {
        auto smooth = [](size_t v) { for (size_t p : {2, 3, 5, 7}) while (v % p == 0) v /= p; return v == 1; };
        const size_t lo = crop ? (size_t)std::ceil(n * (100 - percent) / 100.0) : n, hi = crop ? n : (size_t)std::floor(n * (100 + percent) / 100.0);
        if (crop) { for (size_t v = hi & ~(size_t)1; v >= std::max<size_t>(lo, 2); v -= 2) if (smooth(v)) return v; }
        else for (size_t v = (lo + 1) & ~(size_t)1; v <= hi; v += 2) if (smooth(v)) return v;
        return n & ~(size_t)1;
    }

    void defineParams() override
    {
        // movie_alignment_correlation_base.cpp:111-150, verbatim parameter lines
        addUsageLine("Align a set of frames by cross-correlation of the frames");
        addParamsLine("   -i <metadata>               : Metadata with the list of frames to align");
        addParamsLine("  [-o <fn=\"out.xmd\">]        : Metadata with the shifts of each frame.");
        addParamsLine("                               : If no filename is given, the input is rewritten");
        addParamsLine("  [--bin <s=1>]                : Binning factor, it may be any floating number > 1.");
        addParamsLine("                               : Binning is applied during the data loading, i.e. the program will processed and store binned data.");
        addParamsLine("  [--maxShift <s=50>]          : Maximum shift allowed in A");
        addParamsLine("  [--maxResForCorrelation <R=30>]: Maximum resolution to align (in Angstroms)");
        addParamsLine("  [--sampling <Ts=1>]          : Sampling rate (A/pixel)");
        addParamsLine("  [--oaligned <fn=\"\">]       : Aligned movie consists of aligned frames used for micrograph generation");
        addParamsLine("  [--oavgInitial <fn=\"\">]    : Give the name of a micrograph to generate an unaligned (initial) micrograph");
        addParamsLine("  [--oavg <fn=\"\">]           : Give the name of a micrograph to generate an aligned micrograph");
        addParamsLine("  [--frameRange <n0=-1> <nF=-1>]  : First and last frame to align, frame numbers start at 0");
        addParamsLine("  [--frameRangeSum <n0=-1> <nF=-1>]  : First and last frame to sum, frame numbers start at 0");
        addParamsLine("  [--dark <fn=\"\">]           : Dark correction image");
        addParamsLine("  [--gain <fn=\"\">]           : Gain correction image (we will multiply by it)");
        addParamsLine("  [--skipLocalAlignment]       : If used, only global alignment will be performed. It's faster, but gives worse results.");
        addParamsLine("  [--controlPoints <x=6> <y=6> <t=5>]: Number of control points (including end points) used for defining the BSpline");
        addParamsLine("  [--patches <x=7> <y=7>]: Number of patches used for local alignment");
        addParamsLine("  [--minLocalRes <R=500>]      : Minimal resolution (in A) of patches during local alignment");
        addParamsLine("  [--patchesAvg <avg=3>]       : Number of near frames used for averaging a single patch");
        addParamsLine("  [--device <id=0>]            : HIP device");
        addParamsLine("  [--sizeSearch <mode=off>]    : off: frames, patches and correlations at the sizes the parameters ask for (results depend on the");
        addParamsLine("                               : inputs only).  smooth: a deterministic stand-in for the CUDA program's search of FFT-friendly sizes");
        addParamsLine("                               : (findGoodCropSize / findGoodPatchSize, movie_alignment_correlation_gpu.cpp:73-121, time cuFFT plans on the");
        addParamsLine("                               : installed card): the global alignment runs on the top-left window of the largest even size with prime");
        addParamsLine("                               : factors 2, 3, 5, 7 within 10 % below the frame, the patches are the smallest such size within 20 % above");
        addParamsLine("                               : the requested one");
        addExampleLine("xmipp_movie_alignment_correlation -i movie.xmd --oaligned alignedMovie.stk --oavg alignedMicrograph.mrc");
    }

    void readParams() override
    {
        // movie_alignment_correlation_base.cpp:31-68
        if (!checkParam("-i")) REPORT_ERROR(ERR_ARG_MISSING, "-i is mandatory");
        fnMovie = getParam("-i");
        fnOut = getParam("-o");
        fnInitialAvg = getParam("--oavgInitial");
        fnDark = getParam("--dark");
        fnGain = getParam("--gain");
        binning = (float)getDoubleParam("--bin");
        if (binning < 1.0) REPORT_ERROR(ERR_ARG_INCORRECT, "Binning must be >= 1");
        Ts = (float)getDoubleParam("--sampling") * binning;
        maxShift = (float)getDoubleParam("--maxShift") / Ts;
        maxResForCorrelation = (float)getDoubleParam("--maxResForCorrelation");
        fnAligned = getParam("--oaligned");
        fnAvg = getParam("--oavg");
        nfirst = (int)getIntParam("--frameRange", 0);
        nlast = (int)getIntParam("--frameRange", 1);
        nfirstSum = (int)getIntParam("--frameRangeSum", 0);
        nlastSum = (int)getIntParam("--frameRangeSum", 1);
        skipLocalAlignment = checkParam("--skipLocalAlignment");
        device = (int)getIntParam("--device");
        sizeSearch = getParam("--sizeSearch");
        if (sizeSearch != "off" && sizeSearch != "smooth") REPORT_ERROR(ERR_ARG_INCORRECT, "--sizeSearch is off or smooth");
        minLocalRes = (int)getIntParam("--minLocalRes");
        cpX = (int)getIntParam("--controlPoints", 0);
        cpY = (int)getIntParam("--controlPoints", 1);
        cpT = (int)getIntParam("--controlPoints", 2);
        if (cpX < 3 || cpY < 3 || cpT < 3) REPORT_ERROR(ERR_ARG_INCORRECT, "All control points has to be bigger than 2");
        if (checkParam("--patches")) { patchesX = (int)getIntParam("--patches", 0); patchesY = (int)getIntParam("--patches", 1); }
        patchesAvg = (int)getIntParam("--patchesAvg");
        if (patchesAvg < 1) REPORT_ERROR(ERR_ARG_INCORRECT, "Patch averaging has to be at least 1 (one).");
    }

    void run() override
    {
        // readMovie + correctLoopIndices (movie_alignment_correlation_base.cpp:317-337,447-458)
        MetaDataVec movie;
        const std::string ext = FileName(fnMovie).extension();
        if (ext == "xmd" || ext == "sel" || ext == "doc") movie.read(fnMovie);
        else {
            const ImageInfo I = readInfo(fnMovie);
            size_t nd = I.isStack ? I.n : 1;
            if (ext == "mrc" && nd == 1) nd = I.z;
            for (size_t i = 0; i < nd; ++i) movie.setValue("image", std::to_string(i + 1) + "@" + fnMovie, movie.addObject());
        }
        if (movie.size() < 2) REPORT_ERROR(ERR_MD_NOOBJ, "a movie needs at least two frames: " + fnMovie);
        nfirst = std::max(nfirst, 0);
        nfirstSum = std::max(nfirstSum, 0);
        if (nlast < 0) nlast = (int)movie.size() - 1;
        if (nlastSum < 0) nlastSum = (int)movie.size() - 1;
        if (nfirstSum < nfirst || nlastSum > nlast)         // checkSettings
            REPORT_ERROR(ERR_ARG_INCORRECT, "Summing frames that were not aligned is not allowed. Check the intervals of the alignment and summation "
                         "(--frameRange and --frameRangeSum).");
        if (nlast >= (int)movie.size() || nlast <= nfirst) REPORT_ERROR(ERR_ARG_INCORRECT, "--frameRange outside the movie");
        const int N = nlast - nfirst + 1;
        // loadDarkCorrection / loadGainCorrection (:268-284)
        std::vector<float> dark, gain, frame;
        ImageInfo Id, Ig, I;
        if (!fnDark.empty()) readImage(fnDark, dark, Id);
        if (!fnGain.empty()) {
            readImage(fnGain, gain, Ig);
            double avg = 0;
            for (float v : gain) avg += v;
            if (std::isinf(avg) || std::isnan(avg)) REPORT_ERROR(ERR_ARG_INCORRECT, "The input gain image is incorrect, it contains infinite or nan");
        }
        std::string fn0;
        movie.getValue("image", fn0, (size_t)nfirst);
        readImage(fn0, frame, I);
        if (!dark.empty() && (Id.x != I.x || Id.y != I.y)) REPORT_ERROR(ERR_ARG_INCORRECT, "The dark image size does not match the movie frame size.");
        if (!gain.empty() && (Ig.x != I.x || Ig.y != I.y)) REPORT_ERROR(ERR_ARG_INCORRECT, "The gain image size does not match the movie frame size.");
        // getMovieSize (movie_alignment_correlation_base.cpp:356-370): with --bin the program works on, and stores, binned frames
        const ImageInfo Iraw = I;
        const bool doBin = binning != 1.0f;
        if (doBin) {
            I.x = (size_t)(((float)Iraw.x / binning) / 2.f * 2.f);
            I.y = (size_t)(((float)Iraw.y / binning) / 2.f * 2.f);
            if (I.x < 8 || I.y < 8) REPORT_ERROR(ERR_ARG_INCORRECT, "--bin leaves frames of " + std::to_string(I.x) + " x " + std::to_string(I.y));
        }
        const size_t per = I.x * I.y, perRaw = Iraw.x * Iraw.y;
        // setNoOfPatches (:516-528), getRequestedPatchSize (base.h:217-219), checkSettings (:80-86)
        size_t reqPatch = (size_t)(minLocalRes / Ts);
        if (patchesX <= 0 || patchesY <= 0) {
            patchesX = (int)std::ceil((float)I.x / (float)reqPatch);
            patchesY = (int)std::ceil((float)I.y / (float)reqPatch);
        }
        if (!skipLocalAlignment && (patchesX <= cpX || patchesY <= cpY))
            REPORT_ERROR(ERR_LOGIC_ERROR, "More control points than patches. Decrease the number of control points.");
        // --sizeSearch smooth: the patch grows to an FFT-friendly size (findGoodPatchSize), the global alignment sees a cropped frame (findGoodCropSize)
        size_t cropX = I.x, cropY = I.y;
        if (sizeSearch == "smooth") {
            const size_t want = reqPatch;
            reqPatch = std::min(smoothSize(reqPatch, 20, false), std::min(I.x, I.y) & ~(size_t)1);
            if (!doBin) {       // (the reference crops only when it does not bin, GAOptimize :601)
                cropX = smoothSize(I.x, 10, true);
                cropY = I.x == I.y ? cropX : smoothSize(I.y, 10, true);        // squareOnly for square frames
            }
            if (verbose) std::cout << "Size search (smooth): patches of " << reqPatch << " px (requested " << want << "), global alignment on the top-left "
                                   << cropX << " x " << cropY << " of " << I.x << " x " << I.y << "\n";
        }
        if (verbose) std::cout << "Computing global alignment ...\n";
        CtxGuard g;
        xhCheck(xh_ctx_create_private(device, &g.c));
        std::vector<double> sx(N), sy(N), initial, average;
        const int nC = cpX * cpY * cpT;
        std::vector<double> coeffsX(nC), coeffsY(nC), centers, patchShifts;
        std::unique_ptr<StackWriter> alignedStack;
        int ref = 0, Nsum = 0;
        {
            xh_fa *fa = nullptr;
            xhCheck(xh_fa_create(g.c, (int)I.y, (int)I.x, Ts, maxResForCorrelation, &fa));
            struct FaGuard { xh_fa *f; ~FaGuard() { xh_fa_destroy(f); } } fg{fa};
            // the warp's prefilter of the frames runs while the host fits the spline (not with --oavgInitial: its sum leaves the prefilter pass)
            if (!skipLocalAlignment && fnInitialAvg.empty() && (!fnAvg.empty() || !fnAligned.empty())) xhCheck(xh_fa_set_option(fa, "prefilter_ahead", 1));
            DeviceBuffer d_frames, d_dark, d_gain, d_out, d_sum, d_initial;
            d_frames.reserve(g.c, (size_t)N * per * sizeof(float));
            // the frames are read a few ahead by their own threads while the one before them goes to the device (the reference loads
            // with a thread pool beside its two GPU streams, movie_alignment_correlation_gpu.cpp:667-691)
            // (frames of integer counts -- MRC modes 0, 1, 6 -- travel as they are and become floats on the device, xh_movie_frame_to_float)
            auto readFrame = [&](int n) {
                std::string fn;
                movie.getValue("image", fn, (size_t)(nfirst + n));
                std::vector<unsigned char> f;
                ImageInfo In;
                readImageRaw(fn, f, In);
                if (In.x != Iraw.x || In.y != Iraw.y || In.mode != Iraw.mode) REPORT_ERROR(ERR_MULTIDIM_SIZE, "frames of different sizes or data types in " + fnMovie);
                return f;
            };
            // --bin: a frame is corrected (dark, gain) and binned on its way in (loadFrame + CUDAFlexAlignScale::runScaleIFT,
            // movie_alignment_correlation_gpu.cpp:667-691); everything after works on binned frames without dark / gain
            xh_fft2d *planRaw = nullptr, *planBin = nullptr;
            struct PlanGuard { xh_fft2d **a, **b; ~PlanGuard() { if (*a) xh_fft2d_destroy(*a); if (*b) xh_fft2d_destroy(*b); } } planGuard{&planRaw, &planBin};
            DeviceBuffer d_raw, d_counts;
            const size_t rawBytes = perRaw * Iraw.bytesPerPixel();
            if (Iraw.mode != 2) d_counts.reserve(g.c, rawBytes);
            if (doBin) {
                xhCheck(xh_fft2d_create(g.c, (int)Iraw.y, (int)Iraw.x, &planRaw));
                xhCheck(xh_fft2d_create(g.c, (int)I.y, (int)I.x, &planBin));
                d_raw.reserve(g.c, perRaw * sizeof(float));
                if (!dark.empty()) { d_dark.reserve(g.c, perRaw * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_dark.p, dark.data(), perRaw * sizeof(float))); }
                if (!gain.empty()) { d_gain.reserve(g.c, perRaw * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_gain.p, gain.data(), perRaw * sizeof(float))); }
            }
            // four reader threads for the whole movie, reader w takes frames w, w + 4, ... (a thread per frame -- std::async -- opened and
            // parsed the stack anew for every frame: readImageRaw's header / stream cache is thread_local); a reader runs at most two
            // of its frames ahead of the consumer
            const int ahead = 4;
            std::vector<std::promise<std::vector<unsigned char>>> prom(N);
            std::vector<std::future<std::vector<unsigned char>>> fut;
            for (auto &pr : prom) fut.push_back(pr.get_future());
            std::mutex rdM;
            std::condition_variable rdCv;
            int consumed = 0;
            bool rdStop = false;
            std::vector<std::thread> readers;
            for (int w = 0; w < std::min(ahead, N); ++w)
                readers.emplace_back([&, w] {
                    for (int n = w; n < N; n += ahead) {
                        {
                            std::unique_lock<std::mutex> lk(rdM);
                            rdCv.wait(lk, [&] { return rdStop || n < consumed + 2 * ahead; });
                            if (rdStop) { prom[n].set_exception(std::make_exception_ptr(std::runtime_error("movie read cancelled"))); continue; }
                        }
                        try { prom[n].set_value(readFrame(n)); } catch (...) { prom[n].set_exception(std::current_exception()); }
                    }
                });
            struct JoinReaders {
                std::vector<std::thread> &t; std::mutex &m; std::condition_variable &cv; bool &stop;
                ~JoinReaders() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto &x : t) if (x.joinable()) x.join(); }
            } joinReaders{readers, rdM, rdCv, rdStop};
            for (int n = 0; n < N; ++n) {
                std::vector<unsigned char> cur = fut[n].get();                    // re-throws what the reader threw
                { std::lock_guard<std::mutex> lk(rdM); consumed = n + 1; }
                rdCv.notify_all();
                float *dst = doBin ? d_raw.as<float>() : d_frames.as<float>() + (size_t)n * per;
                if (Iraw.mode == 2) xhCheck(xh_memcpy_h2d(g.c, dst, cur.data(), rawBytes));
                else {
                    xhCheck(xh_memcpy_h2d(g.c, d_counts.p, cur.data(), rawBytes));
                    xhCheck(xh_movie_frame_to_float(g.c, d_counts.p, Iraw.mode, (int64_t)perRaw, dst));
                }
                if (doBin)
                    xhCheck(xh_movie_bin_frame(g.c, planRaw, planBin, d_raw.as<float>(), dark.empty() ? nullptr : d_dark.as<float>(), gain.empty() ? nullptr : d_gain.as<float>(),
                                               (int)Iraw.y, (int)Iraw.x, d_frames.as<float>() + (size_t)n * per, (int)I.y, (int)I.x));
            }
            if (!doBin) {
                if (!dark.empty()) { d_dark.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_dark.p, dark.data(), per * sizeof(float))); }
                if (!gain.empty()) { d_gain.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_gain.p, gain.data(), per * sizeof(float))); }
            }
            const float *pd = (dark.empty() || doBin) ? nullptr : d_dark.as<float>(), *pg = (gain.empty() || doBin) ? nullptr : d_gain.as<float>();
            if (cropX != I.x || cropY != I.y) {
                // getCroppedFrame (:727-734): the correlations see the top-left window of every frame (and of dark / gain); everything after
                // the global alignment works on the whole frames again
                xh_fa *faCrop = nullptr;
                xhCheck(xh_fa_create(g.c, (int)cropY, (int)cropX, Ts, maxResForCorrelation, &faCrop));
                struct FaGuard2 { xh_fa *f; ~FaGuard2() { xh_fa_destroy(f); } } fg2{faCrop};
                DeviceBuffer d_crop, d_dc, d_gc;
                d_crop.reserve(g.c, (size_t)N * cropX * cropY * sizeof(float));
                xhCheck(xh_movie_crop_frames(g.c, d_frames.as<float>(), N, (int)I.y, (int)I.x, (int)cropY, (int)cropX, d_crop.as<float>()));
                if (pd) { d_dc.reserve(g.c, cropX * cropY * sizeof(float)); xhCheck(xh_movie_crop_frames(g.c, pd, 1, (int)I.y, (int)I.x, (int)cropY, (int)cropX, d_dc.as<float>())); }
                if (pg) { d_gc.reserve(g.c, cropX * cropY * sizeof(float)); xhCheck(xh_movie_crop_frames(g.c, pg, 1, (int)I.y, (int)I.x, (int)cropY, (int)cropX, d_gc.as<float>())); }
                xhCheck(xh_fa_global_alignment(faCrop, d_crop.as<float>(), N, pd ? d_dc.as<float>() : nullptr, pg ? d_gc.as<float>() : nullptr, maxShift, nullptr, nullptr,
                                               sx.data(), sy.data(), &ref));
            } else
            xhCheck(xh_fa_global_alignment(fa, d_frames.as<float>(), N, pd, pg, maxShift, nullptr, nullptr, sx.data(), sy.data(), &ref));
            centers.resize((size_t)patchesX * patchesY * 2);
            const bool wantAligned = !fnAligned.empty(), wantAvg = !fnAvg.empty(), wantInitial = !fnInitialAvg.empty();
            if ((patchesX < 2 || patchesY < 2) && (!skipLocalAlignment || wantAligned || wantAvg))
                REPORT_ERROR(ERR_LOGIC_ERROR, "The movie is too small for patches of " + std::to_string(reqPatch) + " px: give --patches and --minLocalRes");
            if (skipLocalAlignment && !(wantAligned || wantAvg)) {
                // only the shifts (and the plain sum) are asked for: no spline needed
            } else if (skipLocalAlignment) {
                // applyShiftsComputeAverage(globAlignment) -> localFromGlobal (movie_alignment_correlation_gpu.cpp:432-465)
                xhCheck(xh_fa_local_from_global(fa, N, sx.data(), sy.data(), patchesX, patchesY, (int)reqPatch, (int)reqPatch, cpX, cpY, cpT, centers.data(), coeffsX.data(),
                                                coeffsY.data()));
            } else {
                if (verbose) std::cout << "Computing local alignment ...\n";
                patchShifts.resize((size_t)patchesX * patchesY * N * 2);
                int dims[4];
                xhCheck(xh_fa_local_alignment(fa, d_frames.as<float>(), N, pd, pg, sx.data(), sy.data(), ref, maxShift, patchesX, patchesY, (int)reqPatch, (int)reqPatch,
                                              patchesAvg, cpX, cpY, cpT, patchShifts.data(), centers.data(), coeffsX.data(), coeffsY.data(), dims));
                if (verbose) std::cout << "Patches: " << patchesX << " x " << patchesY << " of " << dims[0] << " x " << dims[1] << " px, correlated at " << dims[2] << " x " << dims[3] << "\n";
            }
            // applyShiftsComputeAverage (:479-570): every summed frame warped by the B-spline, the sums kept on the device
            if (wantAligned || wantAvg || wantInitial) {
                if (wantAligned) {
                    d_out.reserve(g.c, per * sizeof(float));
                    // frame fi goes to slot fi - nfirst of the stack (tmp.write(fnAligned, frameOffset + 1, true, WRITE_REPLACE),
                    // movie_alignment_correlation_gpu.cpp:528,540): with --frameRangeSum inside --frameRange the leading slots stay empty
                    alignedStack.reset(new StackWriter(fnAligned, I.x, I.y, (size_t)(nlastSum - nfirst + 1)));
                }
                std::vector<float> host(per, 0.f);
                if (wantAvg) { d_sum.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_sum.p, host.data(), per * sizeof(float))); }
                if (wantInitial) { d_initial.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_initial.p, host.data(), per * sizeof(float))); }
                if (!wantAligned) {
                    // only sums are asked for: the whole loop in one call
                    xhCheck(xh_fa_apply_bspline_frames(fa, d_frames.as<float>(), N, nfirstSum - nfirst, nlastSum - nfirst, pd, pg, coeffsX.data(), coeffsY.data(), cpX, cpY, cpT,
                                                       nullptr, wantAvg ? d_sum.as<float>() : nullptr, wantInitial ? d_initial.as<float>() : nullptr));
                    Nsum = nlastSum - nfirstSum + 1;
                } else
                for (int fi = nfirstSum; fi <= nlastSum; ++fi) {
                    const int off = fi - nfirst;
                    xhCheck(xh_fa_apply_bspline(fa, d_frames.as<float>() + (size_t)off * per, pd, pg, coeffsX.data(), coeffsY.data(), cpX, cpY, cpT, N, off,
                                                wantAligned ? d_out.as<float>() : nullptr, wantAvg ? d_sum.as<float>() : nullptr, wantInitial ? d_initial.as<float>() : nullptr));
                    if (wantAligned) {
                        xhCheck(xh_memcpy_d2h(g.c, host.data(), d_out.p, per * sizeof(float)));
                        alignedStack->write((size_t)off, host.data());
                    }
                    ++Nsum;
                }
                if (wantAligned) alignedStack->finish();
                if (wantAvg) { xhCheck(xh_memcpy_d2h(g.c, host.data(), d_sum.p, per * sizeof(float))); average.assign(host.begin(), host.end()); }
                if (wantInitial) { xhCheck(xh_memcpy_d2h(g.c, host.data(), d_initial.p, per * sizeof(float))); initial.assign(host.begin(), host.end()); }
            }
        }
        // storeGlobalShifts (:364-396): the shift that should be applied is the negative of the estimated one
        for (size_t id = 0; id < movie.size(); ++id) {
            const int n = (int)id;
            if (n >= nfirst && n <= nlast) {
                movie.setValue("shiftX", (double)(sx[n - nfirst] * -1) * binning, id);
                movie.setValue("shiftY", (double)(sy[n - nfirst] * -1) * binning, id);
                movie.setValue("enabled", (long)1, id);
            } else {
                movie.setValue("enabled", (long)-1, id);
                movie.setValue("shiftX", 0.0, id);
                movie.setValue("shiftY", 0.0, id);
            }
            movie.setValue("weight", 1.0, id);
        }
        if (verbose) {        // printGlobalShift
            std::cout << "Reference frame: " << ref << "\nEstimated global shifts (in px, from the reference frame):\n";
            for (int n = 0; n < N; ++n) printf("X: %07.4f Y: %07.4f\n", sx[n] * binning, sy[n] * binning);
            std::cout << std::endl;
        }
        const std::string out = fnOut.empty() ? fnMovie : fnOut;
        MetaDataVec mdIref;
        mdIref.setValue("ref", (long)(nfirst + ref), mdIref.addObject());
        mdIref.write("referenceFrame@" + out, true);           // MD_APPEND (movie_alignment_correlation_base.cpp:346): other blocks of an existing file stay
        if (!skipLocalAlignment) {
            // storeResults(localAlignment) (:460-514): how far the spline moves the patch centres from the global shift (2.5 % and
            // 97.5 % of the sorted distances), patches, coefficients, control points
            std::vector<double> dist;
            const int nP = patchesX * patchesY;
            for (int p = 0; p < nP; ++p)
                for (int t = 0; t < N; ++t) {
                    double bx, by;
                    splineShift(coeffsX, coeffsY, (int)I.x, (int)I.y, N, (int)centers[2 * p], (int)centers[2 * p + 1], t, bx, by);
                    // BSplineHelper::getShift(grid, ...) hands (shiftX, shiftY) to a callee declared (T &shiftY, T &shiftX, ...)
                    // (bspline_helper.cpp:99 against :112), so the pair storeResults receives is (Y, X): the reference's two
                    // confidence scalars are hypot(splineY - globalX, splineX - globalY).  Reproduced: same outputs.
                    dist.push_back(std::hypot(by - sx[t], bx - sy[t]));
                }
            std::sort(dist.begin(), dist.end());
            MetaDataVec md;
            const size_t id = md.addObject();
            md.setValue("localAlignmentConf2_5Perc", dist.at((size_t)(dist.size() * 0.025)), id);
            md.setValue("localAlignmentConf97_5Perc", dist.at((size_t)(dist.size() * 0.975)), id);
            md.setValue("localAlignmentPatches", "[ " + std::to_string(patchesX) + " " + std::to_string(patchesY) + " ]", id);
            auto vec = [](const std::vector<double> &v) { std::string s = "[ "; char b[64]; for (double x : v) { snprintf(b, sizeof(b), "%.6f ", x); s += b; } return s + "]"; };
            md.setValue("localAlignmentCoeffsX", vec(coeffsX), id);
            md.setValue("localAlignmentCoeffsY", vec(coeffsY), id);
            md.setValue("localAlignmentControlPoints", "[ " + std::to_string(cpX) + " " + std::to_string(cpY) + " " + std::to_string(cpT) + " ]", id);
            md.write("localAlignment@" + out, true);
        }
        // storeResults (:421-433)
        if (!fnInitialAvg.empty()) {
            for (double &v : initial) v /= Nsum;
            writeVolume(fnInitialAvg, initial.data(), I.x, I.y, 1);
        }
        if (!fnAvg.empty()) {
            for (double &v : average) v /= Nsum;
            writeVolume(fnAvg, average.data(), I.x, I.y, 1);
        }
        movie.write("frameShifts@" + out, true);
    }
};

// xmipp_movie_filter_dose: ProgMovieFilterDose (reconstruction/movie_filter_dose.cpp:36-290) over xh_movie_dose_filter
class ProgMovieFilterDose : public XmippProgram {
public:
    std::string fnIn, fnOut;
    int first = -1, last = -1, device = 0;
    double pixel_size = 1, dose_per_frame = 2, acceleration_voltage = 300, pre_exposure_amount = 0;

    void defineParams() override
    {
        // movie_filter_dose.cpp:68-83, verbatim parameter lines
        addUsageLine("Align a set of frames by cross-correlation of the frames");
        addParamsLine("   -i <movie>                  : input movie");
        addParamsLine("  [-o <fn=\"out.mrcs\">]        : output filtered movie.");
        addParamsLine("  [--frameRange <n0=-1> <nF=-1>]  : First and last frame to align, frame numbers start at 0");
        addParamsLine("  [--sampling <Ts=1>]          : Sampling rate (A/pixel)");
        addParamsLine("  [--dosePerFrame <dose=2>]    : Dose per frame (e/A^2)");
        addParamsLine("  [--accVoltage <voltage=300>] : Acceleration voltage (kV) min_value=200.0e0,max_value=300.0e0)");
        addParamsLine("  [--preExposure <preExp=0>]   : P (e/A^2)");
        addParamsLine("  [--device <id=0>]            : HIP device");
    }

    void readParams() override
    {
        if (!checkParam("-i")) REPORT_ERROR(ERR_ARG_MISSING, "-i is mandatory");
        fnIn = getParam("-i");
        fnOut = getParam("-o");
        first = (int)getIntParam("--frameRange", 0);
        last = (int)getIntParam("--frameRange", 1);
        pixel_size = getDoubleParam("--sampling");
        dose_per_frame = getDoubleParam("--dosePerFrame");
        acceleration_voltage = getDoubleParam("--accVoltage");
        if (!((acceleration_voltage < 301 && acceleration_voltage > 299.) || (acceleration_voltage < 201.0 && acceleration_voltage > 199.0)))
            REPORT_ERROR(ERR_ARG_INCORRECT, "Bad acceleration voltage (must be 200 or 300 kV");          // initVoltage
        pre_exposure_amount = getDoubleParam("--preExposure");
        device = (int)getIntParam("--device");
    }

    void run() override
    {
        MetaDataVec movie;
        const std::string ext = FileName(fnIn).extension();
        if (ext == "xmd" || ext == "sel" || ext == "doc") movie.read(fnIn);
        else {
            const ImageInfo I = readInfo(fnIn);
            size_t nd = I.isStack ? I.n : 1;
            if (ext == "mrc" && nd == 1) nd = I.z;
            for (size_t i = 0; i < nd; ++i) movie.setValue("image", std::to_string(i + 1) + "@" + fnIn, movie.addObject());
        }
        if (first < 0) first = 0;
        if (last < 0) last = (int)movie.size() - 1;
        if (last >= (int)movie.size() || last < first) REPORT_ERROR(ERR_ARG_INCORRECT, "--frameRange outside the movie");
        CtxGuard g;
        xhCheck(xh_ctx_create_private(device, &g.c));
        std::vector<float> frame;
        ImageInfo I0;
        xh_fft2d *plan = nullptr;
        struct PlanGuard { xh_fft2d **p; ~PlanGuard() { if (*p) xh_fft2d_destroy(*p); } } pg{&plan};
        DeviceBuffer d_frame;
        std::unique_ptr<StackWriter> stack;            // the frames keep their place in the stack (frame.write(fn, n + 1, ...)); one frame on the host at a time
        for (int n = 0; n <= last; ++n) {
            if (n < first) continue;
            std::string fn;
            movie.getValue("image", fn, (size_t)n);
            ImageInfo I;
            readImage(fn, frame, I);
            if (!plan) {
                I0 = I;
                xhCheck(xh_fft2d_create(g.c, (int)I.y, (int)I.x, &plan));
                d_frame.reserve(g.c, I.x * I.y * sizeof(float));
                stack.reset(new StackWriter(fnOut, I.x, I.y, (size_t)(last + 1)));
            } else if (I.x != I0.x || I.y != I0.y) REPORT_ERROR(ERR_MULTIDIM_SIZE, "frames of different sizes in " + fnIn);
            const size_t per = I.x * I.y;
            xhCheck(xh_memcpy_h2d(g.c, d_frame.p, frame.data(), per * sizeof(float)));
            xhCheck(xh_movie_dose_filter(g.c, plan, d_frame.as<float>(), (int)I.y, (int)I.x, pixel_size, acceleration_voltage, (n * dose_per_frame) + pre_exposure_amount,
                                         ((n + 1) * dose_per_frame) + pre_exposure_amount));
            xhCheck(xh_memcpy_d2h(g.c, frame.data(), d_frame.p, per * sizeof(float)));
            stack->write((size_t)n, frame.data());
        }
        if (stack) stack->finish();
    }
};

}  // namespace mc
#endif
