// movie_programs.h -- host side of FlexAlign's global alignment: ProgMovieAlignmentCorrelation's command
// line (reconstruction/movie_alignment_correlation_base.cpp:36-150,560-620) over xh_fa_*. The local (patch) alignment and the
// aligned outputs (--oavg, --oaligned: translate(BSPLINE3, DONT_WRAP) per frame) are not on the device path yet and fail loudly.
#ifndef XMIPP3_AMD_MOVIE_PROGRAMS_H
#define XMIPP3_AMD_MOVIE_PROGRAMS_H
#include "ctf_programs.h"

namespace mc {

class ProgMovieAlignmentCorrelation : public XmippProgram {
public:
    std::string fnMovie, fnOut, fnInitialAvg, fnDark, fnGain, fnAligned, fnAvg;
    float binning = 1, Ts = 1, maxShift = 50, maxResForCorrelation = 30;
    int nfirst = -1, nlast = -1, nfirstSum = -1, nlastSum = -1, device = 0;
    bool skipLocalAlignment = false;

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
        addParamsLine("  [--device <id=0>]            : HIP device");
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
        if (binning != 1.0) REPORT_ERROR(ERR_ARG_INCORRECT, "Binning is not supported. Please contact developers if you really need it.");       // movie_alignment_correlation.cpp:41-42
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
        if (!skipLocalAlignment)
            REPORT_ERROR(ERR_NOT_IMPLEMENTED, "the local (patch) alignment is not available on the device path: give --skipLocalAlignment "
                         "(the reference's CPU program has none either, movie_alignment_correlation.cpp:63-76)");
        if (!fnAligned.empty() || !fnAvg.empty())
            REPORT_ERROR(ERR_NOT_IMPLEMENTED, "--oaligned / --oavg (translate with BSPLINE3 and DONT_WRAP per frame) are not available on the device path yet; "
                         "the shifts are written to -o");
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
        const size_t per = I.x * I.y;
        if (!dark.empty() && (Id.x != I.x || Id.y != I.y)) REPORT_ERROR(ERR_ARG_INCORRECT, "The dark image size does not match the movie frame size.");
        if (!gain.empty() && (Ig.x != I.x || Ig.y != I.y)) REPORT_ERROR(ERR_ARG_INCORRECT, "The gain image size does not match the movie frame size.");
        if (verbose) std::cout << "Computing global alignment ...\n";
        CtxGuard g;
        xhCheck(xh_ctx_create_private(device, &g.c));
        std::vector<double> sx(N), sy(N), initial;
        int ref = 0;
        {
            xh_fa *fa = nullptr;
            xhCheck(xh_fa_create(g.c, (int)I.y, (int)I.x, Ts, maxResForCorrelation, &fa));
            struct FaGuard { xh_fa *f; ~FaGuard() { xh_fa_destroy(f); } } fg{fa};
            DeviceBuffer d_frames, d_dark, d_gain;
            d_frames.reserve(g.c, (size_t)N * per * sizeof(float));
            if (!fnInitialAvg.empty()) initial.assign(per, 0.0);
            for (int n = 0; n < N; ++n) {
                std::string fn;
                movie.getValue("image", fn, (size_t)(nfirst + n));
                ImageInfo In;
                readImage(fn, frame, In);
                if (In.x != I.x || In.y != I.y) REPORT_ERROR(ERR_MULTIDIM_SIZE, "frames of different sizes in " + fnMovie);
                xhCheck(xh_memcpy_h2d(g.c, (char *)d_frames.p + (size_t)n * per * sizeof(float), frame.data(), per * sizeof(float)));
                if (!initial.empty() && nfirst + n >= nfirstSum && nfirst + n <= nlastSum)
                    for (size_t k = 0; k < per; ++k) initial[k] += ((double)frame[k] - (dark.empty() ? 0.0 : dark[k])) * (gain.empty() ? 1.0 : gain[k]);
            }
            if (!dark.empty()) { d_dark.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_dark.p, dark.data(), per * sizeof(float))); }
            if (!gain.empty()) { d_gain.reserve(g.c, per * sizeof(float)); xhCheck(xh_memcpy_h2d(g.c, d_gain.p, gain.data(), per * sizeof(float))); }
            xhCheck(xh_fa_global_alignment(fa, d_frames.as<float>(), N, dark.empty() ? nullptr : d_dark.as<float>(), gain.empty() ? nullptr : d_gain.as<float>(),
                                           maxShift, nullptr, nullptr, sx.data(), sy.data(), &ref));
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
        if (!fnInitialAvg.empty()) {     // storeResults (:421-426)
            const int Ninitial = std::min(nlastSum, nlast) - std::max(nfirstSum, nfirst) + 1;
            for (double &v : initial) v /= Ninitial;
            writeVolume(fnInitialAvg, initial.data(), I.x, I.y, 1);
        }
        const std::string out = fnOut.empty() ? fnMovie : fnOut;
        MetaDataVec mdIref;
        mdIref.setValue("ref", (long)(nfirst + ref), mdIref.addObject());
        mdIref.write("referenceFrame@" + out, false);
        movie.write("frameShifts@" + out, true);
    }
};

}  // namespace mc
#endif
