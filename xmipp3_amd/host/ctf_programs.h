// ctf_programs.h -- host side of the CTF pre-steps: the reference's programs with their own flags over
// the C ABI (xh_ctfop_*). xmipp_ctf_phase_flip = ProgCTFPhaseFlipping (reconstruction/ctf_phase_flip.{h,cpp}),
// xmipp_ctf_correct_wiener2d = ProgCorrectWiener2D (reconstruction/ctf_correct_wiener2d.{h,cpp}, an XmippMetadataProgram).
#ifndef XMIPP3_AMD_CTF_PROGRAMS_H
#define XMIPP3_AMD_CTF_PROGRAMS_H
#include "programs.h"

namespace mc {

// CTFDescription::readFromMdRow (data/ctf.cpp:389-470): the columns a ctfparam file / a particle row carries
inline void readCtfRow(const MetaDataVec &md, size_t id, xh_ctf_params &c)
{
    xh_ctf_defaults(&c);     // data/ctf.cpp:365-388
    c.Tm = md.getDouble("ctfSamplingRate", id, 1); c.kV = md.getDouble("ctfVoltage", id, 100);
    c.DeltafU = md.getDouble("ctfDefocusU", id, 0); c.DeltafV = md.getDouble("ctfDefocusV", id, c.DeltafU);
    c.azimuthal_angle = md.getDouble("ctfDefocusAngle", id, 0); c.Cs = md.getDouble("ctfSphericalAberration", id, 0);
    c.Ca = md.getDouble("ctfChromaticAberration", id, 0); c.espr = md.getDouble("ctfEnergyLoss", id, 0);
    c.ispr = md.getDouble("ctfLensStability", id, 0); c.alpha = md.getDouble("ctfConvergenceCone", id, 0);
    c.DeltaF = md.getDouble("ctfLongitudinalDisplacement", id, 0); c.DeltaR = md.getDouble("ctfTransversalDisplacement", id, 0);
    c.Q0 = md.getDouble("ctfQ0", id, 0); c.K = md.getDouble("ctfK", id, 1);
    c.envR0 = md.getDouble("ctfEnvR0", id, 0); c.envR1 = md.getDouble("ctfEnvR1", id, 0); c.envR2 = md.getDouble("ctfEnvR2", id, 0);
    c.phase_shift = md.getDouble("ctfPhaseShift", id, 0); c.VPP_radius = md.getDouble("ctfVPPRadius", id, 0);
}

struct CtxGuard {
    xh_ctx *c = nullptr;
    ~CtxGuard() { if (c) xh_ctx_destroy(c); }
};

class ProgCTFPhaseFlipping : public XmippProgram {
public:
    std::string fn_in, fn_out, fnt_ctf;
    double downsampling = 1, Tm = -1;
    int device = 0;

    void defineParams() override
    {
        // ctf_phase_flip.cpp:29-41, verbatim parameter lines
        addUsageLine("Correct the phase of micrographs");
        addUsageLine("+This program flips the phase of those frequencies that were already ");
        addUsageLine("+flipped by the CTF. Flipping the phase at the level of the micrograph is recommended.");
        addParamsLine(" -i <file>               : Input micrograph");
        addParamsLine(" -o <file>               : Output micrograph");
        addParamsLine(" --ctf <ctfparam_file>   : CTF description");
        addParamsLine(" [--sampling <T=-1>]     : Sampling rate of the input micrograph.");
        addParamsLine("                         : If not given, then it is assumed to be the sampling rate in the ctfparam times the downsampling.");
        addParamsLine(" [--downsampling <D=1>]  : Downsampling factor of the input micrograph with respect to the original");
        addParamsLine("                         : micrograph.");
        addParamsLine(" [--device <id=0>]       : HIP device");
    }

    void readParams() override
    {
        if (!checkParam("-i")) REPORT_ERROR(ERR_ARG_MISSING, "-i is mandatory");
        if (!checkParam("-o")) REPORT_ERROR(ERR_ARG_MISSING, "-o is mandatory");
        if (!checkParam("--ctf")) REPORT_ERROR(ERR_ARG_MISSING, "--ctf is mandatory");
        fn_in = getParam("-i");
        fn_out = getParam("-o");
        fnt_ctf = getParam("--ctf");
        downsampling = getDoubleParam("--downsampling");
        Tm = getDoubleParam("--sampling");
        device = (int)getIntParam("--device");
    }

    void show()
    {
        if (verbose == 0) return;
        std::cout << "input_micrograph:      " << fn_in << std::endl
                  << "output_micrograph:     " << fn_out << std::endl
                  << "ctf_param_file:        " << fnt_ctf << std::endl
                  << "sampling:              " << Tm << std::endl
                  << "downsampling:          " << downsampling << std::endl;
    }

    void run() override
    {
        show();
        // ctf_phase_flip.cpp:65-86
        std::vector<float> img;
        ImageInfo I;
        readImage(fn_in, img, I);
        if (I.z != 1) REPORT_ERROR(ERR_MULTIDIM_SIZE, "xmipp_ctf_phase_flip works on 2-D micrographs");
        MetaDataVec md;
        md.read(fnt_ctf);
        if (md.size() == 0) REPORT_ERROR(ERR_MD_NOOBJ, "no CTF description in " + fnt_ctf);
        xh_ctf_params c;
        readCtfRow(md, 0, c);
        const double sampling = Tm < 0 ? c.Tm * downsampling : Tm;       // changeSamplingRate
        CtxGuard g;
        xhCheck(xh_ctx_create_private(device, &g.c));
        {
            DeviceBuffer d;
            d.reserve(g.c, img.size() * sizeof(float));
            xhCheck(xh_memcpy_h2d(g.c, d.p, img.data(), img.size() * sizeof(float)));
            xh_ctfop *op = nullptr;
            xhCheck(xh_ctfop_create(g.c, (int)I.y, (int)I.x, 1.0, &op));
            const int rc = xh_ctfop_phase_flip(op, d.as<float>(), &c, sampling);
            if (rc == XH_OK) xhCheck(xh_memcpy_d2h(g.c, img.data(), d.p, img.size() * sizeof(float)));
            xh_ctfop_destroy(op);
            xhCheck(rc);
        }
        std::vector<double> out(img.begin(), img.end());
        writeVolume(fn_out, out.data(), I.x, I.y, 1);
    }
};

class ProgCorrectWiener2D : public XmippProgram {
public:
    std::string fn_in, fn_out;
    bool phase_flipped = false, isIsotropic = false, correct_envelope = false;
    double pad = 2, wiener_constant = -1, sampling_rate = 1;
    int device = 0, batch = 256;

    void defineParams() override
    {
        // ctf_correct_wiener2d.cpp:42-55 over the -i / -o of XmippMetadataProgram (each image produces an output)
        addUsageLine("Perform CTF correction to 2D projection images with estimated ctfs using a Wiener filter.");
        addParamsLine("   -i <metadata>           : Input images with their CTF columns (ctfDefocusU, ctfVoltage, ...)");
        addParamsLine("   -o <stack>              : Output stack; the output metadata is written next to it with the extension xmd");
        addParamsLine("   [--phase_flipped]       : Is the data already phase-flipped?");
        addParamsLine("   [--isIsotropic]         : Must be considered the defocus isotropic?");
        addParamsLine("   [--sampling_rate <float=1.0>]     : Sampling rate of the input particles");
        addParamsLine("   [--wc <float=-1>]       : Wiener-filter constant (if < 0: use FREALIGN default)");
        addParamsLine("   [--pad <factor=2.> ]    : Padding factor for Wiener correction");
        addParamsLine("   [--correct_envelope]     : Correct the CTF envelope");
        addParamsLine("   [--device <id=0>]       : HIP device");
        addParamsLine("   [--batch <n=256>]       : Images per device batch");
    }

    void readParams() override
    {
        if (!checkParam("-i")) REPORT_ERROR(ERR_ARG_MISSING, "-i is mandatory");
        if (!checkParam("-o")) REPORT_ERROR(ERR_ARG_MISSING, "-o is mandatory");
        fn_in = getParam("-i");
        fn_out = getParam("-o");
        // ctf_correct_wiener2d.cpp:30-39
        phase_flipped = checkParam("--phase_flipped");
        pad = std::max(1., getDoubleParam("--pad"));
        isIsotropic = checkParam("--isIsotropic");
        wiener_constant = getDoubleParam("--wc");
        correct_envelope = checkParam("--correct_envelope");
        sampling_rate = getDoubleParam("--sampling_rate");
        device = (int)getIntParam("--device");
        batch = std::max(1, (int)getIntParam("--batch"));
    }

    void run() override
    {
        MetaDataVec md;
        md.read(fn_in);
        const size_t n = md.size();
        if (n == 0) REPORT_ERROR(ERR_MD_NOOBJ, "no images in " + fn_in);
        if (!md.containsLabel("ctfDefocusU")) REPORT_ERROR(ERR_MD_BADLABEL, "the input metadata carries no CTF columns (ctfDefocusU ...)");
        std::string fn0;
        md.getValue("image", fn0, 0);
        const ImageInfo I0 = readInfo(fn0);
        const size_t per = I0.x * I0.y;
        CtxGuard g;
        xhCheck(xh_ctx_create_private(device, &g.c));
        xh_ctfop *op = nullptr;
        xhCheck(xh_ctfop_create(g.c, (int)I0.y, (int)I0.x, pad, &op));
        struct OpGuard { xh_ctfop *o; ~OpGuard() { xh_ctfop_destroy(o); } } og{op};
        // one device batch on the host at a time: every corrected batch goes into its slots of the output stack as soon as it is back
        // (the reference reads, filters and writes image by image; a set of 1e6 particles does not fit a host buffer)
        std::vector<float> hb((size_t)batch * per), one;
        StackWriter stack(fn_out, I0.x, I0.y, n);
        DeviceBuffer d;
        d.reserve(g.c, (size_t)batch * per * sizeof(float));
        for (size_t b0 = 0; b0 < n; b0 += (size_t)batch) {
            const size_t m = std::min((size_t)batch, n - b0);
            std::vector<xh_ctf_params> ctfs(m);
            for (size_t k = 0; k < m; ++k) {
                std::string fn;
                md.getValue("image", fn, b0 + k);
                ImageInfo I;
                readImage(fn, one, I);
                if (I.x != I0.x || I.y != I0.y) REPORT_ERROR(ERR_MULTIDIM_SIZE, "images of different sizes in " + fn_in);
                std::copy(one.begin(), one.end(), hb.begin() + k * per);
                readCtfRow(md, b0 + k, ctfs[k]);        // ctf.readFromMdRow(rowIn), wiener2d.cpp:148
            }
            xhCheck(xh_memcpy_h2d(g.c, d.p, hb.data(), m * per * sizeof(float)));
            xhCheck(xh_ctfop_wiener2d(op, d.as<float>(), (int)m, ctfs.data(), sampling_rate, phase_flipped, isIsotropic, wiener_constant, correct_envelope));
            xhCheck(xh_memcpy_d2h(g.c, hb.data(), d.p, m * per * sizeof(float)));
            for (size_t k = 0; k < m; ++k) stack.write(b0 + k, hb.data() + k * per);
        }
        stack.finish();
        // postProcess (ctf_correct_wiener2d.cpp:58-93): the image column points at the corrected stack, the CTF columns go
        MetaDataVec out;
        const char *drop[] = {"ctfDefocusA", "ctfDefocusU", "ctfDefocusAngle", "ctfDefocusV", "ctfBgBaseline", "ctfBgGaussian2Angle",
                              "ctfBgGaussian2CU", "ctfBgGaussian2CV", "ctfBgGaussian2K", "ctfBgGaussian2SigmaU", "ctfBgGaussian2SigmaV",
                              "ctfBgGaussianAngle", "ctfBgGaussianCU", "ctfBgGaussianCV", "ctfBgGaussianK", "ctfBgGaussianSigmaU",
                              "ctfBgGaussianSigmaV", "ctfBgSqrtAngle", "ctfBgSqrtK", "ctfBgSqrtU", "ctfBgSqrtV", "ctfChromaticAberration",
                              "ctfConvergenceCone", "ctfEnergyLoss", "ctfEnvelope", "ctfLensStability", "ctfTransversalDisplacement",
                              "ctfLongitudinalDisplacement", "ctfK"};
        std::vector<std::string> keep;
        for (const std::string &l : md.labels) {
            bool dropped = false;
            for (const char *q : drop) dropped = dropped || l == q;
            if (!dropped) keep.push_back(l);
        }
        for (size_t i = 0; i < n; ++i) {
            const size_t id = out.addObject();
            for (const std::string &l : keep) {
                if (l == "image") { out.setValue("image", std::to_string(i + 1) + "@" + fn_out, id); continue; }
                std::string v;
                md.getValue(l, v, i);
                out.setValue(l, v, id);
            }
        }
        FileName fo(fn_out);
        const std::string stem = fo.path.substr(0, fo.path.find_last_of('.'));
        out.write(stem + ".xmd");
    }
};

}  // namespace mc
#endif
