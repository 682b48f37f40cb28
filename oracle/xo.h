/*
 * oracle/xo.h -- CPU restatement ("oracle") of the Xmipp projection-matching +
 * Fourier-gridding hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (xmipp3_amd/, libxmipp_hip.so) never
 * links, imports or calls anything declared here.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/src/xmipp/libraries unless noted):
 *   APM  = reconstruction/angular_projection_matching.cpp
 *   RFA  = reconstruction/reconstruct_fourier_accel.cpp
 *   POL  = data/polar.{h,cpp}
 *   FIL  = data/filters.cpp
 *   BLB  = data/blobs.cpp
 *   CTF  = data/ctf.{h,cpp}
 *
 * xmippCore (MultidimArray, FourierTransformer/FFTW, B-spline transforms,
 * correlation_matrix, Euler matrices, Bessel functions) is NOT in the
 * reference tree (I2PC/xmippCore @ tag v4, version-info.json:7-10): those
 * pieces are restated from the published algorithms and pinned by the
 * reference's own unit-test known answers (tests/test_oracle_pins.py).
 * Whole-program outputs have no in-tree goldens => "parity unpinned" for
 * those.
 *
 * All arrays are plain row-major; complex data are interleaved (re,im).
 */
#ifndef XO_H
#define XO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- FFT with xmippCore FourierTransformer conventions -------------------
 * forward: divided by N (all dims); inverse: un-normalised.
 * Pinned by applications/tests/function_tests/test_fftw_main.cpp:35-51. */
void xo_fft1d_r2c(const double *in, int n, double *out /* (n/2+1)*2 */);
void xo_fft1d_c2r(const double *in /* (n/2+1)*2 */, int n, double *out);
void xo_fft2d_r2c(const double *in, int ydim, int xdim, double *out /* ydim*(xdim/2+1)*2 */);
void xo_fft2d_c2r(const double *in, int ydim, int xdim, double *out);
void xo_fft3d_c2r(const double *in /* z*y*(x/2+1)*2 */, int zdim, int ydim, int xdim, double *out);
/* plain complex transform, sign=-1 forward / +1 inverse, never normalised */
void xo_fft1d_c2c(const double *in, int n, int sign, double *out);
/* FFT_IDX2DIGFREQ, test_fftw_main.cpp:80-109 */
double xo_fft_idx2digfreq(int idx, int size);

/* ---- cubic B-spline (xmippCore produceSplineCoefficients / interpolatedElementBSpline2D)
 * in-tree evidence: reconstruction_cuda/cuda_gpu_iirconvolve.cu:28-41,113-247,
 * cuda_gpu_multidim_array.cu:78-157, cuda_gpu_bilib.cu:16-25 */
void xo_bspline3_prefilter2d(const double *in, int ydim, int xdim, double *coef);
double xo_bspline3_interp2d(const double *coef, int ydim, int xdim, int starty, int startx,
                            double x, double y);

/* ---- Polar<> (POL) ------------------------------------------------------- */
int xo_polar_nsam(int radius);                       /* polar.h:723-726 */
void xo_polar_layout(int Ri, int Ro, int *nsam /*nrings*/, int *total_samples, int *total_coefs);
/* polar.h:625-703 + polar.cpp:57-83 (float angle cache) */
void xo_polar_from_cartesian_bspline(const double *coef, int ydim, int xdim, int starty,
                                     int startx, int Ri, int Ro, double xoff, double yoff,
                                     double *rings /* total_samples */);
void xo_polar_avg_std(const double *rings, int Ri, int Ro, double *avg, double *stddev); /* polar.h:488-534 */
void xo_polar_fft_rings(const double *rings, int Ri, int Ro, int conjugated,
                        double *coefs /* total_coefs*2 */);                /* polar.cpp:34-54 */
void xo_rotational_correlation(const double *F1, const double *F2, int Ri, int Ro,
                               double *corr /* nsam(Ro) */);               /* polar.cpp:99-148 */

/* ---- geometry / correlation (xmippCore; pins test_transformation_main.cpp:76-113,
 * test_filters_main.cpp:59-103, test_geometry_main.cpp:46-65) ------------- */
void xo_euler_matrix(double rot, double tilt, double psi, double *A /*9*/);
/* applyGeometry 2D; A is 3x3 row-major; degree 1 (LINEAR) or 3 (BSPLINE3) */
void xo_apply_geometry2d(int degree, const double *in, int ydim, int xdim, const double *A,
                         int is_inv, int wrap, double *out);
void xo_rotate2d(int degree, const double *in, int ydim, int xdim, double ang_deg, int wrap,
                 double *out);
void xo_translate2d(int degree, const double *in, int ydim, int xdim, double sx, double sy,
                    int wrap, double *out);
void xo_correlation_matrix(const double *m1, const double *m2, int ydim, int xdim, double *R);
/* FIL:1593-1719; returns max; maxShift=-1 => global max */
double xo_best_shift_mcorr(double *Mcorr, int ydim, int xdim, int maxShift, double *shiftX,
                           double *shiftY);
double xo_best_shift(const double *I1, const double *I2, int ydim, int xdim, int maxShift,
                     double *shiftX, double *shiftY);                       /* FIL:1721-1741 */
double xo_correlation_index(const double *x, const double *y, size_t n);

/* ---- projection matching (APM) -------------------------------------------
 * Reference library handle: holds, per reference, the conjugated polar FT
 * (mean-subtracted), its sigma and the image (APM:408-528). */
typedef struct xo_pm xo_pm;
xo_pm *xo_pm_create(int D, int Ri, int Ro, int nrefs, const double *refs /* nrefs*D*D */,
                    const double *Mctf /* NULL or paddim*(paddim) full FFT-order real */,
                    int paddim);
void xo_pm_destroy(xo_pm *);
int xo_pm_nsam_outer(const xo_pm *);
int xo_pm_ncoef(const xo_pm *);
const double *xo_pm_ref_coefs(const xo_pm *, int ref);  /* ncoef*2 */
double xo_pm_ref_sigma(const xo_pm *, int ref);
/* per-particle preparation (APM:569-597): out coefs straight + conj, sigma */
void xo_pm_prepare_particle(const xo_pm *, const double *img, double xoff, double yoff,
                            double *fP /*ncoef*2*/, double *fPm /*ncoef*2*/, double *sigma);
/* Rotational search (APM:530-773 + Appendix C of SURVEY.md), --thr 1.
 * nbr_off/nbr_ids: CSR neighbour lists (nbr_off NULL => every particle sees refs 0..nrefs-1).
 * first_image_parity: 0 => first particle loops forward (APM:318,1112).
 * Outputs are [n][n_orient]; invalid slots have refno=-1. psi_idx in [0,N). */
void xo_pm_match(const xo_pm *, const double *particles, int n, const int32_t *nbr_off,
                 const int32_t *nbr_ids, int first_image_parity, int n_orient,
                 const int32_t *xoff5d, const int32_t *yoff5d, int ntrans, int nthreads,
                 int32_t *refno, int32_t *psi_idx, uint8_t *flip, double *cc);
void xo_pm_match_thr(const xo_pm *, const double *particles, int n, const int32_t *nbr_off,
                 const int32_t *nbr_ids, int first_image_parity, int n_orient,
                 const int32_t *xoff5d, const int32_t *yoff5d, int ntrans, int nthreads, int ref_threads /* the program's --thr: APM:631,1063-1108 */,
                 int32_t *refno, int32_t *psi_idx, uint8_t *flip, double *cc);
/* full correlation rows for one (particle, ref): corr[2N] = straight || mirror, normalised */
void xo_pm_corr_rows(const xo_pm *, const double *img, int ref, double *corr2N);
/* Translational step (APM:776-868); max_shift<0 => dim/2 as in APM:262-263 */
void xo_pm_translate(const xo_pm *, const double *particles, int n, const int32_t *refno,
                     const int32_t *psi_idx, const uint8_t *flip, double max_shift, int nthreads,
                     double *shiftX, double *shiftY, double *maxCC);

/* ---- Kaiser-Bessel blobs (BLB:37-92,144-172) ----------------------------- */
double xo_kaiser_value(double r, double a, double alpha, int m);
double xo_kaiser_fourier_value(double w, double a, double alpha, int m);
double xo_bessi0(double x);
double xo_bessi1(double x);

/* ---- CTF (CTF: ctf.h:376-501,1002-1029; ctf.cpp:645-679,1392-1402) -------- */
typedef struct {
    double Tm, kV, DeltafU, DeltafV, azimuthal_angle, Cs, Ca, espr, ispr, alpha, DeltaF, DeltaR,
        Q0, K, envR0, envR1, envR2, phase_shift, VPP_radius;
} xo_ctf_params;
void xo_ctf_defaults(xo_ctf_params *p);
/* getValuePureNoKAt at continuous frequency (X,Y) [1/A] */
double xo_ctf_value_pure_nok(const xo_ctf_params *p, double X, double Y);
double xo_ctf_lambda(const xo_ctf_params *p);
/* actualPhaseFlip (ctf_phase_flip.cpp:88-117; with_damping = 1: CTFDescription::correctPhase, ctf.cpp:1553-1582), in place;
 * ctf->Tm = sampling rate of the image, ctf->phase_shift in degrees */
void xo_ctf_phase_flip(double *img, int ydim, int xdim, const xo_ctf_params *ctf, int with_damping);
/* Wiener2D::applyWienerFilter (data/wiener2d.cpp:29-141) on one image, in place */
void xo_ctf_wiener2d(double *img, int ydim, int xdim, const xo_ctf_params *ctf, double sampling_rate, double pad,
                     int phase_flipped, int is_isotropic, double wiener_constant, int correct_envelope);

/* ---- FlexAlign, global alignment (reconstruction/movie_alignment_correlation.cpp:45-157 + _base.cpp + eq_system_solver.cpp) -- */
int xo_fa_global_alignment(const double *frames, int N, int Y, int X, const double *dark, const double *igain, float Ts,
                           float maxShift, float maxRes, double *bX, double *bY, double *shiftX, double *shiftY, int *refFrame,
                           int *newDims);
void xo_fa_solve(const double *bX, const double *bY, int N, int iterations, double *shiftX, double *shiftY, int *refFrame);
/* local (patch) alignment as the CUDA program does it (movie_alignment_correlation_gpu.cpp:140-222,288-430), in double; see the
 * source for what is and what is not taken from the reference */
int xo_fa_local_alignment(const double *frames, int N, int Y, int X, const double *gShiftX, const double *gShiftY, int refFrame,
                          float Ts, float maxShift, float maxRes, int patchesX, int patchesY, int patchSizeX, int patchSizeY,
                          int patchesAvg, int lX, int lY, int lT, double *patchShifts, double *centers, double *coeffsX,
                          double *coeffsY, int *dims);
/* --bin of the CUDA FlexAlign program: Fourier cropping of a frame (cuda_flexalign_scale.cpp:101-121, cuda_scaleFFT_kernels.cu:44-79) */
void xo_fa_bin_frame(const double *frame, int Y, int X, int Yb, int Xb, double *out);
/* the same on float frames for the patches patchMask [py][px] marks (full-size tests): centres of all, shifts of the marked ones, no fit */
int xo_fa_local_patch_shifts_f32(const float *frames, int N, int Y, int X, const double *gShiftX, const double *gShiftY, int refFrame,
                                 float Ts, float maxShift, float maxRes, int patchesX, int patchesY, int patchSizeX, int patchSizeY,
                                 int patchesAvg, const uint8_t *patchMask, double *patchShifts, double *centers, int *dims);
void xo_fa_bspline_shift(const double *coeffsX, const double *coeffsY, int lX, int lY, int lT, int X, int Y, int N, int x, int y, int n,
                         double *shiftX, double *shiftY);
void xo_fa_apply_bspline(const double *frame, int Y, int X, const double *coeffsX, const double *coeffsY, int lX, int lY, int lT, int N, int n,
                         double *out);
double xo_dose_voltage_scaling(double accelerationVoltage);
double xo_dose_filter(double dose_at_end_of_frame, double critical_dose);
double xo_dose_critical(double spatial_frequency, double voltage_scaling_factor);
double xo_dose_optimal(double critical_dose);
void xo_dose_filter_frame(double *frame, int Y, int X, double pixel_size, double voltage_scaling_factor, double dose_start, double dose_finish);
void xo_fa_correlate(const double *frames, int N, int Y, int X, double maxDist, double *pos);

/* ---- Fourier reconstruction (RFA) ---------------------------------------- */
typedef struct {
    int imgSize;              /* D */
    double padding_proj, padding_vol;
    double maxResolution;     /* digital freq, default 0.5 */
    double blob_radius; int blob_order; double blob_alpha;
    int useFast;
    int useCTF; int isPhaseFlipped; double minCTF; double iTs;
    /* derived by xo_rf_setup (RFA:175-257) */
    int paddedImgSize; int maxVolumeIndexX, maxVolumeIndexYZ;
    float iDeltaSqrt, iDeltaFourier;
} xo_rf_params;
typedef struct xo_rf xo_rf;
xo_rf *xo_rf_create(xo_rf_params *p /* in/out */);
void xo_rf_destroy(xo_rf *);
const float *xo_rf_blob_table_sqrt(const xo_rf *);      /* 10000 floats */
const double *xo_rf_fourier_blob_table(const xo_rf *);  /* 10000 doubles */
/* preloadBuffer+cropAndShift (RFA:300-388,271-298): img D*D (shifts already applied) ->
 * half spectrum [mv rows][mv/2 cols] complex<float> interleaved */
void xo_rf_prepare_image(const xo_rf *, const double *img, float *fft_out);
/* preloadCTF (RFA:548-592): arrays [mv rows][mv/2 cols] */
void xo_rf_ctf_arrays(const xo_rf *, const xo_ctf_params *ctf, float *CTF, float *modulator);
/* processBuffer body for one projection & one symmetry matrix R (3x3, row-major double):
 * RFA:939-966 + processProjection RFA:710-763. ctf/modulator may be NULL. */
void xo_rf_insert(xo_rf *, const float *fft, const float *ctf, const float *modulator,
                  const double *localAInv /*9 = Euler^T*/, const double *R /*9*/, float weight);
/* direct access to temp spaces (mv+1)^3 (before mirror) */
float *xo_rf_temp_volume(xo_rf *);  /* complex interleaved */
float *xo_rf_temp_weights(xo_rf *);
void xo_rf_reset(xo_rf *);
/* mirrorAndCropTempSpaces (RFA:853-887) */
void xo_rf_mirror_and_crop(xo_rf *);
/* after mirror: (mv+1)*(mv+1)*(mv/2+1) */
/* finishComputations (RFA:1002-1055) -> volume D^3 doubles */
void xo_rf_finish(xo_rf *, double *vol_out);
/* helper exposing only forceHermitianSymmetry+processWeights on the cropped spaces */
void xo_rf_hermitian_and_weights(xo_rf *);

/* ---- ProgRecFourier, the double-precision scatter variant (RF: reconstruction/reconstruct_fourier.cpp), xo_recfourier2.cpp */
typedef struct xo_rf2 xo_rf2;
xo_rf2 *xo_rf2_create(int D, double pad_proj, double pad_vol, double max_resolution, double blob_radius, int blob_order,
                      double blob_alpha, int niter_weight);
void xo_rf2_destroy(xo_rf2 *);
double *xo_rf2_weights(xo_rf2 *);      /* [V][V][V/2+1] */
double *xo_rf2_fourier(xo_rf2 *);      /* the same, complex interleaved */
int xo_rf2_vol_pad(const xo_rf2 *);
void xo_rf2_insert(xo_rf2 *, const double *img, const double *localAInv, const double *Rsym, double weight,
                   const xo_ctf_params *ctf, double iTs, double minCTF, int phaseFlipped, int reprocess);
void xo_rf2_weights_begin(xo_rf2 *);
void xo_rf2_weights_iter_begin(xo_rf2 *);
void xo_rf2_weights_iter_end(xo_rf2 *);
void xo_rf2_weights_end(xo_rf2 *);
void xo_rf2_finish(xo_rf2 *, double *vol);

int xo_num_threads(void);
/* ---- FourierProjector (data/fourier_projection.cpp:91-330): central-slice projection, the producer of
 * the reference gallery (angular_project_library --method fourier pad maxfreq interp) ---- */
typedef struct xo_fp xo_fp;
xo_fp *xo_fp_create(const double *vol /* D^3, [z][y][x] */, int D, double padding, double max_freq,
                    int degree /* 0 nearest, 1 linear, 3 cubic B-spline */);
void xo_fp_destroy(xo_fp *);
int xo_fp_padded_size(const xo_fp *);
int xo_fp_coef_dim(const xo_fp *);
int xo_fp_coef_start(const xo_fp *);
const double *xo_fp_coefs(const xo_fp *, int imag);
void xo_fp_project(const xo_fp *, double rot, double tilt, double psi, const double *ctf /* nullable */,
                   double *out /* D*D */);

/* ---- Fourier shell correlation (xo_frc.cpp): frc_dpr of xmippCore as called by
 * reconstruction/resolution_fsc.cpp:179-203. Arrays of X/2+1 doubles; returns that length. */
int xo_frc_dpr(const double *m1, const double *m2, int Z, int Y, int X, double sampling_rate, int dodpr,
               int dorfactor, double minFreq, double maxFreq, double *freq, double *frc, double *frc_noise,
               double *dpr /* nullable */, double *error_l2, double *rFactor /* nullable */);

#ifdef __cplusplus
}
#endif
#endif
