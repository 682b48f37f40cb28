/*
 * xmipp_hip.h -- C ABI of libxmipp_hip.so: the MI355X (gfx950) device side of
 * Xmipp's projection-matching + Fourier-gridding hot path.
 *
 * This is the drop-in boundary.  Host programs (our C++ mirrors of
 * ProgAngularProjectionMatching / ProgRecFourierAccel in xmipp3_amd/host, the
 * Python harness in xmipp3_amd/, or upstream Xmipp itself -- see
 * INTEGRATION.md) call only these entry points.  Plain pointers and sizes, no
 * C++/torch types.  Reference paths below are relative to
 * /root/reference/src/xmipp/libraries.
 *
 * Conventions
 *  - every function returns 0 on success, a negative xh_status otherwise;
 *    xh_last_error() gives the message (thread-local).  There is NO CPU
 *    fallback: without a usable HIP device xh_ctx_create fails.
 *  - pointers named d_* are device pointers (HBM) valid on the context's
 *    device, h_* are host pointers.  Device buffers may be owned by the caller
 *    (e.g. a torch tensor) -- the library never frees caller memory.
 *  - all work is enqueued on the context's HIP stream; calls return when the
 *    work is enqueued unless stated otherwise ("synchronous").
 *  - one host thread at a time per context; handles are not thread-safe (same
 *    contract as reconstruction_cuda/cuda_gpu_reconstruct_fourier.h:46-157, one
 *    stream per host thread).  Every entry point makes its context's device
 *    current on the calling thread, so a process may drive several devices
 *    from several threads (xmipp3_amd/host: --gpus).
 *  - images are row-major float32, D x D, logical (Xmipp) origin at pixel
 *    (D/2, D/2) as after MultidimArray::setXmippOrigin().
 */
#ifndef XMIPP_HIP_H
#define XMIPP_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    XH_OK = 0,
    XH_ERR_ARG = -1,     /* invalid argument / unsupported size */
    XH_ERR_HIP = -2,     /* HIP runtime error */
    XH_ERR_NOMEM = -3,
    XH_ERR_STATE = -4,   /* call made in the wrong state */
    XH_ERR_UNSUPPORTED = -5
} xh_status;

const char *xh_last_error(void);
const char *xh_version(void);

/* ------------------------------------------------------------------ context
 * Replaces createStreams/deleteStreams/waitForGPU
 * (reconstruction_cuda/cuda_gpu_reconstruct_fourier.h:46-60, :128) and the
 * GPU HW object (reconstruction_cuda/gpu.h:36-). */
typedef struct xh_ctx xh_ctx;
/* stream: the hipStream_t all work of this context is enqueued on (e.g. torch's current
 * stream); NULL means the device's default (null) stream. The stream stays owned by the caller. */
int xh_ctx_create(int device, void *stream, xh_ctx **out);
/* same, but the context creates (and destroys) a private non-blocking stream */
int xh_ctx_create_private(int device, xh_ctx **out);
int xh_ctx_destroy(xh_ctx *ctx);
int xh_ctx_sync(xh_ctx *ctx);                 /* synchronous: waits for the stream */
void *xh_ctx_stream(xh_ctx *ctx);
int xh_device_count(int *count);
/* NUMA node of the host the device hangs off (/sys/bus/pci/devices/<bus id>/numa_node), -1 when the platform does not say.
 * A host that feeds the device from threads on the other socket loses more than half of the H2D rate (measured: 21 instead of
 * 56 GB/s with 16 reader threads filling page-locked memory beside the copy), so the programs bind their loaders to it. */
int xh_device_numa_node(int device, int *node);
/* device memory helpers for hosts that do not bring their own allocator
 * (allocateTempVolumeGPU/releaseTempVolumeGPU, cuda_gpu_reconstruct_fourier.h:83-91) */
int xh_malloc(xh_ctx *ctx, size_t bytes, void **d_ptr);
int xh_free(xh_ctx *ctx, void *d_ptr);
int xh_memset(xh_ctx *ctx, void *d_ptr, int value, size_t bytes);
int xh_memcpy_h2d(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes); /* synchronous */
int xh_memcpy_d2h(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* synchronous */
/* Page-locked host memory and copies that only enqueue: what a loader thread needs to feed the device while it computes
 * (pinMemory / unpinMemory, cuda_gpu_reconstruct_fourier.h:134-136, gpu.h:78-113; the loader of
 * reconstruct_fourier_accel.cpp:300-388).  The async copies return when enqueued on the context's stream; the host
 * buffer must stay untouched until xh_ctx_sync (or a later synchronous call) on that context.  With pageable host
 * memory they are still correct, only not asynchronous. */
int xh_host_alloc(xh_ctx *ctx, size_t bytes, void **h_ptr);
int xh_host_free(xh_ctx *ctx, void *h_ptr);
int xh_memcpy_h2d_async(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int xh_memcpy_d2h_async(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* everything enqueued so far on `other`'s stream happens before what is enqueued on ctx's stream from now on
 * (an event recorded on one stream and waited for on the other; no host wait).  Both contexts on one device. */
int xh_ctx_wait_for(xh_ctx *ctx, xh_ctx *other);
/* elapsed-time probes on the context stream (HIP events), for bench.py's roofline object */
int xh_timer_create(xh_ctx *ctx, void **timer);
int xh_timer_start(xh_ctx *ctx, void *timer);
int xh_timer_stop(xh_ctx *ctx, void *timer);
int xh_timer_elapsed_ms(xh_ctx *ctx, void *timer, float *ms); /* synchronous */
int xh_timer_destroy(xh_ctx *ctx, void *timer);

/* ------------------------------------------------------- Fourier gridding
 * Replaces ProgRecFourierAccel's inner loops (reconstruction/
 * reconstruct_fourier_accel.cpp: preloadBuffer :300-388, cropAndShift :271-298,
 * preloadCTF :548-592, processProjection :710-763, processVoxelBlob :627-700,
 * processVoxel :595-625, mirrorAndCrop :853-887, finishComputations :1002-1055)
 * and the CUDA seam processBufferGPU/copyTempVolumes/copyBlobTable/copyConstants
 * (reconstruction_cuda/cuda_gpu_reconstruct_fourier.h:93-157). */
typedef struct {
    int32_t imgSize;            /* D; only square images (RFA:192-193) */
    double padding_proj;        /* --padding <proj> (RFA:65,91) */
    double padding_vol;         /* --padding <vol> */
    double max_resolution;      /* --max_resolution, digital frequency (RFA:66) */
    double blob_radius;         /* --blob radius order alpha (RFA:68) */
    int32_t blob_order;
    double blob_alpha;
    int32_t use_fast;           /* --fast */
    int32_t phase_flipped;      /* --phaseFlipped */
    double min_ctf;             /* --minCTF */
    double sampling;            /* --sampling Ts (A/px); iTs = 1/Ts */
} xh_rf_params;

/* CTFDescription fields used by getValuePureNoKAt (data/ctf.h:452-502,
 * data/ctf.cpp:645-679,1392-1402); defaults via xh_ctf_defaults
 * (data/ctf.cpp:365-388). */
typedef struct {
    double Tm, kV, DeltafU, DeltafV, azimuthal_angle, Cs, Ca, espr, ispr, alpha, DeltaF, DeltaR,
        Q0, K, envR0, envR1, envR2, phase_shift, VPP_radius;
} xh_ctf_params;
void xh_ctf_defaults(xh_ctf_params *p);

/* One projection x symmetry placement ("traverse space",
 * reconstruction/reconstruct_fourier_projection_traverse_space.h:37-59). The
 * library fills these itself from Euler angles; exposed for hosts that
 * already hold matrices. */
typedef struct xh_rf xh_rf;

int xh_rf_create(xh_ctx *ctx, const xh_rf_params *p, xh_rf **out);
int xh_rf_destroy(xh_rf *rf);
/* profiling / A-B knobs of the reconstruction handle (defaults are the product path): "unit_z" 4 | 8 (depth of a wave's voxel unit in
 * k_rf_grid), "grid_waves" 0 (= the configuration's default) | 8 | 9 | 12 | 16, "fuse_ctf" 1 | 0 (CTF evaluated while the records
 * are packed, or through planes), "ctf_fast" 1 | 0 (CTFs without envelope terms evaluated by the cheap form of preloadCTF's value, or the
 * general double-precision formula for every pixel), "records_from_images" 0 | 1 (records written by the FFT's row pass), "order_spaces" 1 | 0 (the traverse spaces of a launch
 * ordered by plane -- the gridding kernel then shares the voxel queue between projections of one direction -- or in input order; the order
 * permutes the launch's float additions), "shift_bands" 1 | 0 (256-px images shifted band by band out of LDS, or by k_rf_shift: same bits), "tile_max_spaces",
 * "fft_variant" 1 | 2 (columns-first / rows-first projection FFT). Unknown names fail with XH_ERR_ARG. */
int xh_rf_set_option(xh_rf *rf, const char *name, double value);
/* derived sizes (RFA:196-199): paddedImgSize P, maxVolumeIndexYZ mv, fft crop sizeX=mv/2, sizeY=mv */
int xh_rf_sizes(const xh_rf *rf, int32_t *paddedImgSize, int32_t *maxVolumeIndex,
                int32_t *fftSizeX, int32_t *fftSizeY);
/* host copies of the lookup tables (RFA:201-239): blobTableSqrt[10000] float,
 * Fourier_blob_table[10000] double, iDeltaSqrt, iDeltaFourier */
int xh_rf_tables(const xh_rf *rf, float *h_blobTableSqrt, double *h_fourierBlobTable,
                 float *iDeltaSqrt, float *iDeltaFourier);
/* Temp spaces: one contiguous float buffer [ volume (mv+1)^3 complex | weights (mv+1)^3 ],
 * i.e. 3*(mv+1)^3 floats, layout [z][y][x]. By default owned by the handle; a caller-owned
 * buffer (e.g. a torch tensor, so torch.distributed can all-reduce it) can be attached
 * instead. xh_rf_reset zeroes it. */
size_t xh_rf_temp_floats(const xh_rf *rf);
int xh_rf_attach_temp(xh_rf *rf, float *d_temp /* xh_rf_temp_floats() floats */);
int xh_rf_temp_ptr(xh_rf *rf, float **d_temp);
int xh_rf_reset(xh_rf *rf);
/* Image::readApplyGeo(..., only_apply_shifts) (RFA:304-323): out = in translated by
 * h_shiftXY[i] = (shiftX, shiftY) pixels (and mirrored in x where h_flip[i] != 0; h_flip may be
 * NULL), cubic B-spline interpolation with wrapping. */
int xh_rf_shift_images(xh_rf *rf, const float *d_imgs, const float *h_shiftXY /* [n][2] */,
                       const uint8_t *h_flip /* [n] or NULL */, int32_t n, float *d_out);
/* The same from cubic B-spline coefficients the caller already holds (produceSplineCoefficients of
 * d_imgs, [n][D][D] float): in one refinement iteration the matcher has just computed them for the same
 * particles (xh_pm_last_coefficients), the reference's two programs each compute their own
 * (APM:569, RFA:304-323 through readApplyGeo). */
int xh_rf_shift_images_coefs(xh_rf *rf, const float *d_imgs, const float *d_coefs, const float *h_shiftXY,
                             const uint8_t *h_flip, int32_t n, float *d_out);
/* The same with the shifts and flips where xh_pm_translate / xh_pm_match left them (device memory: d_shiftX, d_shiftY
 * [n] doubles, d_flip [n] bytes or NULL; d_coefs may be NULL): no host copy of the matcher's outputs is needed
 * between the two programs (the reference passes them through a metadata file, APM:820-880 -> RFA:296-323). */
int xh_rf_shift_images_dev(xh_rf *rf, const float *d_imgs, const float *d_coefs, const double *d_shiftX,
                           const double *d_shiftY, const uint8_t *d_flip, int32_t n, float *d_out);
/* preloadBuffer + cropAndShift for n images already shifted (shifts applied):
 * d_imgs [n][D][D] float  ->  d_fft [n][mv][mv/2] complex<float> (interleaved) */
int xh_rf_prepare_images(xh_rf *rf, const float *d_imgs, int32_t n, float *d_fft);
/* preloadCTF for n images: h_ctf [n] params -> d_ctf, d_mod [n][mv][mv/2] float */
int xh_rf_ctf_arrays(xh_rf *rf, const xh_ctf_params *h_ctf, int32_t n, float *d_ctf, float *d_mod);
/* processBuffer: insert n prepared projections. h_angles [n][3] = (rot,tilt,psi) degrees;
 * h_weights [n] or NULL (=1); h_sym [nsym][9] row-major symmetry matrices R (NULL => identity,
 * nsym=1; RFA:241-256); d_ctf/d_mod NULL => no CTF. */
int xh_rf_insert(xh_rf *rf, const float *d_fft, const float *d_ctf, const float *d_mod,
                 const double *h_angles, const float *h_weights, int32_t n, const double *h_sym,
                 int32_t nsym);
/* processBufferGPU in one call (the reference's device variant does the FFT on the device as well,
 * reconstruction_cuda/cuda_gpu_reconstruct_fourier.h:130-157, RFG:417-473): images with the shifts already
 * applied [n][D][D] float + CTF parameters (h_ctf [n], or NULL) + orientations (h_angles [n][3] rot, tilt, psi
 * in degrees) + weights (or NULL) + symmetry matrices -> temp spaces. Same result as xh_rf_ctf_arrays +
 * xh_rf_prepare_images + xh_rf_insert on scratch owned by the handle. */
int xh_rf_insert_images(xh_rf *rf, const float *d_imgs, const xh_ctf_params *h_ctf, const double *h_angles,
                        const float *h_weights, int32_t n, const double *h_sym, int32_t nsym);
/* xh_rf_insert_images with the orientations where the matcher left them: d_angles [n][3] doubles (rot, tilt, psi in
 * degrees) and d_weights [n] floats (or NULL) in device memory. The traverse spaces (RFA:939-966, 430-522) are built
 * by a kernel from the functions the host path uses (double arithmetic on both sides, the device's fused multiply-adds
 * may round the last bit of a matrix element differently), so the two forms give the same temp spaces to float rounding
 * (test_device_side_orientations_give_the_host_forms_temp_spaces: 1e-6); a weight of 0 drops the projection (RFA:327-329). The call enqueues and returns: it never waits for the stream. */
int xh_rf_insert_images_dev(xh_rf *rf, const float *d_imgs, const xh_ctf_params *h_ctf, const double *d_angles,
                            const float *d_weights, int32_t n, const double *h_sym, int32_t nsym);
/* same but taking the 3x3 "localAInv" (= Euler^T) matrices directly, h_ainv [n][9] */
int xh_rf_insert_matrices(xh_rf *rf, const float *d_fft, const float *d_ctf, const float *d_mod,
                          const double *h_ainv, const float *h_weights, int32_t n,
                          const double *h_sym, int32_t nsym);
/* HIP-event time (ms, on the context's stream) and number of launches of the output-stationary
 * gridding kernel since the last reset -- the live per-launch duration behind bench.py's roofline.
 * No reference counterpart (RFG times nothing). */
int xh_rf_kernel_ms(xh_rf *rf, double *h_ms, int64_t *h_launches, int32_t reset);
/* mirrorAndCropTempSpaces (RFA:853-887): temp -> cropped [ (mv+1)^2*(mv/2+1) complex | weights ]
 * stored at the start of the same buffer; 3*(mv+1)^2*(mv/2+1) floats are what a multi-GPU
 * host all-reduces (SUM) before xh_rf_finish. */
int xh_rf_mirror_and_crop(xh_rf *rf);
size_t xh_rf_cropped_floats(const xh_rf *rf);
/* The cropped spaces as data (xh_rf_cropped_floats() floats on the handle's device):
 *  - export: copy them out (asynchronous on the handle's stream);
 *  - import: load (add = 0; the handle then counts as cropped) or accumulate (add != 0) a buffer of
 *    the same layout that lives on the handle's device.
 * This is what ProgRecFourier's --prepare_fsc does through its <fsc>_1_Fourier.vol / _1_Weights.vol
 * files (reconstruction/reconstruct_fourier.cpp:991-1045): keep half 1, rebuild from zero for half 2,
 * then sum both. xh_rf_finish consumes the cropped spaces (weights are divided in place), so export
 * before finishing whatever has to be summed later. */
int xh_rf_cropped_export(xh_rf *rf, float *d_dst);
int xh_rf_cropped_import(xh_rf *rf, const float *d_src, int32_t add);
/* Sum the cropped spaces of n handles (any mix of devices of this node) into rfs[0]; synchronous. Handles
 * that share a device are added there first; the first handle of every device then takes part in ONE
 * in-place RCCL all-reduce (ncclCommInitAll over the devices, grouped ncclAllReduce, float sum) over all
 * xGMI links, so each of those ends up with the total. RCCL is bound at run time; if it cannot be loaded
 * or initialised the call says so on stderr and falls back to a binary tree of peer copies + adds. Replaces
 * the 2*(mv+1)^2 MPI_Reduce calls of mpi_reconstruct_fourier_accel.cpp:245-266 (GPU variant:
 * parallel_adapt_cuda/mpi_reconstruct_fourier_gpu.cpp:250-268) for a single-process, thread-per-device
 * host (the torch.distributed host all-reduces the same buffer instead, see xh_rf_attach_temp). */
int xh_rf_reduce(xh_rf *const *rfs, int32_t n);
/* finishComputations (RFA:1002-1055): synchronous; writes D^3 doubles, [z][y][x] */
int xh_rf_finish(xh_rf *rf, double *h_volume);

/* ---------------------------------------------------- projection matching
 * Replaces ProgAngularProjectionMatching's inner loops (reconstruction/
 * angular_projection_matching.cpp: getCurrentReference :408-528,
 * threadRotationallyAlignOneImage :530-773, translationallyAlignOneImage :776-868)
 * and the primitives under them (data/polar.h:488-534,625-738, data/polar.cpp:34-148,
 * data/filters.cpp:1593-1752). */
typedef struct xh_pm xh_pm;
/* Builds the reference library on the device: d_refs [nrefs][D][D] float. Ri<1 -> 1,
 * Ro<0 -> D/2-1 (APM:266-274). h_Mctf: optional real filter [paddim][paddim] in FFT order
 * applied to every reference (APM:457-481), NULL for none. */
int xh_pm_create(xh_ctx *ctx, int32_t D, int32_t Ri, int32_t Ro, int32_t nrefs,
                 const float *d_refs, const double *h_Mctf, int32_t paddim, xh_pm **out);
int xh_pm_destroy(xh_pm *pm);
int xh_pm_info(const xh_pm *pm, int32_t *nsam_outer /*N*/, int32_t *ncoef, int32_t *nsamples);
/* Rotational search for n particles (d_particles [n][D][D] float), --thr 1 semantics.
 * Neighbour lists in CSR form on the HOST: h_nbr_off [n+1], h_nbr_ids; both NULL => every
 * particle is compared with references 0..nrefs-1 (dense mode).
 * first_image_parity: 0 if the first particle of this call is an even-numbered image of the
 * run (the visiting order of references flips per image, APM:615-626,1112).
 * Outputs (device, one per particle): refno (-1 if the list is empty), psi_idx in [0,N)
 * (psi = psi_idx*360/N, polar.cpp:145-147), flip.
 * The orientation indices are exact w.r.t. a double-precision evaluation: an fp32 coarse
 * pass ranks all (ref, angle, mirror) candidates and every particle whose runner-up lies
 * within the fp32 error margin of the winner is re-scored in fp64. */
int xh_pm_match(xh_pm *pm, const float *d_particles, int32_t n, const int32_t *h_nbr_off,
                const int32_t *h_nbr_ids, int32_t first_image_parity, int32_t *d_refno,
                int32_t *d_psi_idx, uint8_t *d_flip);
/* The general search of threadRotationallyAlignOneImage (APM:530-760), replacing its
 * `--search5d_shift/--search5d_step` and `--number_orientations` loops:
 *  ntrans > 0: 5-D search -- every particle is also resampled about (xoff5d[t], yoff5d[t])
 *    (the list built at APM:334-348) and every (reference, translation) pair is a row of the
 *    search; as in the reference only (refno, psi, flip) of the winner are kept (the shift is
 *    re-estimated by xh_pm_translate). ntrans == 0 is the single translation (0,0).
 *  n_orient > 1: the reference's running top-N (APM:714-735) evaluated exactly (all rows in
 *    fp64); outputs are [n][n_orient], rank-major per particle, refno = -1 for ranks that were
 *    never filled (counterValidCorrs, APM:1068-1090). n_orient <= 16.
 * xh_pm_match(...) == xh_pm_match_ex(..., 1, 0, NULL, NULL, ...). */
int xh_pm_match_ex(xh_pm *pm, const float *d_particles, int32_t n, const int32_t *h_nbr_off,
                   const int32_t *h_nbr_ids, int32_t first_image_parity, int32_t n_orient,
                   int32_t ntrans, const int32_t *h_xoff5d, const int32_t *h_yoff5d,
                   int32_t *d_refno, int32_t *d_psi_idx, uint8_t *d_flip);
/* Translational step for the winners (APM:776-868): bestShift on correlation_matrix(rotated
 * reference, (mirrored) particle), rejection beyond max_shift (<0 => D/2), translate+
 * correlationIndex => maxCC. fp64 on the device. Outputs device double [n]. */
int xh_pm_translate(xh_pm *pm, const float *d_particles, int32_t n, const int32_t *d_refno,
                    const int32_t *d_psi_idx, const uint8_t *d_flip, double max_shift,
                    double *d_shiftX, double *d_shiftY, double *d_maxCC);
/* fp32 cubic B-spline coefficients (produceSplineCoefficients, APM:569) of the particles of the last
 * xh_pm_match[_ex] call, [count][D][D] float on the device, valid until the next call on this handle;
 * first = index of the first particle they belong to (0 and count = n when the call ran in one chunk). */
int xh_pm_last_coefficients(const xh_pm *pm, const float **d_coefs, int32_t *first, int32_t *count);
/* particles the last xh_pm_translate repeated in double precision: its first pass is fp32 and a particle whose arg-max
 * or window decision (FIL:1659-1689) comes within a margin of flipping is done again in the reference's arithmetic
 * (xh_pm_set_option "s6_fp32" 0: everything in double; "s6_eps": the margin, relative to the map's maximum -- default 6.4e-6,
 * twenty times the measured error of the fp32 map) */
int xh_pm_translate_stats(const xh_pm *pm, int64_t *repeated);
/* statistics of the last xh_pm_match call: rows evaluated, particles re-scored in fp64 */
int xh_pm_last_stats(const xh_pm *pm, int64_t *rows, int64_t *rescored_particles,
                     int64_t *rescored_rows);
/* accumulated device time (HIP events) per stage of xh_pm_match since the last reset:
 * h_ms[8] = { prep32, contract, idft_max, select, rescore_fp64, 0, 0, 0 } milliseconds */
int xh_pm_stage_ms(xh_pm *pm, double *h_ms, int32_t reset);
/* rows of the last xh_pm_match[_ex] call that the S3 branch and bound skipped: a correlation row whose
 * coefficient moduli sum to less than (best value found for the particle - 2 tau) is never transformed.
 * Same results with set_option("prune", 0); no reference counterpart (the reference transforms every row,
 * polar.cpp:136-146). */
int xh_pm_rows_pruned(const xh_pm *pm, int64_t *rows_pruned);
/* Two-level contraction: the angular frequency K0 at which the MFMA contraction of the dense search stops
 * (the reference bank holds < 1e-5 of its summed coefficient norms above it); rows that survive the branch and
 * bound get their frequencies K0..nk-1 afterwards, a particle's rows together (set_option "group_high" 0: each by the wave
 * that transforms it), all others are covered by a Cauchy-Schwarz term in their bound. K0 == nk: the bank is not band
 * limited, everything is contracted. set_option("k0", v) overrides (0 = automatic). Identical results for every K0.
 * A gallery with flat correlation peaks leaves a large part of the rows standing: when more than 10 % of a chunk's rows
 * survive, the next chunk (of this or of the next call) is contracted at every frequency with its coefficients kept, which
 * is cheaper from 9 % on; below 6 % it goes back (set_option "adaptive_finish" 0: never; get_option "dense_chunks": how
 * many chunks of the last call took that form). Identical results either way. */
int xh_pm_two_level_cut(const xh_pm *pm, int32_t *K0, int32_t *nk);
/* tuning knobs: fp32 ambiguity margin relative to sum_r 2*pi*r; rows per launch chunk.
 * "threads" n (1..16): the reference program's --thr (angular_projection_matching.cpp:64,631,1018-1108).  Its n worker threads take
 * the positions i % n of a particle's neighbour list and their results are merged, worker 0 first, strictly greater wins -- which
 * only shows where two correlation values are EXACTLY equal (duplicated references): the winner is then the one with the smallest
 * (i % n, position in the image's visiting order) instead of the first visited.  xh_pm_match[_ex] reproduces that order, for the
 * running top-N (n_orient > 1) as well. */
int xh_pm_set_option(xh_pm *pm, const char *name, double value);
/* current value of "tau_rel" (ambiguity margin of the fp32 coarse search, relative to sum_r 2 pi r) or "s6_eps" (margin of the
 * fp32 pass of xh_pm_translate, relative to the maximum of the correlation map): the tests hold the measured fp32 errors against them;
 * "adaptive_finish", "dense_chunks": see xh_pm_two_level_cut */
int xh_pm_get_option(const xh_pm *pm, const char *name, double *value);

/* ---- FourierProjector: central-slice projections of a volume ------------------
 * Replaces the class FourierProjector (libraries/data/fourier_projection.h:111-172) behind
 * xmipp_angular_project_library --method fourier <pad> <maxfreq> bspline
 * (reconstruction/angular_project_library.cpp:194-245): the gallery of references that
 * xh_pm_create consumes, so that volume -> references -> match -> reconstruct stays on the device.
 *   create  == FourierProjector(V, paddFactor, maxFreq, BSPLINE3) / updateVolume
 *              (fourier_projection.cpp:72-89,247-330); d_vol: D^3 floats, [z][y][x], Xmipp origin.
 *   project == project(rot, tilt, psi, ctf) + projection() for n orientations at once
 *              (fourier_projection.cpp:91-245); h_angles: n x (rot, tilt, psi) degrees; d_ctf: optional
 *              [D][D/2+1] doubles multiplied onto the slice; d_out: n x D x D floats.
 * Only cubic B-spline interpolation (the program's default) is implemented; degree 0/1 fail loudly. */
typedef struct xh_fp xh_fp;
int xh_fp_create(xh_ctx *ctx, const float *d_vol, int32_t D, double padding, double max_freq,
                 int32_t degree, xh_fp **out);
int xh_fp_destroy(xh_fp *fp);
int xh_fp_info(const xh_fp *fp, int32_t *padded_size, int32_t *coef_dim, int32_t *coef_start);
/* test hook: the cropped B-spline coefficient cubes (VfourierRealCoefs / VfourierImagCoefs), host doubles */
int xh_fp_coefs(const xh_fp *fp, double *h_re, double *h_im);
int xh_fp_project(xh_fp *fp, const double *h_angles, int32_t n, const double *d_ctf, float *d_out);

/* ------------------------------------------------- Fourier shell correlation
 * Replaces the frc_dpr(refI(), img(), sam, freq, frc, frc_noise, dpr, error_l2, do_dpr, do_rfactor,
 * min_samp, sam/max_sam, &rFactor) call of ProgResolutionFsc::process_img
 * (reconstruction/resolution_fsc.cpp:179-203; frc_dpr itself is xmippCore's xmipp_fftw.cpp).
 * d_m1 (the reference map) and d_m2: [Z][Y][X] doubles on the device (Z = 1 for images), sizes <= 1024.
 * Outputs on the host, X/2+1 doubles each (h_dpr / h_rfactor only touched when requested): digital
 * frequency / sampling_rate, FRC, 2/sqrt(shell count), differential phase residual (degrees), mean
 * |F1 - F2|; R-factor over minFreq <= R <= maxFreq (digital frequencies). Synchronous. */
int xh_frc_dpr(xh_ctx *ctx, const double *d_m1, const double *d_m2, int32_t Z, int32_t Y, int32_t X,
               double sampling_rate, int32_t do_dpr, int32_t do_rfactor, double minFreq, double maxFreq,
               double *h_freq, double *h_frc, double *h_frc_noise, double *h_dpr, double *h_error_l2,
               double *h_rfactor);

/* ---- test hooks (used only by tests/ to localise a parity failure per stage) ---- */
/* polar Fourier transform of one stage: precision 32 or 64; outputs on host */
int xh_pm_debug_prepare(xh_pm *pm, const float *d_particles, int32_t n, int32_t precision,
                        double *h_coefs /* [n][ncoef][2] */, double *h_sigma /* [n] */);
int xh_pm_debug_ref(xh_pm *pm, int32_t ref, double *h_coefs /* [ncoef][2] conj'd */,
                    double *h_sigma);
/* full normalised correlation rows (straight || mirror, 2N) of particle p vs reference r as
 * computed by the fp32 coarse pass (precision 32) or the fp64 re-scorer (64) */
int xh_pm_debug_corr_rows(xh_pm *pm, const float *d_particle, int32_t ref, int32_t precision,
                          double *h_corr2N);
/* the correlation maps (correlation_matrix, FIL:1636) the last chunk of an xh_pm_translate call under
 * set_option("s6_capture", 32 | 64) left behind, [n][D][D] doubles on the host; 64, 128 and 256 px */
int xh_pm_debug_s6_maps(xh_pm *pm, int32_t n, double *h_maps);

/* ---- 2-D complex FFTs of whole movie frames (FlexAlign) --------------------------------
 * reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp:633-725 plans cuFFT transforms of 4096 x 5760 (K3) frames;
 * a line of that length does not fit one LDS transform, so it is done in four steps (two strided passes of the line
 * kernels, a twiddle pass, an untangling transpose). ny x nx complex<float>, row-major, interleaved, in place;
 * forward un-normalised, inverse divided by ny nx (FFTW / cuFFT conventions of the reference's FourierTransformer:
 * forward/normalised-inverse pair). Any ny, nx whose prime factors allow a split into two lines of at most 2048 (powers
 * of two) / 1024 (other lengths, Bluestein) points: up to about a million points per axis. */
typedef struct xh_fft2d xh_fft2d;
int xh_fft2d_create(xh_ctx *ctx, int32_t ny, int32_t nx, xh_fft2d **out);
int xh_fft2d_destroy(xh_fft2d *f);
/* h_factors[4] = (ny1, ny2, nx1, nx2): the split of either axis (n2 = 1: one LDS transform per line) */
int xh_fft2d_factors(const xh_fft2d *f, int32_t *h_factors);
int xh_fft2d_exec(xh_fft2d *f, float *d_data /* [ny][nx][2] */, int32_t inverse);
int xh_fft2d_exec_axis(xh_fft2d *f, float *d_data, int32_t inverse, int32_t axis /* 0: rows only, 1: columns only; un-normalised */);

/* ---- ProgRecFourier's own arithmetic: the program behind xmipp_reconstruct_fourier ----------
 * reconstruction/reconstruct_fourier.cpp: double accumulators, image-driven scatter into the FFTW-layout Fourier volume with
 * wrap and, beyond the half, the point-mirrored conjugated slot (RF:571-793), correctWeight with its re-processing passes
 * (RF:1056-1101), enforceHermitianSymmetry + PROCESS_WEIGHTS + inverse transform + corrections (RF:451-480,1103-1178).
 *   create   == produceSideInfo (RF:184-287); niter_weight = --iter (0: weights set to one).
 *   insert   == processImages over n projections (shifts already applied, D x D floats on the device): their transform in
 *               double, then the scatter with double-precision atomics; h_ctf nullable (--useCTF, RF:600-625), h_angles
 *               (rot, tilt, psi), h_weights nullable, h_sym nsym x 9 nullable. reprocess != 0: the weight re-processing pass
 *               (images ignored).
 *   weights_step == correctWeight: 0 begin; per further iteration { 1; replay every projection with reprocess = 1; 2 }; 3 end.
 *   finish   == finishComputations -> D^3 doubles on the host; the Fourier volume and weights stay (state_export / import
 *               [F (2 nF) | W (nF)] doubles keep the halves of --prepare_fsc, RF:991-1045).
 * Built for exactness (BASELINE config 1's plumbing path); the accel program is the fast one. */
typedef struct xh_rf2 xh_rf2;
int xh_rf2_create(xh_ctx *ctx, const xh_rf_params *p, int32_t niter_weight, xh_rf2 **out);
int xh_rf2_destroy(xh_rf2 *h);
int xh_rf2_reset(xh_rf2 *h);
int xh_rf2_insert(xh_rf2 *h, const float *d_imgs, const xh_ctf_params *h_ctf, const double *h_angles, const float *h_weights, int32_t n,
                  const double *h_sym, int32_t nsym, int32_t reprocess);
int xh_rf2_weights_step(xh_rf2 *h, int32_t step);
size_t xh_rf2_state_doubles(const xh_rf2 *h);
int xh_rf2_state_export(xh_rf2 *h, double *d_dst);
int xh_rf2_state_import(xh_rf2 *h, const double *d_src, int32_t add);
int xh_rf2_finish(xh_rf2 *h, double *h_volume);

/* ---- CTF pre-steps -------------------------------------------------------------------
 * xmipp_ctf_phase_flip: actualPhaseFlip (reconstruction/ctf_phase_flip.cpp:88-117) -- the coefficients of a micrograph where
 * the undamped CTF (getValuePureWithoutDampingAt, data/ctf.h:541-570) is negative change sign.
 * xmipp_ctf_correct_wiener2d: Wiener2D::wienerFilter + applyWienerFilter (data/wiener2d.cpp:29-141) -- every particle is
 * padded about the Xmipp origin, multiplied in Fourier space by CTF / (CTF^2 + wc) (wc < 0: 0.1 mean CTF^2) and cropped back.
 * A handle serves images of one size: ydim x xdim floats on the device, in place; pad = 1 for phase flipping. CTF
 * descriptions on the host (phase_shift in degrees, as the metadata holds it); sampling_rate = Tm the CTF is evaluated at.
 * is_isotropic is accepted and, as in the reference (which averages the defoci after produceSideInfo has used them), has no
 * effect. CTF in double precision, transforms in fp32. */
typedef struct xh_ctfop xh_ctfop;
int xh_ctfop_create(xh_ctx *ctx, int32_t ydim, int32_t xdim, double pad, xh_ctfop **out);
int xh_ctfop_destroy(xh_ctfop *h);
int xh_ctfop_phase_flip(xh_ctfop *h, float *d_img, const xh_ctf_params *ctf, double sampling_rate);
int xh_ctfop_wiener2d(xh_ctfop *h, float *d_imgs /* [n][ydim][xdim] */, int32_t n, const xh_ctf_params *ctfs, double sampling_rate,
                      int32_t phase_flipped, int32_t is_isotropic, double wiener_constant, int32_t correct_envelope);

/* ---- FlexAlign: global alignment of a movie (BASELINE config 5, first slice) -----------
 * ProgMovieAlignmentCorrelationGPU<T>::computeGlobalAlignment (reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp:
 * 633-725) with the arithmetic of the CPU program (reconstruction/movie_alignment_correlation.cpp:45-157, _base.cpp:152-320,
 * 399-418, eq_system_solver.cpp:35-106): dark / gain correction, Fourier-space reduction of every frame to the size the
 * maximal resolution needs, Gaussian low-pass, correlation of all frame pairs, bestShift within max_shift, least squares over
 * the pair shifts with outlier rejection, reference frame, shift of every frame from it.
 *   create: frames of Y x X pixels at sampling_rate A/px, --maxResForCorrelation A; fails when the scale factor is >= 1.
 *   global_alignment: d_frames [N][Y][X] float on the device, d_dark / d_gain [Y][X] or null, max_shift_px = --maxShift /
 *   sampling rate. Host outputs: pair shifts h_bX / h_bY [N (N-1)/2] (nullable; movie pixels, order (0,1), (0,2) ...),
 *   h_shiftX / h_shiftY [N] from the reference frame h_ref (what storeGlobalShifts negates into the metadata).
 *
 * Local (patch) alignment, ProgMovieAlignmentCorrelationGPU<T>::computeLocalAlignment (movie_alignment_correlation_gpu.cpp:
 * 288-430; the reference has it for CUDA only) and the output of the aligned movie (applyShiftsComputeAverage, :479-570):
 *   local_alignment: patches_x x patches_y patches of patch_size pixels (made even) laid out by getPatchesLocation over what
 *   the global shifts h_gShiftX/Y [N] leave of the frame; per patch the sum of patches_avg frames at their rounded global
 *   shift, reduced in Fourier space to the correlation size (getCorrelationHint: the smallest even size keeping the scale
 *   factor; the reference may pick a larger one after benchmarking cuFFT on the installed GPU), low-passed, all frame pairs
 *   correlated, first maximum within max_shift of the centre, 3 x 3 centre of mass, least squares per patch from frame
 *   ref_frame. Host outputs: h_patchShifts [patches_y][patches_x][N][2] (x, y; rounded global + local), h_centers
 *   [patches_y][patches_x][2], the B-spline coefficients h_coeffsX/Y [lT][lY][lX] of BSplineHelper::computeBSplineCoeffs
 *   (nullable), h_dims[4] = patch size x, y, correlation size x, y (nullable).
 *   local_from_global: localFromGlobal (:432-456), the B-spline of a movie aligned globally only.
 *   apply_bspline: frame n of N (d_frame [Y][X], dark / gain applied on the way) warped by the B-spline like
 *   GeoTransformer::applyBSplineTransform(3, ...) (cuda_gpu_geo_transformer.cpp:186-239): d_out = the aligned frame (nullable),
 *   d_sum += it (nullable), d_initial_sum += the corrected, unaligned frame (nullable). */
typedef struct xh_fa xh_fa;
int xh_fa_create(xh_ctx *ctx, int32_t Y, int32_t X, float sampling_rate, float max_res_for_correlation, xh_fa **out);
int xh_fa_destroy(xh_fa *h);
int xh_fa_info(const xh_fa *h, int32_t *newY, int32_t *newX, double *size_factor);
int xh_fa_set_option(xh_fa *h, const char *name, double value); /* "window" 0: every pair correlation through the full inverse transform (A/B);
                                                                  "pruned_columns" 0: the column pass of the frame transform as full-length line transforms (A/B; default 1:
                                                                  two steps that compute the rows the reduced frame keeps only, as small DFTs on the vector ALUs when Y
                                                                  has a divisor that allows it, else -- or with 2 -- as products on the matrix cores);
                                                                  "rows_kept" 0: the row pass of frames with 5760-point rows by the general kernels (A/B; default 1: one kernel that
                                                                  writes the kept columns only);
                                                                  "pairwin_form" 0: pair / patch correlation windows by the plain kernels (A/B; default 1: packed multiply-adds);
                                                                  "prefilter_ahead" 1: xh_fa_local_alignment ends with the warp's B-spline prefilter of every frame
                                                                  (N Y X floats kept with the handle), run while the host fits the spline; the following
                                                                  xh_fa_apply_bspline[_frames] calls on the same frames (without an initial sum) use it.
                                                                  CONTRACT: the frames are recognised by their device address (d_frames, d_dark, d_gain, N): a caller
                                                                  that rewrites d_frames in place after xh_fa_local_alignment must call it (or xh_fa_global_alignment)
                                                                  again, or switch the option off, before warping. When the device cannot spare the N Y X floats
                                                                  the option silently does nothing (the warp prefilters frame by frame).
                                                                  Control grids: a layer of lX x lY control points must fit 64 KB of LDS (lX lY <= 2048), else
                                                                  xh_fa_apply_bspline fails with XH_ERR_UNSUPPORTED */
int xh_fa_last_full_pairs(const xh_fa *h);                        /* pairs of the last global alignment that needed the full transform */
int xh_fa_global_alignment(xh_fa *h, const float *d_frames, int32_t N, const float *d_dark, const float *d_gain, float max_shift_px,
                           double *h_bX, double *h_bY, double *h_shiftX, double *h_shiftY, int32_t *h_ref);
int xh_fa_local_alignment(xh_fa *h, const float *d_frames, int32_t N, const float *d_dark, const float *d_gain, const double *h_gShiftX,
                          const double *h_gShiftY, int32_t ref_frame, float max_shift_px, int32_t patches_x, int32_t patches_y,
                          int32_t patch_size_x, int32_t patch_size_y, int32_t patches_avg, int32_t lX, int32_t lY, int32_t lT,
                          double *h_patchShifts, double *h_centers, double *h_coeffsX, double *h_coeffsY, int32_t *h_dims);
/* CUDAFlexAlignCorrelate<T>::run (reconstruction_cuda/cuda_flexalign_correlate.cpp:95-140): d_frames [N][Y][X] (even sizes) -> h_pos
 * [N (N-1)/2][2] = (x, y) of the correlation maximum of every pair i < j within max_dist of the centre (X/2, Y/2), refined over 3 x 3 */
int xh_fa_correlate(xh_ctx *ctx, const float *d_frames, int32_t N, int32_t Y, int32_t X, float max_dist, double *h_pos);
int xh_fa_local_from_global(xh_fa *h, int32_t N, const double *h_gShiftX, const double *h_gShiftY, int32_t patches_x, int32_t patches_y,
                            int32_t patch_size_x, int32_t patch_size_y, int32_t lX, int32_t lY, int32_t lT, double *h_centers,
                            double *h_coeffsX, double *h_coeffsY);
int xh_fa_apply_bspline(xh_fa *h, const float *d_frame, const float *d_dark, const float *d_gain, const double *h_coeffsX,
                        const double *h_coeffsY, int32_t lX, int32_t lY, int32_t lT, int32_t N, int32_t n, float *d_out, float *d_sum,
                        float *d_initial_sum);
/* the loop of applyShiftsComputeAverage (:479-560) in one call: frames n0 .. n1 of d_frames [N][Y][X] warped with their frame index and added
 * into d_sum / d_initial_sum (nullable); d_out_stack (nullable, [n1 - n0 + 1][Y][X]) receives the aligned frames */
int xh_fa_apply_bspline_frames(xh_fa *h, const float *d_frames, int32_t N, int32_t n0, int32_t n1, const float *d_dark, const float *d_gain,
                               const double *h_coeffsX, const double *h_coeffsY, int32_t lX, int32_t lY, int32_t lT, float *d_out_stack, float *d_sum,
                               float *d_initial_sum);

/* ProgMovieFilterDose::applyDoseFilterToImage between the two transforms of a frame (reconstruction/movie_filter_dose.cpp:85-170,
 * 283-287): d_frame [Y][X] in place; plan = xh_fft2d_create(ctx, Y, X); acc_voltage 200 or 300 kV (anything else is refused like
 * initVoltage does); dose_start / dose_finish = n / (n + 1) x dosePerFrame + preExposure of frame n. */
int xh_movie_dose_filter(xh_ctx *ctx, xh_fft2d *plan, float *d_frame, int32_t Y, int32_t X, double pixel_size, double acc_voltage,
                         double dose_start, double dose_finish);
/* --bin of the CUDA FlexAlign program: a frame binned while it is loaded (CUDAFlexAlignScale::runScaleIFT,
 * reconstruction_cuda/cuda_flexalign_scale.cpp:101-121; scaleFFT2DKernel, cuda_scaleFFT_kernels.cu:44-79; sizes from
 * AProgMovieAlignmentCorrelation::getMovieSize, reconstruction/movie_alignment_correlation_base.cpp:356-370): half spectrum of the
 * raw frame (minus d_dark, times d_gain: loadFrame corrects before it bins; either may be null) cropped to the binned size, times
 * 1 / (X Y), inverse transform.  d_frame [Y][X] -> d_out [Yb][Xb]; planRaw =
 * xh_fft2d_create(ctx, Y, X), planBinned = xh_fft2d_create(ctx, Yb, Xb). */
int xh_movie_bin_frame(xh_ctx *ctx, xh_fft2d *planRaw, xh_fft2d *planBinned, const float *d_frame, const float *d_dark, const float *d_gain, int32_t Y,
                       int32_t X, float *d_out, int32_t Yb, int32_t Xb);

/* The top-left cropY x cropX window of N frames of Y x X floats, frame after frame (getCroppedFrame,
 * reconstruction_adapt_cuda/movie_alignment_correlation_gpu.cpp:727-734: the CUDA program correlates frames cropped to an
 * FFT-friendly size, findGoodCropSize :73-88).  d_dst: N x cropY x cropX floats. */
int xh_movie_crop_frames(xh_ctx *ctx, const float *d_src, int32_t N, int32_t Y, int32_t X, int32_t cropY, int32_t cropX, float *d_dst);
/* A frame as the detector stores it -> float32 on the device: the cast Image<float>::read does on the host while it reads
 * (xmippCore rwMRC / castPage2T; the movie programs read through it, reconstruction/movie_alignment_correlation_base.cpp:262-266,
 * movie_alignment_correlation_gpu.cpp:667-691), moved behind the host copy so that counts cross the link, not floats.  mode: the MRC
 * data mode of the file -- 0 int8, 1 int16, 2 float32 (a copy), 6 uint16 -- or 100 for uint8 (not an MRC mode); n elements. */
int xh_movie_frame_to_float(xh_ctx *ctx, const void *d_raw, int32_t mode, int64_t n, float *d_out);

/* ---- batched estimator API, first slice ------------------------------------------------------
 * ExtremaFinder::SingleExtremaFinder<T> (reconstruction/single_extrema_finder.cpp:146-300): n signals [n][z][y][x] on the device;
 * search_type 0 Max, 1 Lowest (first of equals, like std::max_element / min_element), 2 MaxAroundCenter, 3 LowestAroundCenter (2-D
 * only: within max_dist of (x/2, y/2), first in raster order); h_positions [n] element offsets as floats (-1: nothing searched),
 * h_values [n]; either may be NULL. */
int xh_extrema_find(xh_ctx *ctx, const float *d_data, int32_t n, int32_t zdim, int32_t ydim, int32_t xdim, int32_t search_type, float max_dist,
                    float *h_positions, float *h_values);
/* Alignment::ShiftCorrEstimator<T>, AlignType::OneToN (reconstruction/shift_corr_estimator.cpp:33-300): create = init2D (even sizes,
 * 0 < max_shift < size / 2); load_reference = load2DReferenceOneToN(const T *); correlate = the static
 * computeCorrelations2DOneToN (d_inout [n][fy][fx] complex spectra <- ref conj(inout), times (-1)^(x+y) when center);
 * compute_shifts = computeShift2DOneToN + getShifts2D: h_shifts [n][2] = (x, y) as the reference returns them.
 * The transforms run in double through line plans that keep a (padded) line in the 64 KB of LDS a workgroup may take: powers of two up
 * to 4096 samples per side, other sizes (Bluestein, padded to the next power of two above 2 n) up to ~2048; larger sides fail with
 * XH_ERR_UNSUPPORTED at create (the typed tests of the reference stop at 768). */
typedef struct xh_shiftcorr xh_shiftcorr;
int xh_shiftcorr_create(xh_ctx *ctx, int32_t xdim, int32_t ydim, int32_t max_shift, xh_shiftcorr **out);
int xh_shiftcorr_destroy(xh_shiftcorr *h);
int xh_shiftcorr_load_reference(xh_shiftcorr *h, const float *d_ref);
int xh_shiftcorr_correlate(xh_ctx *ctx, float *d_inout, const float *d_ref, int32_t n, int32_t fy, int32_t fx, int32_t center);
int xh_shiftcorr_compute_shifts(xh_shiftcorr *h, const float *d_others, int32_t n, float *h_shifts);
/* Alignment::PolarRotationEstimator<T>, AlignType::OneToN (reconstruction/polar_rotation_estimator.cpp:33-144): d_ref [D][D], d_others
 * [n][D][D] -> h_rotations [n] degrees as getRotations2D returns them (an image rotated by a reads 360 - a). The reference's own
 * arithmetic: rings sampled with BsplineOrder 1 (bilinear, zero outside), not normalised, correlation over 2 N - 1 angles
 * (polar_rotation_estimator.cpp:58-60,94-99), double precision, the first maximum (data/polar.cpp:212-233). */
int xh_rotation_estimate(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t D, int32_t first_ring, int32_t last_ring,
                         float *h_rotations);
/* BSplineGeoTransformer<T>::interpolate (reconstruction/bspline_geo_transformer.cpp:103-137): d_dst[i] = applyGeometry(LINEAR, d_src[i],
 * h_matrices[i] (3 x 3 row major), IS_INV, DONT_WRAP), outside value 0; d_src and d_dst [n][ydim][xdim], not aliased */
int xh_apply_geometry2d(xh_ctx *ctx, const float *d_src, int32_t n, int32_t ydim, int32_t xdim, const float *h_matrices, float *d_dst);
/* CorrelationComputer<T>, OneToN, normalised (reconstruction/correlation_computer.cpp:30-56): h_merit[i] = correlationIndex(ref, others[i]) */
int xh_correlation_merit(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t ydim, int32_t xdim, float *h_merit);
/* Alignment::IterativeAlignmentEstimator<T>::compute(others, iters) (reconstruction/iterative_alignment_estimator.cpp:96-176) over the three
 * estimators above: rotation -> shift and shift -> rotation, `iters` rounds each, the images re-interpolated from the originals by the
 * inverse pose after every step; per image the order with the better merit. h_poses [n][9], h_merit [n]. */
int xh_iterative_alignment(xh_ctx *ctx, const float *d_ref, const float *d_others, int32_t n, int32_t D, int32_t max_shift, int32_t first_ring,
                           int32_t last_ring, int32_t iters, float *h_poses, float *h_merit);

#ifdef __cplusplus
}
#endif
#endif /* XMIPP_HIP_H */
