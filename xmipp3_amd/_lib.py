"""ctypes loader for libxmipp_hip.so (C ABI: include/xmipp_hip.h). Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class XhError(RuntimeError):
    pass


def lib_path():
    # XMIPP_HIP_LIB: another build of the same library (A/B measurements of kernel variants on one box)
    return os.environ.get("XMIPP_HIP_LIB") or os.path.join(_HERE, "libxmipp_hip.so")


class RfParams(C.Structure):
    _fields_ = [("imgSize", C.c_int32), ("padding_proj", C.c_double), ("padding_vol", C.c_double),
                ("max_resolution", C.c_double), ("blob_radius", C.c_double),
                ("blob_order", C.c_int32), ("blob_alpha", C.c_double), ("use_fast", C.c_int32),
                ("phase_flipped", C.c_int32), ("min_ctf", C.c_double), ("sampling", C.c_double)]


class CtfParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "Tm", "kV", "DeltafU", "DeltafV", "azimuthal_angle", "Cs", "Ca", "espr", "ispr", "alpha",
        "DeltaF", "DeltaR", "Q0", "K", "envR0", "envR1", "envR2", "phase_shift", "VPP_radius")]


# every symbol include/xmipp_hip.h declares: name -> (restype, argtypes)
vp, i32, i64, d, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
pvp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "xh_last_error": (C.c_char_p, []),
    "xh_version": (C.c_char_p, []),
    "xh_ctx_create": (C.c_int, [C.c_int, vp, pvp]),
    "xh_ctx_create_private": (C.c_int, [C.c_int, pvp]),
    "xh_ctx_destroy": (C.c_int, [vp]),
    "xh_ctx_sync": (C.c_int, [vp]),
    "xh_ctx_stream": (vp, [vp]),
    "xh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "xh_device_numa_node": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "xh_malloc": (C.c_int, [vp, sz, pvp]),
    "xh_free": (C.c_int, [vp, vp]),
    "xh_memset": (C.c_int, [vp, vp, C.c_int, sz]),
    "xh_memcpy_h2d": (C.c_int, [vp, vp, vp, sz]),
    "xh_memcpy_d2h": (C.c_int, [vp, vp, vp, sz]),
    "xh_host_alloc": (C.c_int, [vp, sz, pvp]),
    "xh_host_free": (C.c_int, [vp, vp]),
    "xh_memcpy_h2d_async": (C.c_int, [vp, vp, vp, sz]),
    "xh_memcpy_d2h_async": (C.c_int, [vp, vp, vp, sz]),
    "xh_ctx_wait_for": (C.c_int, [vp, vp]),
    "xh_timer_create": (C.c_int, [vp, pvp]),
    "xh_timer_start": (C.c_int, [vp, vp]),
    "xh_timer_stop": (C.c_int, [vp, vp]),
    "xh_timer_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "xh_timer_destroy": (C.c_int, [vp, vp]),
    "xh_ctf_defaults": (None, [C.POINTER(CtfParams)]),
    "xh_rf_create": (C.c_int, [vp, C.POINTER(RfParams), pvp]),
    "xh_rf_destroy": (C.c_int, [vp]),
    "xh_rf_set_option": (C.c_int, [vp, C.c_char_p, d]),
    "xh_rf_sizes": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "xh_rf_tables": (C.c_int, [vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "xh_rf_temp_floats": (sz, [vp]),
    "xh_rf_attach_temp": (C.c_int, [vp, vp]),
    "xh_rf_temp_ptr": (C.c_int, [vp, pvp]),
    "xh_rf_reset": (C.c_int, [vp]),
    "xh_rf_shift_images": (C.c_int, [vp, vp, vp, vp, i32, vp]),
    "xh_rf_shift_images_coefs": (C.c_int, [vp, vp, vp, vp, vp, i32, vp]),
    "xh_rf_prepare_images": (C.c_int, [vp, vp, i32, vp]),
    "xh_rf_ctf_arrays": (C.c_int, [vp, C.POINTER(CtfParams), i32, vp, vp]),
    "xh_rf_insert": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, i32]),
    "xh_rf_insert_images": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i32]),
    "xh_rf_insert_images_dev": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i32]),
    "xh_rf_shift_images_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp]),
    "xh_rf_insert_matrices": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, i32]),
    "xh_rf_mirror_and_crop": (C.c_int, [vp]),
    "xh_rf_cropped_floats": (sz, [vp]),
    "xh_rf_cropped_export": (C.c_int, [vp, vp]),
    "xh_rf_cropped_import": (C.c_int, [vp, vp, i32]),
    "xh_rf_reduce": (C.c_int, [pvp, i32]),
    "xh_rf_finish": (C.c_int, [vp, vp]),
    "xh_rf2_create": (C.c_int, [vp, C.POINTER(RfParams), i32, pvp]),
    "xh_rf2_destroy": (C.c_int, [vp]),
    "xh_rf2_reset": (C.c_int, [vp]),
    "xh_rf2_insert": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i32, i32]),
    "xh_rf2_weights_step": (C.c_int, [vp, i32]),
    "xh_rf2_state_doubles": (sz, [vp]),
    "xh_rf2_state_export": (C.c_int, [vp, vp]),
    "xh_rf2_state_import": (C.c_int, [vp, vp, i32]),
    "xh_rf2_finish": (C.c_int, [vp, vp]),
    "xh_fa_create": (C.c_int, [vp, i32, i32, C.c_float, C.c_float, pvp]),
    "xh_fa_destroy": (C.c_int, [vp]),
    "xh_fa_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(d)]),
    "xh_fa_set_option": (C.c_int, [vp, C.c_char_p, d]),
    "xh_fa_last_full_pairs": (C.c_int, [vp]),
    "xh_fa_global_alignment": (C.c_int, [vp, vp, i32, vp, vp, C.c_float, vp, vp, vp, vp, C.POINTER(i32)]),
    "xh_fa_local_alignment": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, i32, C.c_float, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "xh_extrema_find": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, C.c_float, vp, vp]),
    "xh_shiftcorr_create": (C.c_int, [vp, i32, i32, i32, pvp]),
    "xh_shiftcorr_destroy": (C.c_int, [vp]),
    "xh_shiftcorr_load_reference": (C.c_int, [vp, vp]),
    "xh_shiftcorr_correlate": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
    "xh_shiftcorr_compute_shifts": (C.c_int, [vp, vp, i32, vp]),
    "xh_apply_geometry2d": (C.c_int, [vp, vp, i32, i32, i32, vp, vp]),
    "xh_correlation_merit": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "xh_iterative_alignment": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "xh_rotation_estimate": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "xh_movie_dose_filter": (C.c_int, [vp, vp, vp, i32, i32, d, d, d, d]),
    "xh_movie_bin_frame": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, i32]),
    "xh_movie_crop_frames": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "xh_movie_frame_to_float": (C.c_int, [vp, vp, i32, C.c_int64, vp]),
    "xh_fa_correlate": (C.c_int, [vp, vp, i32, i32, i32, C.c_float, vp]),
    "xh_fa_local_from_global": (C.c_int, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "xh_fa_apply_bspline": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "xh_fa_apply_bspline_frames": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "xh_ctfop_create": (C.c_int, [vp, i32, i32, d, pvp]),
    "xh_ctfop_destroy": (C.c_int, [vp]),
    "xh_ctfop_phase_flip": (C.c_int, [vp, vp, C.POINTER(CtfParams), d]),
    "xh_ctfop_wiener2d": (C.c_int, [vp, vp, i32, vp, d, i32, i32, d, i32]),
    "xh_pm_create": (C.c_int, [vp, i32, i32, i32, i32, vp, vp, i32, pvp]),
    "xh_pm_destroy": (C.c_int, [vp]),
    "xh_pm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "xh_pm_match": (C.c_int, [vp, vp, i32, vp, vp, i32, vp, vp, vp]),
    "xh_frc_dpr": (C.c_int, [vp, vp, vp, i32, i32, i32, d, i32, i32, d, d, vp, vp, vp, vp, vp, vp]),
    "xh_rf_kernel_ms": (C.c_int, [vp, vp, vp, i32]),
    "xh_fft2d_create": (C.c_int, [vp, i32, i32, pvp]),
    "xh_fft2d_destroy": (C.c_int, [vp]),
    "xh_fft2d_factors": (C.c_int, [vp, vp]),
    "xh_fft2d_exec": (C.c_int, [vp, vp, i32]),
    "xh_fft2d_exec_axis": (C.c_int, [vp, vp, i32, i32]),
    "xh_fp_create": (C.c_int, [vp, vp, i32, C.c_double, C.c_double, i32, pvp]),
    "xh_fp_destroy": (C.c_int, [vp]),
    "xh_fp_info": (C.c_int, [vp, vp, vp, vp]),
    "xh_fp_coefs": (C.c_int, [vp, vp, vp]),
    "xh_fp_project": (C.c_int, [vp, vp, i32, vp, vp]),
    "xh_pm_match_ex": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "xh_pm_translate": (C.c_int, [vp, vp, i32, vp, vp, vp, d, vp, vp, vp]),
    "xh_pm_last_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "xh_pm_translate_stats": (C.c_int, [vp, C.POINTER(i64)]),
    "xh_pm_last_coefficients": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]),
    "xh_pm_stage_ms": (C.c_int, [vp, vp, i32]),
    "xh_pm_rows_pruned": (C.c_int, [vp, vp]),
    "xh_pm_two_level_cut": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "xh_pm_set_option": (C.c_int, [vp, C.c_char_p, d]),
    "xh_pm_debug_prepare": (C.c_int, [vp, vp, i32, i32, vp, vp]),
    "xh_pm_debug_ref": (C.c_int, [vp, i32, vp, vp]),
    "xh_pm_debug_corr_rows": (C.c_int, [vp, vp, i32, i32, vp]),
    "xh_pm_debug_s6_maps": (C.c_int, [vp, i32, vp]),
    "xh_pm_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(d)]),
}


def lib():
    """Load the HIP library; raise XhError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own HIP runtime; load it first so that this process ends up with ONE
    # libamdhip64 (ours would otherwise pull /opt/rocm's next to torch's and lose the device)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise XhError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(xmipp3_amd/csrc/build.sh). There is no CPU fallback.")
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise XhError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(L, name)
        except AttributeError as e:
            raise XhError(f"{path} does not export {name}") from e
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise XhError(f"xmipp_hip error {rc}: {lib().xh_last_error().decode()}")
