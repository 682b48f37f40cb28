"""xmipp3_amd -- MI355X-native projection matching + Fourier gridding (Xmipp hot path).

The product is libxmipp_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/xmipp_hip.h).  This package is the thin Python host side used by the tests and
bench.py: torch supplies device memory, streams and torch.distributed; every numerical step
runs in the HIP library.  There is no CPU fallback: importing works anywhere, but creating
a Context without the built library or without a gfx950 device raises.
"""
from ._lib import XhError, lib, lib_path  # noqa: F401
from .api import (Context, CtfOps, ShiftCorrEstimator, apply_geometry2d, correlation_merit, extrema_find, iterative_alignment, rotation_estimate, CtfParams, Fft2D, FlexAlign, FourierProjector, ProjectionMatcher, RecFourier, RecFourier2, search5d_offsets, shard_range,  # noqa: F401
                  allreduce_reconstruction, fa_correlate, frc_dpr, movie_bin_frame, movie_binned_size, movie_dose_filter, movie_frames_to_float, reduce_reconstructions)
