"""disparity_to_point_cloud_amd -- MI355X-native disparity -> point-cloud path.

The product is the C-ABI shared library `libd2pc.so` (include/d2pc.h) built
from the hand-written gfx950 kernels under csrc/.  This package is only the
Python plumbing used by the tests, bench.py and the multi-GPU harness:
a ctypes binding of that ABI plus torch.distributed helpers.  It contains no
compute path of its own and no CPU fallback: if the library is missing or no
GPU is usable, calls raise.
"""
from .capi import (  # noqa: F401
    D2pcError,
    PinnedBuffer,
    Context,
    DTYPE_F32,
    DTYPE_U8,
    DTYPE_U16,
    DTYPE_MONO16,
    MODE_PARITY,
    MODE_COMPACT,
    FORM_DEFAULT,
    FORM_CV24,
    FORM_CV4,
    CALIB_BLOB_BYTES,
    abi_version,
    calib_pack,
    calib_unpack,
    device_count,
    library_path,
    load_library,
    make_q,
    make_q_disparity_image,
    make_q_flavour,
    STEREORECTIFY_CONTINUOUS,
    STEREORECTIFY_CV24,
    STEREORECTIFY_CV3,
    roi_points,
    status_string,
    FuseDesc,
    fuse_desc_init,
    crop_to_square,
    FUSE_WEIGHTED_AVERAGE,
    FUSE_MAX_DIST,
    FUSE_MAX_DIST_UNLESS_BLACK,
    FUSE_BETTER_SCORE,
    FUSE_ONLY_GOOD_1,
    FUSE_ONLY_GOOD_AVG,
    FUSE_OVERLAP,
    FUSE_BLACK_TO_WHITE,
    FUSE_GRAD_FILTER,
)
