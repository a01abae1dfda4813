"""ctypes binding of libsynthray.so (include/synthray.h).

The library is the product: there is no NumPy or CPU fallback behind it.  If it has
not been built, importing this module raises; if no MI355X is visible, the first
call that needs the device raises RuntimeError with the HIP error text.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYNTHRAY_LIB") or os.path.join(_HERE, "libsynthray.so")  # override: A/B builds of the kernels


class SynthrayError(RuntimeError):
    """A libsynthray call returned a negative status."""


class Optic(C.Structure):
    _fields_ = [("op", C.c_int32), ("iarg", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class TraceParams(C.Structure):
    _fields_ = [("t_end", C.c_double), ("extent", C.c_double), ("dt", C.c_double), ("probing_axis", C.c_int32),
                ("row_order", C.c_int32), ("substeps", C.c_int32), ("sort_rays", C.c_int32), ("precision", C.c_int32),
                ("handoff", C.c_int32)]


class TraceStats(C.Structure):
    _fields_ = [("ray_steps", C.c_int64), ("fallback_rays", C.c_int64), ("trace_kernel_ms", C.c_double),
                ("total_ms", C.c_double)]


MAX_REF_BEAMS = 4  # SR_MAX_REF_BEAMS


class DepositParams(C.Structure):
    _fields_ = [("kwave", C.c_double), ("ref_n_fringes", C.c_double * MAX_REF_BEAMS), ("ref_deg", C.c_double * MAX_REF_BEAMS),
                ("ref_on", C.c_int32), ("lds_tiles", C.c_int32), ("exact_counts", C.c_int32), ("reserved", C.c_int32)]


class DepositStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("deposited", C.c_int64), ("retraced", C.c_int64)]


# every symbol include/synthray.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "sr_init": (_i, [_i]),
    "sr_device_count": (_i, []),
    "sr_synchronize": (_i, []),
    "sr_device_memory": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sr_last_error": (C.c_char_p, []),
    "sr_version": (C.c_char_p, []),
    "sr_stream_select": (_i, [_i]),
    "sr_stream_wait": (_i, [_i, _i]),
    "sr_host_alloc": (_i, [_pp, C.c_size_t]),
    "sr_host_free": (None, [_vp]),
    "sr_volume_create": (_i, [_pp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _d, _i, _i]),
    "sr_volume_create_from_fields": (_i, [_pp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _vp, _i]),
    "sr_volume_fields": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "sr_field_ifft_real": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sr_radial_spectrum2d": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "sr_volume_sample": (_i, [_vp, _vp, _i64, _vp]),
    "sr_volume_attach_aux": (_i, [_vp, _vp, _vp, _vp, _d]),
    "sr_volume_create_slab": (_i, [_pp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _d, _i, _i, _i, _i]),
    "sr_rays_handoff_download": (_i, [_vp, _vp]),
    "sr_rays_handoff_upload": (_i, [_vp, _vp]),
    "sr_rays_handoff_send": (_i, [_vp, _vp, _i]),
    "sr_rays_handoff_recv": (_i, [_vp, _vp, _i]),
    "sr_volume_sample_aux": (_i, [_vp, _vp, _i64, _vp]),
    "sr_volume_omega": (_d, [_vp]),
    "sr_volume_bytes": (_i64, [_vp]),
    "sr_volume_destroy": (None, [_vp]),
    "sr_trace": (_i, [_vp, _vp, _i64, C.POINTER(TraceParams), _vp, _vp, _vp, C.POINTER(TraceStats)]),
    "sr_ray_to_jones": (_i, [_vp, _i64, _d, _i, _i, _vp, _vp]),
    "sr_rays_create": (_i, [_pp, _i64]),
    "sr_rays_download_s0": (_i, [_vp, _vp]),
    "sr_rays_error_bound": (_i, [_vp, _vp]),
    "sr_rays_generate": (_i, [_vp, _i, _d, _d, _d, _d, _i, C.c_uint64, C.c_uint64]),
    "sr_rays_upload": (_i, [_vp, _vp]),
    "sr_rays_upload_part": (_i, [_vp, _vp, _i64, _i64, _i]),
    "sr_rays_trace": (_i, [_vp, _vp, C.POINTER(TraceParams), C.POINTER(TraceStats)]),
    "sr_release_caches": (_i, []),
    "sr_rays_trace_stats": (_i, [_vp, C.POINTER(TraceStats)]),
    "sr_rays_tile_segments": (_i, [_vp]),
    "sr_tile_min_density": (_d, []),
    "sr_rays_tile_records": (_i, [_vp]),
    "sr_rays_set_bbox": (_i, [_vp, _vp]),
    "sr_rays_get_bbox": (_i, [_vp, _vp, C.POINTER(C.c_int)]),
    "sr_rays_download": (_i, [_vp, _vp, _vp, _vp]),
    "sr_rays_count": (_i64, [_vp]),
    "sr_rays_destroy": (None, [_vp]),
    "sr_optics": (_i, [C.POINTER(Optic), _i, _d, _i64, _vp, _vp, _vp, _vp]),
    "sr_hist2d": (_i, [_vp, _vp, _i64, _i, _i, _d, _d, _d, _d, _vp]),
    "sr_interferogram": (_i, [_vp, _vp, _vp, _i64, _i, _i, _d, _d, _d, _d, _vp, _vp]),
    "sr_interfere_ref_beam": (_i, [_vp, _vp, _i64, _d, _d, _vp]),
    "sr_image_create": (_i, [_pp, _i, _i, _i, _d, _d, _d, _d]),
    "sr_image_zero": (_i, [_vp]),
    "sr_image_download": (_i, [_vp, _vp]),
    "sr_image_amplitude": (_i, [_vp, _vp]),
    "sr_image_counts_f64": (_i, [_vp, _vp]),
    "sr_rays_optics": (_i, [_vp, C.POINTER(Optic), _i, C.POINTER(DepositParams), _vp, _vp]),
    "sr_image_bytes": (_i64, [_vp]),
    "sr_image_destroy": (None, [_vp]),
    "sr_rays_deposit": (_i, [_vp, C.POINTER(Optic), _i, C.POINTER(DepositParams), _vp, C.POINTER(DepositStats)]),
    "sr_rays_refine": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "sr_comm_unique_id": (_i, [_vp]),
    "sr_comm_create": (_i, [_pp, _vp, _i, _i]),
    "sr_image_reduce": (_i, [_vp, _vp, _i]),
    "sr_comm_ranks": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sr_comm_destroy": (None, [_vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: synthpy_amd has no CPU path. Build the HIP library first:\n"
        "    python -c 'import __graft_entry__ as g; g.build()'   (or: make -C synthpy_amd/csrc)")

# RCCL between processes (sr_comm_create, the slab hand-off) shares device buffers through dmabuf IPC; hosts whose driver has no
# legacy IPC fail in hipIpcGetMemHandle unless the runtime is told so BEFORE it starts.  A launcher that exported a value keeps it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of step
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.sr_last_error().decode("utf-8", "replace")


gpu_touched = False  # set by the first library call that goes through check(): from then on this process must not fork


def check(rc: int) -> None:
    global gpu_touched
    gpu_touched = True
    if rc != 0:
        raise SynthrayError(f"libsynthray error {rc}: {last_error()}")


def ptr(a):
    """Raw pointer of a C-contiguous NumPy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
