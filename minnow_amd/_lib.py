"""ctypes binding of libminnow_hip.so -- exactly the declarations of include/minnow_hip.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MNW_LIB_PATH") or os.path.join(HERE, "libminnow_hip.so")   # (MNW_LIB_PATH: A/B runs of two builds)

MNW_OK = 0
STATUS_NAMES = {
    0: "MNW_OK", -1: "MNW_E_ARG", -2: "MNW_E_BUFLEN", -3: "MNW_E_NOGPU", -4: "MNW_E_HIP",
    -5: "MNW_E_CAPACITY", -6: "MNW_E_DOMAIN", -7: "MNW_E_NOMEM", -8: "MNW_E_IO",
}
K_COUNT = 21
C_COUNT = 8
COUNTER_NAMES = ("single_read_batches", "single_read_aborts", "two_pass_batches", "exact_blocks", "fused_batches", "int_single_read_batches", "vec3_single_read_calls", "backoff_batches")


class MnwError(RuntimeError):
    """Raised where the reference panics (writer.go:34,91-95; bit.go:31,86,90)."""

    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("%s (%d): %s" % (STATUS_NAMES.get(status, "?"), status, msg))


class EncJobC(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("cap", C.c_int64),
                ("pixels", C.c_int64), ("low", C.c_float), ("high", C.c_float),
                ("kind", C.c_int32), ("periodic", C.c_int32), ("xform", C.c_int32), ("reserved", C.c_int32)]


class DecJobC(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("src_len", C.c_int64),
                ("min", C.c_int64), ("bits", C.c_int64), ("pixels", C.c_int64),
                ("low", C.c_float), ("high", C.c_float), ("kind", C.c_int32), ("periodic", C.c_int32),
                ("seed", C.c_uint64), ("block", C.c_uint64), ("xform", C.c_int32), ("reserved", C.c_int32)]


class BlockMetaC(C.Structure):
    _fields_ = [("min", C.c_int64), ("bits", C.c_int64), ("nbytes", C.c_int64), ("status", C.c_int64)]


class MinhColC(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("low", C.c_float), ("high", C.c_float),
                ("log", C.c_int32), ("reserved", C.c_int32)]


class CutColC(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("elem_bytes", C.c_int32), ("reserved", C.c_int32)]


class KernelTimeC(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("launches", C.c_int64)]


# every symbol include/minnow_hip.h declares: name -> (restype, argtypes)
_p, _i64, _u64, _int, _f32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_float
SIGNATURES = {
    "mnw_ctx_create": (_int, [_int, C.POINTER(_p)]),
    "mnw_ctx_destroy": (_int, [_p]),
    "mnw_strerror": (C.c_char_p, [_int]),
    "mnw_last_error": (C.c_char_p, [_p]),
    "mnw_abi_version": (_int, []),
    "mnw_device_count": (_int, []),
    "mnw_precision_needed": (_i64, [_u64]),
    "mnw_array_bytes": (_i64, [_i64, _i64]),
    "mnw_float_group_pixels": (_i64, [_f32, _f32, _f32]),
    "mnw_pack_u64": (_int, [_p, _p, _i64, _int, _p, _i64]),
    "mnw_unpack_u64": (_int, [_p, _p, _i64, _int, _i64, _p]),
    "mnw_bits_u64": (_int, [_p, _p, _i64, C.POINTER(_i64)]),
    "mnw_int_encode": (_int, [_p, _p, _i64, C.POINTER(_i64), C.POINTER(_i64), _p, _i64, C.POINTER(_i64)]),
    "mnw_int_decode": (_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "mnw_float_encode": (_int, [_p, _p, _i64, _f32, _f32, _i64, _int, C.POINTER(_i64), C.POINTER(_i64),
                                _p, _i64, C.POINTER(_i64)]),
    "mnw_float_decode": (_int, [_p, _p, _i64, _i64, _i64, _f32, _f32, _i64, _int, _u64, _u64, _i64, _p]),
    "mnw_encode_blocks": (_int, [_p, _p, C.POINTER(EncJobC), _int, C.POINTER(BlockMetaC)]),
    "mnw_encode_blocks_async": (_int, [_p, _p, C.POINTER(EncJobC), _int]),
    "mnw_encode_blocks_finish": (_int, [_p, _p, C.POINTER(BlockMetaC)]),
    "mnw_decode_blocks": (_int, [_p, _p, C.POINTER(DecJobC), _int]),
    "mnw_pack_u64_dev": (_int, [_p, _p, _p, _i64, _int, _p, _i64]),
    "mnw_unpack_u64_dev": (_int, [_p, _p, _p, _i64, _int, _i64, _p]),
    "mnw_timing_enable": (_int, [_p, _int]),
    "mnw_timing_reset": (_int, [_p]),
    "mnw_vectors_encode": (_int, [_p, _p, _p, _i64, _i64, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_i64), _p, _i64,
                                  C.POINTER(BlockMetaC)]),
    "mnw_vectors_decode": (_int, [_p, _p, _p, _i64, C.POINTER(BlockMetaC), _i64, _i64, C.POINTER(_f32), C.POINTER(_f32),
                                  C.POINTER(_i64), _int, _f32, _u64, _u64, _p]),
    "mnw_vectors_encode_host": (_int, [_p, _p, _i64, _i64, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_i64), _p, _i64,
                                       C.POINTER(BlockMetaC)]),
    "mnw_vectors_decode_host": (_int, [_p, _p, _i64, C.POINTER(BlockMetaC), _i64, _i64, C.POINTER(_f32), C.POINTER(_f32),
                                       C.POINTER(_i64), _int, _f32, _u64, _u64, _p]),
    "mnw_vectors_bounds": (_int, [_p, _p, _p, _i64, C.POINTER(_f32), C.POINTER(_f32)]),
    "mnw_subcell_gather": (_int, [_p, _p, _p, _i64, _i64, _p]),
    "mnw_subcell_scatter": (_int, [_p, _p, _p, _i64, _i64, _int, _f32, _p]),
    "mnw_minh_process": (_int, [_p, _p, C.POINTER(MinhColC), _int]),
    "mnw_minh_unlog": (_int, [_p, _p, C.POINTER(MinhColC), _int]),
    "mnw_minh_selftest": (_int, [_p, _int, _u64, _u64, C.POINTER(_u64)]),
    "mnw_text_parse": (_int, [_p, _p, _i64, C.c_uint8, C.c_uint8, C.POINTER(C.c_int32), _int, C.POINTER(_p),
                              C.POINTER(C.c_int32), _int, C.POINTER(_p), _i64, C.POINTER(_i64)]),
    "mnw_dev_alloc": (_int, [_p, _i64, C.POINTER(_p)]),
    "mnw_dev_free": (_int, [_p, _p]),
    "mnw_copy_to_device": (_int, [_p, _p, _p, _i64]),
    "mnw_copy_file_to_device": (_int, [_p, _p, _int, _int, _p, _p, _p]),
    "mnw_copy_to_host": (_int, [_p, _p, _p, _i64]),
    "mnw_copy_to_host_batch": (_int, [_p, _int, C.POINTER(_p), C.POINTER(_p), C.POINTER(_i64)]),
    "mnw_generic_cut": (_int, [_p, _p, _p, C.c_float, _i64, C.POINTER(CutColC), _int, C.POINTER(_i64)]),
    "mnw_text_rows": (_int, [_p, _p, _i64, C.c_uint8, C.c_uint8, C.POINTER(_i64), C.POINTER(_i64)]),
    "mnw_text_stats": (_int, [_p, C.POINTER(_i64)]),
    "mnw_text_rows_resident": (_int, [_p, _p, _p, _i64, C.c_uint8, C.c_uint8, C.POINTER(_i64), C.POINTER(_i64)]),
    "mnw_timing_get": (_int, [_p, C.POINTER(KernelTimeC)]),
    "mnw_kernel_name": (C.c_char_p, [_int]),
    "mnw_counters_get": (_int, [_p, C.POINTER(_i64)]),
    "mnw_counters_reset": (_int, [_p]),
    "mnw_set_option": (_int, [_p, C.c_char_p, _i64]),
}

_lib = None


def lib_path():
    return _LIB


def lib():
    """Load libminnow_hip.so.  Fails loudly when the HIP extension is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(
                "libminnow_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `python minnow_amd/build.py`. There is no CPU fallback." % _LIB)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; when torch is
        # importable, load it first so that this library binds to the runtime torch already mapped
        # (same SONAME) and device pointers / streams from torch tensors are valid in our launches.
        # Without torch the library resolves libamdhip64.so.7 from /opt/rocm via its RUNPATH.
        if os.environ.get("MNW_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(_LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count():
    return int(lib().mnw_device_count())


def precision_needed(mx):
    """bit.PrecisionNeeded (go/bit/bit.go:19-21)"""
    return int(lib().mnw_precision_needed(C.c_uint64(int(mx) & (2 ** 64 - 1))))


def array_bytes(bits, length):
    """bit.ArrayBytes (go/bit/bit.go:23-25)"""
    return int(lib().mnw_array_bytes(int(bits), int(length)))


def float_group_pixels(low, high, dx):
    """pixels of Writer.FloatGroup (go/writer.go:73)"""
    return int(lib().mnw_float_group_pixels(float(low), float(high), float(dx)))
