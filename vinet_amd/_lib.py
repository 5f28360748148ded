"""ctypes binding of libvinet_hip.so (include/vinet_hip.h).

There is no CPU fallback: if the library is missing or a tensor is not on a
GPU, the call raises.  (tests/ may install a C-ABI *test double* through
``_install_test_double`` to exercise the host logic on CPU; the product never
does.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (VINET_LIB: another build of the same ABI, for A/B runs of two builds -- tools/conv_ab.py, bench.py)
LIB_PATH = os.environ.get("VINET_LIB") or os.path.join(HERE, "libvinet_hip.so")

F32, BF16 = 0, 1
F32S = 2     # conv / weight-gradient descriptors: fp32 tensors, split-bf16 matrix arithmetic (include/vinet_hip.h)
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
CONV_GENERIC, CONV_STEM = 0, 1
ABI_VERSION = 13


class CTensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("ld", C.c_int32), ("sB", C.c_int64)]


class CAffine(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int32)]


class CConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_dtype", C.c_int32), ("mode", C.c_int32), ("x", CTensor), ("y", CTensor),
                ("oT", C.c_int32), ("oH", C.c_int32), ("oW", C.c_int32),
                ("sT", C.c_int32), ("sH", C.c_int32), ("sW", C.c_int32),
                ("omT", C.c_int32), ("omH", C.c_int32), ("omW", C.c_int32),
                ("ooT", C.c_int32), ("ooH", C.c_int32), ("ooW", C.c_int32),
                ("ntaps", C.c_int32), ("taps", C.c_void_p), ("w", C.c_void_p), ("Kp", C.c_int32),
                ("pre", CAffine), ("out_scale", C.c_void_p), ("out_shift", C.c_void_p),
                ("act", C.c_int32), ("accumulate", C.c_int32), ("stats", C.c_void_p), ("n_valid", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("tline", C.c_int32), ("tpad", C.c_int32),
                ("bnb_z", C.c_void_p), ("bnb_ld", C.c_int32), ("bnb_sB", C.c_int64), ("bnb_fwd", CAffine),
                ("bnb_mean", C.c_void_p), ("bnb_invstd", C.c_void_p), ("bnb_partials", C.c_void_p)]


class CWgradDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("mode", C.c_int32), ("x", CTensor), ("dy", CTensor),
                ("sT", C.c_int32), ("sH", C.c_int32), ("sW", C.c_int32),
                ("ntaps", C.c_int32), ("taps", C.c_void_p), ("dw", C.c_void_p), ("Kp", C.c_int32), ("pre", CAffine),
                ("tline", C.c_int32), ("tpad", C.c_int32),
                ("bnb_z", C.c_void_p), ("bnb_ld", C.c_int32), ("bnb_sB", C.c_int64), ("bnb_fwd", CAffine),
                ("bnb_mean", C.c_void_p), ("bnb_invstd", C.c_void_p), ("bnb_c1", C.c_void_p), ("bnb_c2", C.c_void_p),
                ("max_cus", C.c_int32)]


class CPoolDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32)] + [(n, C.c_int32) for n in ("kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW")]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_PT, _PC, _PW, _PP = C.POINTER(CTensor), C.POINTER(CConvDesc), C.POINTER(CWgradDesc), C.POINTER(CPoolDesc)

# name -> argtypes (restype is always int unless listed in _RESTYPE)
SIGNATURES = {
    "vinet_conv3d": [_PC, _vp],
    "vinet_conv3d_tile_m": [_PC],
    "vinet_conv3d_stats_rows": [_PC],
    "vinet_conv3d_applies_pre_once": [_PC],
    "vinet_conv3d_splitk_bytes": [_PC],
    "vinet_conv3d_kernel_name": [_PC, C.c_char_p, _i32],
    "vinet_conv3d_wgrad": [_PW, _vp],
    "vinet_conv3d_wgrad_kernel_name": [_PW, C.c_char_p, _i32],
    "vinet_pack_weights": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "vinet_unpack_wgrad": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "vinet_import_ncdhw": [_vp, _i64, _i64, _i64, _i64, _i64, _i32, _PT, _i32, _vp],
    "vinet_import_ncdhw_pad": [_vp, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _PT, _i32, _vp],
    "vinet_export_ncdhw": [_PT, _i32, CAffine, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "vinet_copy_affine": [_PT, _i32, CAffine, _PT, _i32, _i32, _vp],
    "vinet_bn_finalize": [_vp, _i32, _i32, _i32, _f64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "vinet_bn_fold": [_vp, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp],
    "vinet_channel_stats": [_PT, _i32, _vp, _vp],
    "vinet_stats_rows": [_PT],
    "vinet_bn_bwd_reduce": [_PT, _PT, _i32, CAffine, _vp, _vp, _vp, _vp],
    "vinet_conv3d_wgrad_fuses_bn_bwd": [_PW],
    "vinet_conv3d_fuses_dgrad_phases": [_PC],
    "vinet_conv3d_bn_bwd_stats_rows": [_PC],
    "vinet_bn_partials_fold": [_vp, _i32, _i32, _vp, _i32, _vp],
    "vinet_bn_bwd_finalize": [_vp, _i32, _i32, _i32, _f64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "vinet_bn_bwd_apply": [_PT, _PT, _i32, CAffine, _vp, _vp, _vp, _vp, _PT, _vp],
    "vinet_bn_bwd_apply_split": [_PT, _PT, CAffine, _vp, _vp, _vp, _vp, _PT, _PT, _PT, _vp],
    "vinet_act_bwd": [_PT, _i32, _PT, _i32, _i32, _PT, _i32, _vp],
    "vinet_channel_sum": [_PT, _i32, _vp, _i32, _vp, _i32, _vp],
    "vinet_maxpool3d": [_PP, _PT, CAffine, _PT, _vp, _vp],
    "vinet_maxpool3d_bwd": [_PP, _PT, _vp, _PT, _i32, _vp],
    "vinet_upsample2x": [_PT, _PT, _i32, _vp],
    "vinet_unfold1d": [_PT, _PT, _i32, _i32, _i32, _vp],
    "vinet_upsample2x_bwd": [_PT, _PT, _i32, _i32, _vp],
    "vinet_upsample2x_bwd_relu": [_PT, _PT, _PT, _i32, _vp],
    "vinet_loss_fwd": [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "vinet_loss_bwd": [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp],
    "vinet_adam_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "vinet_bilinear_fwd": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "vinet_bilinear_bwd": [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "vinet_resize_blur": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "vinet_minmax": [_vp, _i32, _i64, _vp, _vp],
    "vinet_normalize_u8": [_vp, _vp, _i32, _i64, _vp, _vp],
    "vinet_frames_preprocess_ws_bytes": [_i32, _i32, _i32, _i32, _i32],
    "vinet_frames_preprocess": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, C.POINTER(C.c_float), _vp, _vp],
    "vinet_audio_excerpt": [_vp, _i64, _i64, _i64, _vp, _i32, _vp],
    "vinet_gt_preprocess_ws_bytes": [_i32, _i32, _i32],
    "vinet_gt_preprocess": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "vinet_set_option": [C.c_char_p, _i32],
    "vinet_pack_weights_multi": [_vp, _i32, _i64, _i32, _vp],
    "vinet_unpack_wgrad_multi": [_vp, _i32, _i64, _i32, _vp],
    "vinet_fill_f32": [_vp, _i64, _f32, _vp],
    "vinet_split_bf16": [_PT, CAffine, _PT, _PT, _vp],
    "vinet_debug_spin": [_i64, _vp],
    "vinet_abi_version": [],
    "vinet_last_error": [],
}
_RESTYPE = {"vinet_last_error": C.c_char_p, "vinet_conv3d_splitk_bytes": C.c_int64, "vinet_frames_preprocess_ws_bytes": C.c_int64,
            "vinet_gt_preprocess_ws_bytes": C.c_int64}

_LIB = None
_TEST_DOUBLE = None


class VinetLibraryError(RuntimeError):
    pass


def load(path=LIB_PATH):
    """dlopen the in-tree library and bind every symbol include/vinet_hip.h declares."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(path):
        raise VinetLibraryError(
            "libvinet_hip.so is missing (%s). Build it with `python -m vinet_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % path)
    # torch first: the library depends on libamdhip64.so.7 by soname, and a process must hold ONE HIP runtime -- the
    # one PyTorch-ROCm ships (streams and device pointers cross this boundary).  Loaded before torch, the dependency
    # would bind to /opt/rocm's copy and every launch on a torch stream would fail ("no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise VinetLibraryError("libvinet_hip.so does not export %s" % name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.vinet_abi_version() != ABI_VERSION:
        raise VinetLibraryError("libvinet_hip.so ABI %d != expected %d" % (lib.vinet_abi_version(), ABI_VERSION))
    _LIB = lib
    return lib


LIB_OPTIONS = {}     # library tuning switches set through set_option in this process (engine.config() reports them)


def set_option(name, value):
    """vinet_set_option(name, value) on the active backend (names: include/vinet_hip.h); engine.configure_from_string and
    bench.py --cfg reach it as `lib.<name>=<int>`"""
    lib = get()
    rc = lib.vinet_set_option(name.encode() if isinstance(name, str) else name, int(value))
    if rc != 0:
        msg = lib.vinet_last_error()
        raise VinetLibraryError("set_option(%s): %s" % (name, msg.decode() if isinstance(msg, bytes) else msg))
    LIB_OPTIONS[name if isinstance(name, str) else name.decode()] = int(value)


def get():
    """The active backend: the HIP library, or the installed test double."""
    if _TEST_DOUBLE is not None:
        return _TEST_DOUBLE
    return load()


def is_test_double():
    return _TEST_DOUBLE is not None


def _install_test_double(obj):
    """tests/ only: route C-ABI calls to a CPU model of the ABI (tests/abi_emulator.py)."""
    global _TEST_DOUBLE
    _TEST_DOUBLE = obj


def check(rc, what=""):
    if rc != 0:
        lib = get()
        msg = lib.vinet_last_error()
        if isinstance(msg, bytes):
            msg = msg.decode()
        raise RuntimeError("libvinet_hip %s failed (rc=%d): %s" % (what, rc, msg))
