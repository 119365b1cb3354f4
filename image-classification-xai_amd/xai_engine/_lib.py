"""ctypes binding of libxai_hip.so (C ABI: include/xai_hip.h).

There is no CPU fallback: if the shared library is missing or a tensor is not on a HIP
device the call raises.  Build with `make -C image-classification-xai_amd/csrc`
(or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libxai_hip.so")

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_d = C.c_double

# name -> argument types, exactly the prototypes of include/xai_hip.h (return type int unless noted)
SIGNATURES = {
    "xai_version": [],
    "xai_version_minor": [],
    "xai_strerror": [_i],
    "xai_ig_interp_f32": [_p, _p, _f, _p, _l, _i, _i, _l, _p, _p],
    "xai_ig_cutoff_f32": [_p, _i, _i, _f, _p, _p],
    "xai_ig_accum_f32": [_p, _i, _i, _p, _i, _p, _p, _p, _p, _f, _i, _l, _p, _p, _p],
    "xai_ig_accum_timed_f32": [_p, _i, _i, _p, _i, _p, _p, _p, _p, _f, _i, _l, _p, _p, _p, _p, _p],
    "xai_ig_store_grads_f32": [_p, _p, _l, _p],
    "xai_ig_accum_add_f32": [_p, _i, _p, _l, _p],
    "xai_ig_finish_f32": [_p, _i, _i, _p, _p, _f, _i, _l, _p, _p, _p],
    "xai_sumsq_f32": [_p, _i, _l, _p, _p],
    "xai_idgi_accum_f32": [_p, _i, _p, _p, _l, _p, _p],
    "xai_gradcam_workspace_bytes": [_i, _i, _i, _i],
    "xai_gradcam_f32": [_p, _p, _i, _i, _i, _i, _i, _p, _p, C.c_size_t, _p],
    "xai_bilinear_up_f32": [_p, _i, _i, _i, _i, _i, _f, _i, _p, _p],
    "xai_rise_apply_f32": [_p, _p, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p],
    "xai_rise_accum_f64": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _d, _p, _p],
    "xai_rank_workspace_bytes": [_i, _l],
    "xai_rank_f32": [_p, _i, _l, _p, _p, _p, C.c_size_t, _p],
    "xai_flip_steps_i32": [_p, _l, _i, _i, _p, _p],
    "xai_perturb_batch_f32": [_p, _p, _p, _i, _l, _i, _i, _p, _p],
    "xai_segment_sums_f32": [_p, _p, _l, _i, _i, _i, _p, _p, _p],
    "xai_blur_sep_f32": [_p, _p, _i, _i, _i, _i, _i, _p, _p],
    "xai_blur_1d_f32": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "xai_softmax_stats_f32": [_p, _i, _i, _p, _i, _p, _p, _p, _p],
    "xai_up_rownorm_f32": [_p, _i, _i, _i, _i, _i, _p, _p],
    "xai_rownorm_f32": [_p, _i, _l, _p, _p],
    "xai_cluster_sum_f32": [_p, _p, _p, _i, _l, _p, _p],
    "xai_causal_apply_f32": [_p, _p, _p, _i, _i, _l, _f, _p, _p],
    "xai_masked_sums_f32": [_p, _p, _i, _l, _p, _p, _p],
    "xai_bn_act_fwd_f32": [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _f, _i, _i, _i, _i, _i, _p, _p],
    "xai_bn_relu_bwd_f32": [_p, _p, _p, _p, _p, _f, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p],
    "xai_maxpool_bwd_f32": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    "xai_bn_relu_maxpool_fwd_f32": [_p, _p, _p, _p, _p, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p],
}
_RESTYPE = {"xai_strerror": C.c_char_p, "xai_rank_workspace_bytes": C.c_size_t, "xai_gradcam_workspace_bytes": C.c_size_t}

ABI_VERSION, ABI_MINOR = 1, 3       # XAI_ABI_VERSION / XAI_ABI_MINOR of include/xai_hip.h this binding was written against

_lib = None


class XaiHipError(RuntimeError):
    pass


def load():
    """Load the library once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XaiHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C "
            f"{os.path.join(os.path.dirname(_HERE), 'csrc')}` (needs hipcc, targets gfx950). "
            "There is deliberately no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # version first, by the two symbols every build of the library has had or that say "older" by their absence
    lib.xai_version.restype = C.c_int
    major = lib.xai_version()
    minor = 0
    if hasattr(lib, "xai_version_minor"):
        lib.xai_version_minor.restype = C.c_int
        minor = lib.xai_version_minor()
    if major != ABI_VERSION or minor < ABI_MINOR:
        raise XaiHipError(f"{LIB_PATH} has ABI {major}.{minor}, this package needs {ABI_VERSION}.{ABI_MINOR} or a later minor "
                          f"(include/xai_hip.h); rebuild with `make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}`")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().xai_strerror(code).decode()
        raise XaiHipError(f"{what} failed with code {code}: {msg}")
