"""ctypes binding of libtvae_hip.so (the C ABI declared in include/tvae_kernels.h).

There is NO fallback: if the library is missing or does not load, importing this module's
``lib()`` raises.  The product path never routes around the HIP kernels.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch bundles its own libamdhip64; loading ours before it would
#                              put two HIP runtimes in the process (kernels launched on a stream of
#                              the other runtime fail with "no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TVAE_LIB_PATH") or os.path.join(_HERE, "lib", "libtvae_hip.so")   # override: kernel experiments

F32, BF16, F32X3 = 0, 1, 2
GEMM_RELU, GEMM_ACCUM, GEMM_A_KMAJOR, GEMM_B_KMAJOR, GEMM_SPLITK, GEMM_A_DROPMASK, GEMM_RELUMASK = 1, 2, 4, 8, 16, 32, 64
GEMM_B_SPLIT3 = 128
GEMM_B_BF16 = 256
LOSS_STATIC_MAX_GROUPS = 1024

_vp, _i, _f, _u, _i64, _d, _l = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_int64, C.c_double, C.c_long

# name -> argtypes; every function returns int (0 = ok) except the two string getters.
SIGNATURES = {
    "tvae_gemm": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _i, _vp, _u, _f, _vp, _i, _vp],
    "tvae_gemm_batched": [_vp, _i, _l, _vp, _i, _l, _vp, _i, _l, _i, _i, _i, _i, _vp, _vp, _i, _l, _f, _i, _vp],
    "tvae_gru_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tvae_gru_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tvae_softmax_rows_fwd": [_vp, _i, _vp, _i, _i, _i, _vp],
    "tvae_softmax_rows_bwd": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp],
    "tvae_split3": [_vp, _i, _vp, _vp, _i, _i, _vp],
    "tvae_cast_bf16": [_vp, _i, _vp, _vp, _i, _i, _vp],
    "tvae_colsum": [_vp, _i, _vp, _i, _i, _i, _vp, _u, _f, _vp],
    "tvae_layernorm_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "tvae_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "tvae_layernorm_bwd_masked": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _u, _f, _vp],
    "tvae_attn_fwd": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _u, _f, _i, _vp],
    "tvae_attn_bwd": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i,
                      _i, _i, _i, _i, _i, _f, _vp, _u, _f, _i, _vp],
    "tvae_embed_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _u, _f, _vp],
    "tvae_embed_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _u, _f, _vp],
    "tvae_conv_w_pack": [_vp, _vp, _i, _i, _i, _i, _vp],
    "tvae_conv_w_unpack_grad": [_vp, _vp, _i, _i, _i, _i, _vp],
    "tvae_deconv_poly_pack": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tvae_pool_relu_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tvae_pool_relu_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tvae_upsample_fwd": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tvae_upsample_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tvae_reparam_fwd": [_vp, _vp, _vp, _vp, _i, _f, _vp],
    "tvae_reparam_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp],
    "tvae_loss_fwd_bwd": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _i,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tvae_loss_fwd_bwd_static": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _i,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tvae_adam_multi": [_vp, _vp, _vp, _vp, _i, _i, _vp, _d, _d, _d, _d, _d, _d, _d, _i, _vp],
    "tvae_decode_attn": [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tvae_argmax_next": [_vp, _i, _i, _vp, _i, _i, _i, _vp],
    "tvae_scatter_rows": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "tvae_argmax_rows": [_vp, _i, _i, _vp, _i, _vp],
    "tvae_embed_step": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "tvae_residual_dropout": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _u, _f, _vp],
    "tvae_rng_advance": [_vp, _vp],
    "tvae_encode_smiles": [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp, _vp],
    "tvae_gather_batch": [_vp, _i64, _vp, _vp, _i64, _i64, _i, _i, _vp, _i, _vp],
}

_lib = None


class TvaeError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it is not there)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TvaeError("libtvae_hip.so not found at %s -- build it with "
                            "`python -m transvae_amd.build` (there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = C.c_int
        handle.tvae_last_error.restype = C.c_char_p
        handle.tvae_version.restype = C.c_char_p
        _lib = handle
    return _lib


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise TvaeError("%s failed (%d): %s" % (name, rc, lib().tvae_last_error().decode()))
