"""ctypes binding of libnb_hip.so (include/nb_hip.h).  PyTorch-ROCm tensors in, raw device
pointers out.  There is NO fallback: if the library is missing or a call fails this raises."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NB_LIB_PATH") or os.path.join(HERE, "lib", "libnb_hip.so")  # NB_LIB_PATH: experiment builds
HEADER = os.path.join(os.path.dirname(HERE), "include", "nb_hip.h")

NB_N_LEVELS = 4
ABI_VERSION = 20
SLOT_DEAD = -2147483648  # NB_SLOT_DEAD
PRECISIONS = {"f32": 0, "f16f6": 1}
PACK_SECTIONS = {"f32": 1, "f16f6": 2}
ILL_SIGMA, ILL_T_MIN = 4e-3, 1e-6  # NB_ILL_SIGMA, NB_ILL_T_MIN


def ill_scratch_bytes(cap):
    """NB_ILL_SCRATCH_BYTES(cap): header + `cap` records of the march's last-sample fix-up list."""
    return 4 * (16 + int(cap) * 16)




class NbFold(C.Structure):
    _fields_ = [
        ("urows", C.c_void_p),                 # dev [(rows + 1), 512] uint16: fp16 heads | remainders of fc_0 . V per active voxel
        ("grid", C.c_void_p * NB_N_LEVELS),    # dev index grids (row id inside the level or -1)
        ("row_base", C.c_int32 * NB_N_LEVELS),
        ("zero_row", C.c_int32),
    ]


class NbScene(C.Structure):
    _fields_ = [
        ("vol", C.c_void_p * NB_N_LEVELS),
        ("vol_dhw", (C.c_int32 * 3) * NB_N_LEVELS),
        ("pose", C.c_void_p),  # dev: R[9] | Th[3] | bounds_min[3]
        ("voxel_size", C.c_float * 3),
        ("out_sh", C.c_int32 * 3),
        ("fold", C.POINTER(NbFold)),  # host pointer or NULL
    ]


class NbCull(C.Structure):
    _fields_ = [
        ("n_views", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("pre_affine", C.c_int32),
        ("msk", C.c_void_p),   # dev [n_views, H, W] uint8
        ("cam", C.c_void_p),   # dev [n_views, 21]: RT 3x4 | K 3x3
        ("snap", C.c_void_p),  # dev R0 (9) | Th0 (3) or NULL
    ]


MLP_PARAM_FIELDS = ["fc0_w", "fc0_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "alpha_w", "alpha_b", "feature_w",
                    "feature_b", "latent_w", "latent_b", "view_w", "view_b", "rgb_w", "rgb_b"]


class NbMlpParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in MLP_PARAM_FIELDS]


_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_I32x3 = C.c_int32 * 3

# name -> (restype, argtypes); must list every function include/nb_hip.h declares
SIGNATURES = {
    "nb_last_error": (C.c_char_p, []),
    "nb_abi_version": (C.c_int, []),
    "nb_device_count": (C.c_int, []),
    "nb_mlp_pack_size": (_I64, []),
    "nb_mlp_latent_bias_size": (_I64, []),
    "nb_mlp_six_bit_stats_offset": (_I64, []),
    "nb_mlp_pack": (C.c_int, [C.POINTER(NbMlpParams), _P, _P]),
    "nb_mlp_pack_sections": (C.c_int, [C.POINTER(NbMlpParams), _P, C.c_int, _P]),
    "nb_mlp_latent_bias": (C.c_int, [C.POINTER(NbMlpParams), _P, _P, _P]),
    "nb_fold_build": (C.c_int, [C.c_void_p * 4, C.c_void_p * 4, C.c_void_p * 4, C.c_int32 * 4, _P, _P, _P, _P]),
    "nb_sparsify": (C.c_int, [_P, _I32x3, _I32, _P, _P, _P, _I32, _P, _P]),
    "nb_decode_points": (C.c_int, [C.POINTER(NbScene), _P, _P, _P, _P, _I64, C.c_int, _P, _P, C.c_int, _P]),
    "nb_march": (C.c_int, [C.POINTER(NbScene), _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _I64, C.POINTER(NbCull), C.c_int,
                           _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int, _P]),
    "nb_composite": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_int, _P, _P, _P, _P, _P, _P]),
    "nb_composite_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_int, _P, _P, _P, _P, _P]),
    "nb_sgemm": (C.c_int, [C.c_int, C.c_int, _I32, _I32, _I32, C.c_float, _P, _I32, _P, _I32, C.c_float, _P, _I32, _P]),
    "nb_gemm_fused": (C.c_int, [C.c_int, C.c_int, _I32, _I32, _I32, C.c_float, _P, _I32, _P, _I32, C.c_float, _P, _I32, _P, _I32, _P, _P]),
    "nb_relu_bwd": (C.c_int, [_P, _P, _I64, _P]),
    "nb_colsum": (C.c_int, [_P, _I64, _I32, _I32, _P, _P]),
    "nb_trilinear_bwd": (C.c_int, [C.POINTER(NbScene), C.c_void_p * 4, C.c_void_p * 4, _P, _P, _I64, _I32, _P]),
    "nb_scan_scratch_size": (_I64, [_I64]),
    "nb_enc_voxelize": (C.c_int, [_P, _I32, _I32x3, _P, _P, _P, _P, _P, _I32, _P]),
    "nb_enc_downsample_index": (C.c_int, [_P, _P, _I32, _I32x3, _I32x3, _P, _P, _P, _I32, _P, _I32, _P]),
    "nb_enc_downsample_index_all": (C.c_int, [_P, _P, _I32, _I32x3, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "nb_enc_conv": (C.c_int, [_P, _P, _I32x3, _P, _P, _I32, _I32x3, _I32, _P, _I32, _I32, _P, _P, _I32, _P]),
    "nb_enc_bn_relu": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "nb_enc_conv_pack16": (C.c_int, [_P, _I32, _I32, _P, _I32, _P]),
    "nb_enc_conv_pack16_batch": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P]),
    "nb_enc_bn_relu_split": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    "nb_enc_conv16": (C.c_int, [_P, _I32, _P, _I32x3, _P, _P, _I32, _I32x3, _I32, _P, _I32, _I32, _P, _P, _I32, _P]),
    "nb_enc_bn_relu_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, C.c_float, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "nb_enc_conv_bwd_input": (C.c_int, [_P, _P, _I32x3, _P, _P, _I32, _I32x3, _I32, _P, _I32, _I32, _P, _P]),
    "nb_enc_conv_bwd_weight": (C.c_int, [_P, _P, _I32x3, _P, _P, _I32, _I32x3, _I32, _P, _P, _I32, _I32, _P, _P, _I32, _P]),
    "nb_enc_scatter_codes_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "nb_enc_gather_codes": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "nb_raygen": (C.c_int, [_I32, _I32, C.c_double * 9, C.c_double * 9, C.c_double * 3, C.c_float * 6, _P, _P, _P,
                            _P, _P, _P, _P, _P]),
    "nb_image_assemble": (C.c_int, [_P, _I64, _P, _P, _I64, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P]),
}

_lib = None


class NbError(RuntimeError):
    pass


def header_functions():
    """Names of all functions declared in include/nb_hip.h (used by the export test)."""
    with open(HEADER) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", src)))


def lib():
    """Load libnb_hip.so (once).  Raises NbError when it has not been built — the product path
    never substitutes another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NbError("libnb_hip.so is missing (%s). Build it with `python -m neuralbody_amd.build`; "
                      "there is no non-HIP fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.nb_abi_version() != ABI_VERSION:
        raise NbError("libnb_hip.so ABI version %d, expected %d" % (L.nb_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().nb_last_error()
        raise NbError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
