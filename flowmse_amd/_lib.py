"""ctypes binding of libflowse_hip.so (C ABI declared in include/flowse_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  Importing this module fails loudly when the
shared object is missing (build it with ``python -m flowmse_amd.build`` or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

# torch FIRST: PyTorch-ROCm ships its own libamdhip64.  The library must bind to that same HIP runtime instance
# (one runtime per process: device pointers and streams are shared with torch tensors); loaded the other way round
# the dynamic linker would give libflowse_hip.so the system runtime and its launches would not see torch's device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLOWSE_LIB_PATH: load another build of the same library (A-B timing of kernel variants, tools/build_variants.py)
LIB_PATH = os.environ.get("FLOWSE_LIB_PATH") or os.path.join(_HERE, "libflowse_hip.so")

FLOWSE_MAX_LEVELS = 8
FLOWSE_MAX_ATTN = 4


class FlowseError(RuntimeError):
    pass


class flowse_config(C.Structure):
    _fields_ = [("nf", C.c_int32), ("num_levels", C.c_int32), ("ch_mult", C.c_int32 * FLOWSE_MAX_LEVELS),
                ("num_res_blocks", C.c_int32), ("num_attn", C.c_int32),
                ("attn_resolutions", C.c_int32 * FLOWSE_MAX_ATTN), ("image_size", C.c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
        "Build it with `python -m flowmse_amd.build`.")

lib = C.CDLL(LIB_PATH)

_vp, _fp, _i, _i64, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol declared in include/flowse_hip.h
SIGNATURES = {
    "flowse_abi_version": (_i, []),
    "flowse_last_error": (C.c_char_p, []),
    "flowse_device_count": (_i, []),
    "flowse_model_create": (_i, [C.POINTER(flowse_config), C.POINTER(_vp)]),
    "flowse_model_destroy": (None, [_vp]),
    "flowse_block_create": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "flowse_block_forward": (_i, [_vp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "flowse_model_num_params": (_i, [_vp]),
    "flowse_model_num_modules": (_i, [_vp]),
    "flowse_model_blob_numel": (_i64, [_vp]),
    "flowse_model_param_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i64)]),
    "flowse_model_load_weights": (_i, [_vp, _fp, _i64]),
    "flowse_model_set_precision": (_i, [_vp, _i]),
    "flowse_model_reserve": (_i, [_vp, _i, _i, _i, C.POINTER(_i64)]),
    "flowse_vf_forward": (_i, [_vp, _vp, _vp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "flowse_prior_sample": (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    "flowse_euler_sample": (_i, [_vp, _vp, _vp, C.POINTER(_f), C.POINTER(_f), _i, _i, _i, _i, _vp]),
    "flowse_rk_sample": (_i, [_vp, _vp, _vp, C.POINTER(_f), C.POINTER(_f), _i, _i, _i, _i, _i, _vp]),
    "flowse_model_graph_launches": (_i64, [_vp]),
    "flowse_axpy": (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    "flowse_stft_compress": (_i, [_fp, _i, _i, _f, _vp, _i, _i, _f, _f, _vp]),
    "flowse_istft_decompress": (_i, [_vp, _i, _i, _i, _f, _f, _fp, _i, _f, _vp]),
    "flowse_profile_begin": (_i, [_vp, _i]),
    "flowse_profile_end": (_i, [_vp, C.c_char_p, _i]),
    "flowse_upfirdn2d": (_i, [_fp, _fp] + [_i] * 13 + [_fp, _i, _i, _vp]),
    "flowse_op_conv2d": (_i, [_fp, _i, _fp, _i, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _vp]),
    "flowse_op_conv2d_16": (_i, [_fp, _i, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _i, _f, _i, _vp, _i64,
                                _vp]),
    "flowse_op_resblock_tail_16": (_i, [_fp, _i, _fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _f,
                                       _i, _vp, _i64, _vp]),
    "flowse_op_pc16_channel_blocks": (_i, [_i]),
    "flowse_op_conv2d_scratch_floats": (_i64, [_i, _i, _i, _i, _i, _i]),
    "flowse_op_conv3x3_gn": (_i, [_fp, _i, _fp, _i, _fp, _fp, _f, _i, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _f, _fp, _vp]),
    "flowse_op_conv3x3_f43": (_i, [_fp, _i, _fp, _i, _fp, _fp, _f, _i, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _f, _fp, _vp]),
    "flowse_op_conv3x3_f43_scratch_floats": (_i64, [_i, _i, _i, _i, _i]),
    "flowse_op_conv3x3_w2d": (_i, [_fp, _i, _fp, _i, _fp, _fp, _f, _i, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _f, _fp, _vp]),
    "flowse_op_conv3x3_w2d_scratch_floats": (_i64, [_i, _i, _i, _i, _i]),
    "flowse_op_group_norm_scratch_floats": (_i64, [_i, _i, _i]),
    "flowse_op_group_norm": (_i, [_fp, _i, _fp, _i, _fp, _fp, _f, _i, _fp, _i, _i, _i, _fp, _vp]),
    "flowse_op_fir_up": (_i, [_fp, _fp, _i, _i, _i, _i, _vp]),
    "flowse_op_fir_down": (_i, [_fp, _fp, _i, _i, _i, _i, _vp]),
    "flowse_op_attention": (_i, [_fp, _fp, _i, _i, _i, _vp]),
    "flowse_op_gfp": (_i, [_fp, _fp, _fp, _i, _i, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header / library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc):
    if rc != 0:
        raise FlowseError(f"flowse error {rc}: {lib.flowse_last_error().decode(errors='replace')}")


def make_config(nf, ch_mult, num_res_blocks, attn_resolutions, image_size):
    ch_mult = tuple(int(c) for c in ch_mult)
    attn = tuple(int(a) for a in attn_resolutions)
    if len(ch_mult) > FLOWSE_MAX_LEVELS or len(attn) > FLOWSE_MAX_ATTN:
        raise ValueError("too many levels / attention resolutions")
    cfg = flowse_config()
    cfg.nf = int(nf)
    cfg.num_levels = len(ch_mult)
    for i, c in enumerate(ch_mult):
        cfg.ch_mult[i] = c
    cfg.num_res_blocks = int(num_res_blocks)
    cfg.num_attn = len(attn)
    for i, a in enumerate(attn):
        cfg.attn_resolutions[i] = a
    cfg.image_size = int(image_size)
    return cfg


def ptr(t):
    """Raw address of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
