"""Parameter table of an NCSN++ configuration, straight from the C ABI.

The module list / parameter order lives in exactly one place, the library's
``flowse_model_create`` (csrc/model.hip: build_structure, mirroring
NCSNpp.__init__, reference flowmse/backbones/ncsnpp.py:97-245).  This helper
only reads it back; it needs no GPU.
"""
import ctypes as C

from flowmse_amd import _lib

DEFAULTS = dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), image_size=256)


def normalize_cfg(cfg):
    out = dict(DEFAULTS)
    out.update({k: v for k, v in cfg.items() if k in DEFAULTS})
    out["ch_mult"] = tuple(out["ch_mult"])
    out["attn_resolutions"] = tuple(out["attn_resolutions"])
    return out


def create_handle(cfg):
    cfg = normalize_cfg(cfg)
    c = _lib.make_config(cfg["nf"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["attn_resolutions"],
                         cfg["image_size"])
    h = C.c_void_p()
    _lib.check(_lib.lib.flowse_model_create(C.byref(c), C.byref(h)))
    return h


def handle_param_table(h):
    n = _lib.lib.flowse_model_num_params(h)
    names, shapes, offsets = [], [], []
    buf = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    ndim = C.c_int()
    off = C.c_int64()
    for i in range(n):
        _lib.check(_lib.lib.flowse_model_param_info(h, i, buf, 256, shape, C.byref(ndim), C.byref(off)))
        names.append(buf.value.decode())
        shapes.append([int(shape[k]) for k in range(ndim.value)])
        offsets.append(int(off.value))
    return names, shapes, offsets


def param_table(cfg):
    """(names, shapes) in NCSNpp.state_dict() order for the configuration `cfg`."""
    h = create_handle(cfg)
    try:
        names, shapes, _ = handle_param_table(h)
    finally:
        _lib.lib.flowse_model_destroy(h)
    return names, shapes
