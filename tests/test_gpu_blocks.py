"""GPU: the network's composite modules, one at a time, against the REFERENCE's own module outputs
(tests/golden/op_rb_*, op_attn_*, op_combine, written by oracle/gen_golden.py from the unmodified
ResnetBlockBigGANpp / AttnBlockpp / Combine, layerspp.py:44-91,212-274).  The single-module C-ABI handles
(flowse_block_create / flowse_block_forward) run the same weight packer, launch planner and kernels as the full
model, so this pins GroupNorm -> conv -> temb bias -> GroupNorm -> conv -> shortcut -> skip per block rather than only
through whole-network forwards."""
import pytest
import torch

import _cases as C
from flowmse_amd.util import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5

RB = [("rb_plain", 32, 32, (2, 32, 16, 8), {}, None),
      ("rb_widen", 48, 32, (2, 48, 16, 8), {}, None),
      ("rb_down", 32, 32, (2, 32, 16, 8), dict(down=True), True),
      ("rb_up", 32, 32, (2, 32, 8, 8), dict(up=True), True),
      ("rb_gn12", 384, 128, (1, 384, 8, 8), {}, None)]


def _weights(keys, prefix):
    return {k: torch.from_numpy(synth.synth_param(prefix + k, s)) for k, s in keys}


@pytest.mark.parametrize("tag,cin,cout,shp,kw,sc", RB)
def test_resblock_vs_reference(tag, cin, cout, shp, kw, sc):
    import _gpu as G
    g = C.gold("op_" + tag)
    blk = G.Block("resnet", cin, cout, temb_dim=64, **kw).load(_weights(C.resblock_keys(cin, cout, 64, sc), tag + "."))
    x = torch.from_numpy(synth.normal(5, 5, shp))
    temb = torch.from_numpy(synth.normal(5, 4, (2, 64)))[:shp[0]]
    out = blk(x, temb=temb)
    err = C.rel_l2(out, g["out"])
    print(tag, "rel-L2 vs reference module", err)
    assert out.shape == g["out"].shape and err < TOL


def test_resblock_concat_input_straddling_group():
    """The up-path blocks read cat[h, skip] from two tensors (ncsnpp.py:337); 384 = 256 + 128 channels puts one
    GroupNorm group (12 channels) across the seam."""
    import _gpu as G
    g = C.gold("op_rb_gn12")
    blk = G.Block("resnet", 384, 128, temb_dim=64).load(_weights(C.resblock_keys(384, 128, 64, None), "rb_gn12."))
    x = torch.from_numpy(synth.normal(5, 5, (1, 384, 8, 8)))
    temb = torch.from_numpy(synth.normal(5, 4, (2, 64)))[:1]
    out = blk(x[:, :256].contiguous(), x[:, 256:].contiguous(), temb=temb)
    assert C.rel_l2(out, g["out"]) < TOL


@pytest.mark.parametrize("tag,shape", [("attn_L64", (32, 8, 8)), ("attn_L256", (64, 16, 16)), ("attn_L16", (256, 4, 4))])
def test_attnblock_vs_reference(tag, shape):
    import _gpu as G
    Cc, H, W = shape
    g = C.gold("op_" + tag)
    blk = G.Block("attn", Cc, Cc).load(_weights(C.attn_keys(Cc), tag + "."))
    x = torch.from_numpy(synth.normal(5, 3, (2, Cc, H, W)))
    out = blk(x)
    err = C.rel_l2(out, g["out"])
    print(tag, "rel-L2 vs reference module", err)
    assert err < TOL


def test_combine_vs_reference():
    import _gpu as G
    g = C.gold("op_combine")
    keys = [("Conv_0.weight", (32, 4, 1, 1)), ("Conv_0.bias", (32,))]
    blk = G.Block("combine", 4, 32).load(_weights(keys, "comb."))
    xp = torch.from_numpy(synth.normal(5, 6, (2, 4, 8, 8)))
    yh = torch.from_numpy(synth.normal(5, 7, (2, 32, 8, 8)))
    assert C.rel_l2(blk(xp, yh), g["out"]) < TOL


def test_block_handle_rejects_misuse():
    import ctypes as Ct
    import _gpu as G
    from flowmse_amd import _lib
    h = Ct.c_void_p()
    assert _lib.lib.flowse_block_create(0, 32, 32, 1, 1, 64, Ct.byref(h)) != 0          # up and down
    assert _lib.lib.flowse_block_create(1, 32, 48, 0, 0, 0, Ct.byref(h)) != 0           # attention changes width
    blk = G.Block("resnet", 32, 32, temb_dim=64)
    x = torch.zeros(1, 8, 8, 32, device="cuda")
    with pytest.raises(_lib.FlowseError):                                                 # weights not loaded
        _lib.check(_lib.lib.flowse_block_forward(blk.h, _lib.ptr(x), 32, None, _lib.ptr(x), _lib.ptr(x), 1, 8, 8, None))
