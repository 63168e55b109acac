"""CPU: pin the oracle (oracle/*.py) against the reference's own outputs.

The golden vectors were produced by oracle/gen_golden.py from the unmodified
reference imported on CPU.  Tolerance: 1e-5 rel-L2 (same torch, fp32; the
oracle differs from the reference only in op grouping).
"""
import numpy as np
import pytest
import torch

from oracle import ncsnpp_oracle as O
from oracle import sampler_oracle as S
from flowmse_amd.util import synth
import _cases as C

TOL = 1e-5


def test_fir_up_down():
    g = C.gold("op_fir")
    x = torch.from_numpy(synth.normal(5, 1, (2, 8, 16, 32)))
    assert C.rel_l2(O.upsample_2d(x), g["up"]) < TOL
    assert C.rel_l2(O.downsample_2d(x), g["down"]) < TOL
    assert tuple(g["up"].shape) == (2, 8, 32, 64) and tuple(g["down"].shape) == (2, 8, 8, 16)


def test_gfp():
    g = C.gold("op_gfp")
    W = torch.from_numpy(synth.synth_param("gfp.W", (16,)))
    t = torch.from_numpy(g["t"])
    xp = torch.log(t)[:, None] * W[None, :] * 2 * np.pi
    out = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
    assert C.rel_l2(out, g["out"]) < TOL


def test_nin():
    g = C.gold("op_nin")
    W = torch.from_numpy(synth.synth_param("nin.W", (32, 48)))
    b = torch.from_numpy(synth.synth_param("nin.b", (48,)))
    x = torch.from_numpy(synth.normal(5, 2, (2, 32, 8, 4)))
    assert C.rel_l2(O.nin(x, W, b), g["out"]) < TOL


@pytest.mark.parametrize("tag,shape", [("attn_L64", (32, 8, 8)), ("attn_L256", (64, 16, 16)),
                                       ("attn_L16", (256, 4, 4))])
def test_attn(tag, shape):
    Cc, H, W = shape
    g = C.gold("op_" + tag)
    w = C.module_weights(C.attn_keys(Cc), tag + ".")
    x = torch.from_numpy(synth.normal(5, 3, (2, Cc, H, W)))
    out = O.attnblock(O._W(w), 0, x)
    assert C.rel_l2(out, g["out"]) < TOL


RB = [("rb_plain", 32, 32, (2, 32, 16, 8), {}, None),
      ("rb_widen", 48, 32, (2, 48, 16, 8), {}, None),
      ("rb_down", 32, 32, (2, 32, 16, 8), dict(down=True), True),
      ("rb_up", 32, 32, (2, 32, 8, 8), dict(up=True), True),
      ("rb_gn12", 384, 128, (1, 384, 8, 8), {}, None)]


@pytest.mark.parametrize("tag,cin,cout,shp,kw,sc", RB)
def test_resblock(tag, cin, cout, shp, kw, sc):
    g = C.gold("op_" + tag)
    w = C.module_weights(C.resblock_keys(cin, cout, 64, sc), tag + ".")
    x = torch.from_numpy(synth.normal(5, 5, shp))
    temb = torch.from_numpy(synth.normal(5, 4, (2, 64)))[:shp[0]]
    out = O.resblock(O._W(w), 0, x, temb, **kw)
    assert C.rel_l2(out, g["out"]) < TOL


def test_pad_spec_contract():
    g = C.gold("op_pad_spec")
    assert list(g["shape"]) == [1, 1, 256, 512] and float(g["tail"]) == 0.0


def _net_weights(tag):
    t = C.param_tables()[tag]
    return C.synth_weights(t["names"], t["shapes"])


def test_tiny_forward():
    g = C.gold("tiny_forward")
    w = _net_weights("tiny")
    xt, y, _ = C.tiny_inputs()
    out = O.ncsnpp_forward(w, O.make_cfg(**C.TINY), torch.cat([xt, y], 1), torch.from_numpy(g["t"]))
    assert out.shape == (2, 1, 64, 64) and out.dtype == torch.complex64
    assert C.rel_l2(out, g["out"]) < TOL


def test_tiny_sampler():
    g = C.gold("tiny_sampler")
    w = _net_weights("tiny")
    _, y, z = C.tiny_inputs()
    cfg = O.make_cfg(**C.TINY)
    for N in (1, 5):
        x, n = S.euler_sample_net(w, cfg, y, z, N=N)
        assert n == N
        assert C.rel_l2(x, g[f"x_N{N}"]) < 5e-5
    x, _ = S.euler_sample_net(w, cfg, y, z, T_rev=0.8, t_eps=0.05, N=3)
    assert C.rel_l2(x, g["x_N3_T08_e005"]) < 5e-5
    ts, steps = S.time_grid(1.0, 0.03, 5)
    assert np.array_equal(ts.numpy(), g["timesteps_N5"])
    assert abs(float(steps[-1]) - 0.03) < 1e-7 and abs(float(steps[0]) - 0.2425) < 1e-6


def test_wide_forward():
    g = C.gold("wide_forward")
    cfg = O.make_cfg(**C.WIDE)
    # keys/shapes of the wide net are derived from the oracle-independent C-ABI table in
    # test_host_logic; here they come from running the (already pinned) key grammar.
    from flowmse_amd.backbones.structure import param_table
    names, shapes = param_table(cfg)
    w = C.synth_weights(names, shapes)
    xt, y = C.wide_inputs()
    out = O.ncsnpp_forward(w, cfg, torch.cat([xt, y], 1), torch.from_numpy(g["t"]))
    assert C.rel_l2(out, g["out"]) < TOL


@pytest.mark.timeout(300)
def test_full_forward_T64():
    g = C.gold("full_forward_T64")
    w = _net_weights("full")
    xt, y = C.full_inputs()
    out = O.ncsnpp_forward(w, O.make_cfg(), torch.cat([xt, y], 1), torch.from_numpy(g["t"]))
    assert C.rel_l2(out, g["out"]) < TOL
