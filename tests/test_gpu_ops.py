"""GPU: per-kernel parity through the C ABI against the CPU oracle / plain torch fp32 on the same seeded inputs.

Tolerance (fp32 path): 1e-5 rel-L2 per op unless stated (north_star bar for the whole sampler: 1e-3).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases as C
from flowmse_amd.util import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def G():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import _gpu
    return _gpu


def rnd(seed, shape, std=1.0):
    return torch.from_numpy(synth.normal(77, seed, shape, std))


CONV_CASES = [
    # B, H, W, C1, C2, Cout, k, bias, bias2, res, scale
    (2, 16, 16, 32, 0, 32, 3, True, False, False, 1.0),
    (1, 8, 12, 128, 0, 128, 3, True, True, True, 0.70710678),
    (2, 6, 10, 256, 128, 256, 3, True, True, False, 1.0),      # concat 384, M not multiple of 128
    (1, 16, 16, 256, 256, 256, 1, True, False, False, 1.0),    # 1x1 on concat 512
    (3, 4, 4, 64, 0, 192, 1, True, False, False, 1.0),         # qkv-like, tile spans samples
    (1, 32, 16, 128, 0, 4, 3, True, False, True, 1.0),         # pyramid head (Cout=4) + in-place style residual
    (2, 8, 8, 16, 0, 16, 3, False, False, False, 1.0),         # tiny-net widths (K chunk padded)
    (2, 8, 8, 32, 16, 48, 3, True, False, False, 1.0),         # concat 48 (chunk straddles sources)
    (2, 16, 8, 4, 0, 64, 3, True, False, False, 1.0),          # input layer (direct kernel)
    (2, 16, 8, 4, 0, 32, 1, True, False, True, 1.0),           # Combine (direct kernel, residual)
    (1, 64, 64, 128, 0, 128, 3, True, True, True, 0.70710678),  # several M tiles
    (1, 2, 2, 512, 0, 256, 3, True, False, False, 1.0),        # deep K, tiny image
    (2, 128, 144, 4, 0, 128, 3, True, False, False, 1.0),      # input layer on the matrix cores (>= 256 blocks)
    (1, 256, 128, 4, 0, 128, 3, True, True, True, 0.5),
    (4, 64, 64, 256, 0, 4, 3, True, False, True, 1.0),          # 4-channel head on the 4x4x1 MFMA kernel
    (1, 144, 272, 64, 0, 4, 3, False, False, False, 0.5),       # same, 9 x 17 tiles, no bias / residual
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(G, case):
    B, H, W, C1, C2, Cout, k, has_b, has_b2, has_res, scale = case
    x1 = rnd(1, (B, C1, H, W))
    x2 = rnd(2, (B, C2, H, W)) if C2 else None
    w = rnd(3, (Cout, C1 + C2, k, k), (1.0 / ((C1 + C2) * k * k)) ** 0.5)
    bias = rnd(4, (Cout,), 0.1) if has_b else None
    bias2 = rnd(5, (B, Cout + 8), 0.1) if has_b2 else None
    res = rnd(6, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = F.conv2d(xin, w, bias, padding=k // 2)
    if has_b2:
        ref = ref + bias2[:, :Cout, None, None]
    if has_res:
        ref = ref + res
    ref = ref * scale
    got = G.conv2d(x1, w, bias, x2, bias2, res, scale)
    assert C.rel_l2(got, ref) < TOL


SPLITK_CASES = [
    (8, 4, 4, 256, 256, 256, 3, True, True, True, 0.70710678),    # level-6 up block (K = 4608 over 2 tiles)
    (8, 8, 8, 256, 0, 256, 3, True, False, False, 1.0),
    (2, 16, 16, 512, 0, 256, 1, True, False, True, 1.0),         # 1x1 shortcut on concat width
    (3, 6, 6, 128, 64, 192, 3, False, True, False, 1.0),         # ragged: M, K, N all off the tile grid
    (4, 32, 32, 256, 0, 256, 3, True, True, True, 0.70710678),    # B = 8 would take the Winograd kernel instead
]


STREAM_CASES = [
    # B, H, W, C1, C2, Cout, bias, bias2, res, scale      (1x1, > 2048 pixels, >= 256 blocks of 256 pixels x 128 channels)
    (8, 64, 64, 256, 256, 256, True, False, False, 1.0),          # shortcut on concat 512 -> 256: two channel blocks
    (4, 128, 128, 128, 128, 128, True, True, True, 0.70710678),   # per-sample bias and residual through the direct stores
    (1, 256, 256, 128, 0, 128, False, False, False, 1.0),         # one utterance at the top level
    (2, 128, 256, 32, 0, 128, True, False, True, 1.0),            # one chunk (K = 32)
    # bias-and-scale-only launches (the form the model's shortcuts take: no residual, no statistics), one / two channel blocks
    (8, 128, 192, 128, 128, 128, True, False, False, 0.70710678),
    (6, 64, 96, 128, 128, 256, True, False, False, 0.70710678),
]


@pytest.mark.parametrize("case", STREAM_CASES)
def test_conv1x1_stream(G, case):
    """The streaming fp32 1x1 kernel (every wave its own GEMM, operands straight into fragment registers) against torch's
    fp32 conv and against the flat kernel on the same input (the op entry runs it when given the weight-copy scratch)."""
    B, H, W, C1, C2, Cout, has_b, has_b2, has_res, scale = case
    x1 = rnd(1, (B, C1, H, W))
    x2 = rnd(2, (B, C2, H, W)) if C2 else None
    w = rnd(3, (Cout, C1 + C2, 1, 1), (1.0 / (C1 + C2)) ** 0.5)
    bias = rnd(4, (Cout,), 0.1) if has_b else None
    bias2 = rnd(5, (B, Cout + 8), 0.1) if has_b2 else None
    res = rnd(6, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = F.conv2d(xin.double(), w.double(), bias.double() if has_b else None).float()
    if has_b2:
        ref = ref + bias2[:, :Cout, None, None]
    if has_res:
        ref = ref + res
    ref = ref * scale
    got = G.conv2d(x1, w, bias, x2, bias2, res, scale, splitk=True)
    assert G.conv2d.last_split, "no weight-copy scratch requested: the streaming kernel did not take this shape"
    err = C.rel_l2(got, ref)
    flat = G.conv2d(x1, w, bias, x2, bias2, res, scale)
    print(f"1x1 stream rel-L2 {err:.2e}   flat kernel {C.rel_l2(flat, ref):.2e}")
    assert err < TOL and torch.equal(got, G.conv2d(x1, w, bias, x2, bias2, res, scale, splitk=True))


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_conv2d_splitk(G, case):
    B, H, W, C1, C2, Cout, k, has_b, has_b2, has_res, scale = case
    x1 = rnd(1, (B, C1, H, W))
    x2 = rnd(2, (B, C2, H, W)) if C2 else None
    w = rnd(3, (Cout, C1 + C2, k, k), (1.0 / ((C1 + C2) * k * k)) ** 0.5)
    bias = rnd(4, (Cout,), 0.1) if has_b else None
    bias2 = rnd(5, (B, Cout + 8), 0.1) if has_b2 else None
    res = rnd(6, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = F.conv2d(xin, w, bias, padding=k // 2)
    if has_b2:
        ref = ref + bias2[:, :Cout, None, None]
    if has_res:
        ref = ref + res
    ref = ref * scale
    got = G.conv2d(x1, w, bias, x2, bias2, res, scale, splitk=True)
    assert G.conv2d.last_split, "policy did not split this shape"
    assert C.rel_l2(got, ref) < TOL
    # split-K must agree with the single-pass kernel to rounding
    single = G.conv2d(x1, w, bias, x2, bias2, res, scale)
    assert C.rel_l2(got, single) < 5e-6


HALO_CASES = [
    # B, H, W, C1, C2, Cout, bias2, res, scale, silu   (grid = B*H*W/128 * ceil(Cout/128) tiles must be >= 256)
    (2, 128, 128, 128, 0, 128, True, True, 0.70710678, True),
    (2, 256, 64, 256, 128, 128, True, False, 1.0, True),      # concat 384: GroupNorm group straddles the sources
    (4, 64, 64, 128, 128, 256, False, True, 1.0, True),
    (1, 256, 128, 128, 0, 4, False, True, 1.0, True),         # pyramid head, 32-wide N tile
    (1, 128, 144, 32, 0, 160, False, False, 1.0, False),      # W = 9 tiles (not a power of two), no activation
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_fused_gn(G, case):
    """LDS-halo 3x3 kernel with GroupNorm(+SiLU) applied while staging == conv(act(GN(cat[x1,x2])))."""
    B, H, W, C1, C2, Cout, has_b2, has_res, scale, silu = case
    Cc = C1 + C2
    x1 = rnd(41, (B, C1, H, W)) * 1.5 + 0.3
    x2 = rnd(42, (B, C2, H, W)) * 0.7 - 0.2 if C2 else None
    g = 1.0 + rnd(43, (Cc,), 0.2)
    be = rnd(44, (Cc,), 0.2)
    w = rnd(45, (Cout, Cc, 3, 3), (1.0 / (Cc * 9)) ** 0.5)
    bias = rnd(46, (Cout,), 0.1)
    bias2 = rnd(47, (B, Cout + 4), 0.1) if has_b2 else None
    res = rnd(48, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    hn = F.group_norm(xin, min(Cc // 4, 32), g, be, eps=1e-6)
    if silu:
        hn = F.silu(hn)
    ref = F.conv2d(hn, w, bias, padding=1)
    if has_b2:
        ref = ref + bias2[:, :Cout, None, None]
    if has_res:
        ref = ref + res
    ref = ref * scale
    got = G.conv3x3_gn(x1, g, be, w, bias, x2, bias2, res, scale, silu)
    assert C.rel_l2(got, ref) < TOL
    # the same kernel without the fused normalisation (plain 3x3 through the halo path)
    plain = G.conv2d(hn, w, bias, None, bias2, res, scale)
    assert C.rel_l2(plain, ref) < TOL


WINO_CASES = [
    # B, H, W, C1, C2, Cout, bias2, res, scale, silu
    (2, 128, 128, 128, 0, 128, True, True, 0.70710678, True),
    (2, 256, 64, 256, 128, 128, True, False, 1.0, True),
    (4, 64, 64, 128, 128, 256, False, True, 1.0, True),
    (1, 128, 144, 32, 0, 192, False, False, 1.0, False),      # W = 9 tiles, three 64-wide N tiles
    (1, 256, 128, 64, 0, 64, False, True, 1.0, True),
    (8, 32, 32, 256, 256, 256, True, True, 0.70710678, True),   # one block per CU: preferred over split-K
    (2, 72, 256, 64, 32, 128, True, True, 1.0, True),           # H = 9 tiles (odd), 64+32 concat, strip-major walk
    (8, 16, 32, 256, 0, 256, True, True, 0.70710678, True),     # 128 blocks: F(4,3) split over 4 slices of chunks (above the small-M kernel's 2048 pixels)
    (1, 64, 64, 256, 256, 256, True, False, 1.0, True),         # single utterance: 128 blocks, split 4 x 4 chunks
    (8, 64, 128, 128, 0, 128, True, True, 0.70710678, True),    # 512 blocks of 128 channels: the F(4,3) 128-channel block form
    (2, 128, 128, 128, 128, 256, True, True, 1.0, True),        # same form, two channel blocks, concat input
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd(G, case):
    """F(4,3) Winograd kernel (with and without the fused GroupNorm + SiLU input stage) against the plain fp64 direct
    convolution; also against the direct HIP kernel on the same input."""
    form = "f43"
    B, H, W, C1, C2, Cout, has_b2, has_res, scale, silu = case
    Cc = C1 + C2
    x1 = rnd(41, (B, C1, H, W)) * 1.5 + 0.3
    x2 = rnd(42, (B, C2, H, W)) * 0.7 - 0.2 if C2 else None
    g = 1.0 + rnd(43, (Cc,), 0.2)
    be = rnd(44, (Cc,), 0.2)
    w = rnd(45, (Cout, Cc, 3, 3), (1.0 / (Cc * 9)) ** 0.5)
    bias = rnd(46, (Cout,), 0.1)
    bias2 = rnd(47, (B, Cout + 4), 0.1) if has_b2 else None
    res = rnd(48, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    hn = F.group_norm(xin, min(Cc // 4, 32), g, be, eps=1e-6)
    if silu:
        hn = F.silu(hn)

    def finish(t):
        if has_b2:
            t = t + bias2[:, :Cout, None, None]
        if has_res:
            t = t + res
        return t * scale

    ref = finish(F.conv2d(hn.double(), w.double(), bias.double(), padding=1).float())
    got = G.conv3x3_f43(x1, w, g, be, bias, x2, bias2, res, scale, silu)
    err = C.rel_l2(got, ref)
    direct = C.rel_l2(G.conv3x3_gn(x1, g, be, w, bias, x2, bias2, res, scale, silu), ref)
    print(f"winograd {form} rel-L2 {err:.2e}   direct kernel {direct:.2e}")
    assert err < TOL
    # plain conv (no normalisation), unnormalised input with a DC offset: the row differences d0 - d2 cancel it
    ref0 = finish(F.conv2d(xin.double(), w.double(), bias.double(), padding=1).float())
    got0 = G.conv3x3_f43(x1, w, None, None, bias, x2, bias2, res, scale)
    assert C.rel_l2(got0, ref0) < TOL


W2D_CASES = [
    # B, H, W, C1, C2, Cout, has_b2, has_res, scale, silu
    (8, 64, 64, 64, 0, 64, True, True, 0.70710678, True),        # 128 tiles x 1 channel block, two chunks
    (2, 128, 128, 128, 0, 128, True, True, 0.70710678, True),    # 128 tiles x 2 channel blocks, 4 chunks, tiles per block 1
    (8, 128, 128, 128, 0, 128, False, True, 1.0, True),          # 1024 block-tiles: two tiles per block (pipeline across tiles)
    (2, 256, 128, 64, 32, 128, True, False, 1.0, True),          # concat 64 + 32: odd chunk count (3), tiles per block 1
    (1, 256, 256, 128, 128, 256, True, True, 1.0, False),        # concat, 8 chunks, GroupNorm without SiLU, Cout 256
    (4, 64, 192, 32, 0, 64, False, False, 1.0, True),            # one chunk per tile, W = 12 tiles (row-major walk)
    (8, 32, 32, 256, 0, 256, True, True, 0.70710678, True),      # 32 tiles: the 32-channel-block form (256 blocks), Cout 256
    (8, 256, 256, 128, 0, 128, False, True, 1.0, True),          # the dominant launch: 16 tiles per block
]


@pytest.mark.parametrize("case", W2D_CASES)
def test_conv3x3_winograd_2d(G, case):
    """F(4,3) x F(2,3) two-dimensional Winograd kernel (with and without the fused GroupNorm + SiLU input stage) against
    the plain fp64 direct convolution, next to the 1-D F(4,3) kernel's error on the same input."""
    B, H, W, C1, C2, Cout, has_b2, has_res, scale, silu = case
    Cc = C1 + C2
    x1 = rnd(41, (B, C1, H, W)) * 1.5 + 0.3
    x2 = rnd(42, (B, C2, H, W)) * 0.7 - 0.2 if C2 else None
    g = 1.0 + rnd(43, (Cc,), 0.2)
    be = rnd(44, (Cc,), 0.2)
    w = rnd(45, (Cout, Cc, 3, 3), (1.0 / (Cc * 9)) ** 0.5)
    bias = rnd(46, (Cout,), 0.1)
    bias2 = rnd(47, (B, Cout + 4), 0.1) if has_b2 else None
    res = rnd(48, (B, Cout, H, W)) if has_res else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    hn = F.group_norm(xin, min(Cc // 4, 32), g, be, eps=1e-6)
    if silu:
        hn = F.silu(hn)

    def finish(t):
        if has_b2:
            t = t + bias2[:, :Cout, None, None]
        if has_res:
            t = t + res
        return t * scale

    ref = finish(F.conv2d(hn.double(), w.double(), bias.double(), padding=1).float())
    got = G.conv3x3_f43(x1, w, g, be, bias, x2, bias2, res, scale, silu, form="w2d")
    err = C.rel_l2(got, ref)
    err1 = C.rel_l2(G.conv3x3_f43(x1, w, g, be, bias, x2, bias2, res, scale, silu), ref)
    print(f"winograd 2-D rel-L2 {err:.2e}   1-D F(4,3) kernel {err1:.2e}")
    assert err < TOL
    again = G.conv3x3_f43(x1, w, g, be, bias, x2, bias2, res, scale, silu, form="w2d")
    assert torch.equal(got, again), "two runs differ"
    ref0 = finish(F.conv2d(xin.double(), w.double(), bias.double(), padding=1).float())
    got0 = G.conv3x3_f43(x1, w, None, None, bias, x2, bias2, res, scale, form="w2d")
    assert C.rel_l2(got0, ref0) < TOL


def test_conv3x3_winograd_rejects_uncovered_shapes(G):
    """The Winograd entry point refuses shapes its tiling does not cover (status SHAPE + message) instead of
    silently running another kernel: small image, Cout not a multiple of 64, H not a multiple of 8."""
    from flowmse_amd._lib import FlowseError
    for (B, Cc, H, W, Cout) in ((1, 64, 32, 32, 64), (2, 64, 128, 128, 96), (2, 64, 100, 128, 64)):
        x = rnd(61, (B, Cc, H, W))
        w = rnd(62, (Cout, Cc, 3, 3), 0.05)
        with pytest.raises(FlowseError, match="not covered"):
            G.conv3x3_f43(x, w)


@pytest.mark.parametrize("shape", [(2, 32, 16, 8), (1, 128, 32, 32), (2, 16, 8, 8), (1, 512, 4, 4),
                                   (1, 256, 64, 64)])
@pytest.mark.parametrize("silu", [True, False])
def test_group_norm(G, shape, silu):
    B, Cc, H, W = shape
    x = rnd(10, shape) * 2.0 + 0.7
    g = 1.0 + rnd(11, (Cc,), 0.2)
    b = rnd(12, (Cc,), 0.2)
    ref = F.group_norm(x, min(Cc // 4, 32), g, b, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    assert C.rel_l2(G.group_norm(x, g, b, silu=silu), ref) < TOL


@pytest.mark.parametrize("c1,c2", [(256, 128), (32, 16), (256, 256)])
def test_group_norm_concat_straddle(G, c1, c2):
    """GroupNorm over cat[h, skip]; for 256+128 group 21 straddles the two tensors (ncsnpp.py:337)."""
    x1 = rnd(13, (2, c1, 8, 8)) + 1.0
    x2 = rnd(14, (2, c2, 8, 8)) * 3.0 - 0.5
    Cc = c1 + c2
    g = 1.0 + rnd(15, (Cc,), 0.2)
    b = rnd(16, (Cc,), 0.2)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), min(Cc // 4, 32), g, b, eps=1e-6))
    assert C.rel_l2(G.group_norm(x1, g, b, x2=x2, silu=True), ref) < TOL


@pytest.mark.parametrize("offset", [50.0, 1000.0])
def test_group_norm_large_mean(G, offset):
    """|mean| >> std: raw sum / sum-of-squares statistics would cancel catastrophically in fp32; the pivoted
    mean / M2 partials must stay as accurate as a float64 evaluation of the same fp32 input allows."""
    x = rnd(17, (2, 64, 32, 32)) + offset
    g = 1.0 + rnd(18, (64,), 0.2)
    b = rnd(19, (64,), 0.2)
    ref = F.group_norm(x.double(), 16, g.double(), b.double(), eps=1e-6).float()
    got = G.group_norm(x, g, b, silu=False)
    # floor: the mean itself is stored in fp32 (half an ulp of 1000 is 3e-5 of one standard deviation)
    assert C.rel_l2(got, ref) < (2e-5 if offset < 100 else 6e-5)
    # same through the statistics fused into a conv epilogue + GroupNorm fused into the next conv's input:
    # y = conv_b(GN(conv_a(x0)))  with conv_a producing a large-mean tensor
    x0 = rnd(20, (2, 32, 128, 128))
    wa = rnd(21, (128, 32, 3, 3), (1.0 / (32 * 9)) ** 0.5)
    ba = torch.full((128,), float(offset))
    wb = rnd(22, (128, 128, 3, 3), (1.0 / (128 * 9)) ** 0.5)
    g2 = 1.0 + rnd(23, (128,), 0.2)
    b2 = rnd(24, (128,), 0.2)
    mid = F.conv2d(x0, wa, ba, padding=1)
    ref2 = F.conv2d(F.group_norm(mid.double(), 32, g2.double(), b2.double(), eps=1e-6).float(), wb, None, padding=1)
    got2 = G.conv3x3_gn(G.conv2d(x0, wa, ba), g2, b2, wb, None, silu=False)
    assert C.rel_l2(got2, ref2) < 5e-5


def test_fir_golden(G):
    g = C.gold("op_fir")
    x = torch.from_numpy(synth.normal(5, 1, (2, 8, 16, 32)))
    assert C.rel_l2(G.fir(x, True), g["up"]) < TOL
    assert C.rel_l2(G.fir(x, False), g["down"]) < TOL


@pytest.mark.parametrize("shape", [(1, 4, 2, 2), (2, 16, 6, 10), (1, 128, 32, 48)])
def test_fir_oracle(G, shape):
    from oracle import ncsnpp_oracle as O
    x = rnd(20, shape)
    assert C.rel_l2(G.fir(x, True), O.upsample_2d(x)) < TOL
    assert C.rel_l2(G.fir(x, False), O.downsample_2d(x)) < TOL


@pytest.mark.parametrize("up,down,pad,ksz", [(2, 1, (2, 1), 4), (1, 2, (1, 1), 4), (1, 1, (1, 1), 3), (2, 2, (3, 2), 4),
                                            (1, 1, (0, 0), 1)])
def test_upfirdn2d_nchw(G, up, down, pad, ksz):
    """Drop-in for the reference's native upfirdn2d ABI (asymmetric random kernel catches a missing flip)."""
    from oracle import ncsnpp_oracle as O
    x = rnd(21, (2, 3, 9, 14))
    k = rnd(22, (ksz, ksz)).numpy()
    ref = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
    got = G.upfirdn2d(x, k, up=up, down=down, pad=pad)
    assert got.shape == ref.shape
    assert C.rel_l2(got, ref) < TOL


@pytest.mark.parametrize("Cc,Lt", [(32, 64), (64, 256), (256, 16), (256, 48), (128, 100), (256, 256)])
def test_attention(G, Cc, Lt):
    q, k, v = rnd(30, (2, Cc, Lt)), rnd(31, (2, Cc, Lt)), rnd(32, (2, Cc, Lt))
    w = torch.einsum("bci,bcj->bij", q, k) * (int(Cc) ** (-0.5))
    w = F.softmax(w, dim=-1)
    ref = torch.einsum("bij,bcj->bci", w, v)
    assert C.rel_l2(G.attention(q, k, v), ref) < TOL


def test_attention_peaked(G):
    """Large logits: the online-softmax rescale path (running max jumps between key tiles)."""
    Cc, Lt = 64, 128
    q, k, v = rnd(33, (1, Cc, Lt)) * 4.0, rnd(34, (1, Cc, Lt)) * 4.0, rnd(35, (1, Cc, Lt))
    k[:, :, 97] = q[:, :, 5] * 3.0          # one key dominates query 5, in the last key tile
    w = F.softmax(torch.einsum("bci,bcj->bij", q, k) * (int(Cc) ** (-0.5)), dim=-1)
    ref = torch.einsum("bij,bcj->bci", w, v)
    assert C.rel_l2(G.attention(q, k, v), ref) < TOL


def test_gfp_golden(G):
    g = C.gold("op_gfp")
    W = torch.from_numpy(synth.synth_param("gfp.W", (16,)))
    got = G.gfp(torch.from_numpy(g["t"]), W)
    assert float((got - torch.from_numpy(g["out"])).abs().max()) < 2e-6


def test_stft_istft_fused_vs_reference_golden(G):
    """Fused STFT+compression / decompression+iSTFT kernels against the reference's SpecsDataModule outputs."""
    g = C.gold("op_spec")
    sig = torch.from_numpy(synth.normal(5, 8, (1, 4000), 0.1))
    Y, T = G.stft_compress(sig)
    assert T == g["fwd"].shape[-1] == 32 and Y.shape == (1, 1, 256, 64)
    assert C.rel_l2(Y[0, 0, :, :T], g["fwd"][0]) < 1e-5
    assert float(Y[0, 0, :, T:].abs().max()) == 0.0                     # pad_spec zeros
    back = G.istft_decompress(Y, T, 4000)
    assert C.rel_l2(back, g["istft"]) < 2e-5
    assert C.rel_l2(back, sig) < 2e-5                                   # the chain is an identity


@pytest.mark.parametrize("Ls,scale", [(16000, 1.0), (64000, 0.37), (12345, 2.5)])
def test_stft_istft_fused_vs_torch(G, Ls, scale):
    """Other lengths (incl. one that is not a multiple of the hop) and the waveform normalisation factor."""
    from flowmse_amd.data_module import SpecTransform
    st = SpecTransform()
    sig = rnd(60, (2, Ls), 0.2)
    ref = st.spec_fwd(st.stft(sig * scale))
    Y, T = G.stft_compress(sig, scale)
    assert T == ref.shape[-1]
    assert C.rel_l2(Y[:, 0, :, :T], ref) < 1e-5
    ref_back = st.istft(st.spec_back(ref), Ls) * (1.0 / scale)
    back = G.istft_decompress(Y, T, Ls, 1.0 / scale)
    assert C.rel_l2(back, ref_back) < 2e-5


# ---------------------------------------------------------------------------------------------------------------------
# 16-bit storage convolution (flowse_op_conv2d_16): fp32 tensors at the boundary, bf16 / half activations + weights and
# the 16-bit matrix cores inside.  Reference = torch fp32 conv of the UNROUNDED tensors, so the bound is the storage
# type's rounding (bf16 2^-9, half 2^-12 per operand) through a K-term sum -- asserted as a measured ceiling.
def _conv16(x, w, dt, bias=None, x2=None, res=None, gn=None, scale=1.0):
    import _gpu as G
    from flowmse_amd import _lib
    L = _lib.lib
    Cout, Cin, k, _ = w.shape
    taps = k * k
    a1 = G.nhwc(x)
    a2 = G.nhwc(x2) if x2 is not None else None
    B, H, W, C1 = a1.shape
    C2 = a2.shape[3] if a2 is not None else 0
    wp = w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous().cuda()
    bb = bias.cuda() if bias is not None else None
    rr = G.nhwc(res) if res is not None else None
    mean = scl = beta = None
    if gn is not None:
        mean, scl, beta = (t.contiguous().cuda() for t in gn)
    out = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    _lib.check(L.flowse_op_conv2d_16(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(rr),
                                     _lib.ptr(mean), _lib.ptr(scl), _lib.ptr(beta), 1, _lib.ptr(out), B, H, W, Cout, taps,
                                     float(scale), dt, _lib.ptr(scratch), scratch.numel(), G.stream()))
    torch.cuda.synchronize()
    return G.nchw(out)


@pytest.mark.parametrize("dt,bound", [(1, 4e-3), (2, 5e-4)])
@pytest.mark.parametrize("case", ["halo", "halo_gn_concat", "halo_32ch", "flat_small_splitk", "flat_1x1_concat", "flat_w8",
                                  "pc", "pc_gn_concat", "pc_gn_256out", "pc_ragged_items", "pc_64items", "head4_gn", "head4_plain"])
def test_conv2d_16bit_storage(case, dt, bound):
    """pc*: >= 64 (16 x 16 pixel tile, 128-channel block) items -> the persistent producer / consumer kernel
    (conv3x3_pc16_kernel); pc_64items: its smallest launch (64 blocks of one tile); pc_ragged_items: an item count that is no
    multiple of the 256 blocks (blocks with 1 and 2 tiles, tiles of several samples and both channel blocks in one block's
    stream).  head4*: the progressive-output heads C -> 4 (conv3x3_head4_16_kernel: 16-bit operands on v_mfma_f32_4x4x4, fp32
    residual and output).  halo*: image heights that are multiples of 8 but not of 16 -> the per-tap kernel (conv3x3_halo_bf16_kernel),
    which no power-of-two image reaches in the storage modes any more."""
    g = torch.Generator().manual_seed(3)
    shapes = {"halo": (6, 128, 0, 128, 24, 128, 3), "halo_gn_concat": (6, 128, 128, 128, 24, 128, 3),
              "halo_32ch": (6, 32, 0, 128, 24, 128, 3), "flat_small_splitk": (2, 256, 0, 256, 16, 16, 3),
              "pc_64items": (2, 128, 0, 128, 64, 128, 3),
              "flat_1x1_concat": (2, 256, 128, 128, 32, 64, 1), "flat_w8": (3, 256, 0, 256, 8, 8, 3),
              "pc": (1, 128, 0, 128, 256, 256, 3), "pc_gn_concat": (4, 64, 32, 128, 128, 128, 3),
              "pc_gn_256out": (2, 64, 0, 256, 128, 128, 3), "pc_ragged_items": (3, 32, 0, 256, 112, 128, 3),
              "head4_gn": (4, 128, 0, 4, 64, 64, 3), "head4_plain": (4, 64, 0, 4, 64, 64, 3)}
    B, C1, C2, Cout, H, W, k = shapes[case]
    C = C1 + C2
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, k, k, generator=g) / (C * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    gn, xin = None, x
    if case in ("halo_gn_concat", "pc_gn_concat", "pc_gn_256out", "head4_gn"):
        mean = 0.2 * torch.randn(B, C, generator=g)
        scl = 1 + 0.2 * torch.randn(B, C, generator=g)
        beta = 0.2 * torch.randn(C, generator=g)
        gn = (mean, scl, beta)
        xin = F.silu((x - mean[:, :, None, None]) * scl[:, :, None, None] + beta[None, :, None, None])
    ref = (F.conv2d(xin, w, bias, padding=k // 2) + res) * 0.7071
    # the producer / consumer kernel in both channel-block widths (128: NJ = 2, 64: NJ = 1), whatever the launch policy picks
    from flowmse_amd import _lib
    for blocks in ((0, 1) if case.startswith("pc") else (-1,)):
        _lib.check(_lib.lib.flowse_op_pc16_channel_blocks(blocks))
        try:
            got = _conv16(x[:, :C1].contiguous(), w, dt, bias, x[:, C1:].contiguous() if C2 else None, res, gn, 0.7071)
        finally:
            _lib.check(_lib.lib.flowse_op_pc16_channel_blocks(-1))
        err = float((got - ref).norm() / ref.norm())
        print(f"conv2d_16 {case} dt={dt} blocks={blocks}: rel-L2 vs fp32 torch {err:.3e}")
        assert err < bound


# ---------------------------------------------------------------------------------------------------------------------
# The 1x1 shortcut folded into Conv_1's launch (flowse_op_resblock_tail_16 -> ConvArgs::sc1 in conv3x3_pc16_kernel):
# out = (conv3x3(act(GroupNorm(h))) + b1 + conv1x1(cat[x1, x2]) + b2) / sqrt(2), layerspp.py:265-274.  Reference = torch
# fp32 of the unrounded tensors; ceilings as in test_conv2d_16bit_storage.  Shortcut step counts 4 / 8 / 12 / 16 cover
# every remainder of the three-entry fragment ring; one or two source tensors; blocks with several tiles.
TAIL_CASES = {"up_256to128": (2, 128, 128, 128, 128, 64, 128), "one_source_8": (1, 128, 256, 0, 128, 128, 128),
              "steps4_256out": (2, 256, 128, 0, 256, 64, 64), "steps12": (1, 128, 256, 128, 128, 128, 128),
              "steps16": (2, 256, 256, 256, 256, 64, 64), "ragged_items": (3, 64, 96, 32, 256, 112, 128),
              "no_gn": (1, 128, 128, 0, 128, 128, 128)}


@pytest.mark.parametrize("blocks", [0, 1])
@pytest.mark.parametrize("dt,bound", [(1, 4e-3), (2, 5e-4)])
@pytest.mark.parametrize("case", sorted(TAIL_CASES))
def test_resblock_tail_16bit_shortcut_fold(case, dt, bound, blocks):
    import _gpu as G
    from flowmse_amd import _lib
    L = _lib.lib
    B, C, X1, X2, Cout, H, W = TAIL_CASES[case]
    g = torch.Generator().manual_seed(11)
    h = torch.randn(B, C, H, W, generator=g)
    x = torch.randn(B, X1 + X2, H, W, generator=g)
    w1 = torch.randn(Cout, C, 3, 3, generator=g) / (C * 9) ** 0.5
    w2 = torch.randn(Cout, X1 + X2, 1, 1, generator=g) / (X1 + X2) ** 0.5
    b1, b2 = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    mean = scl = beta = None
    hin = h
    if case != "no_gn":
        mean = 0.2 * torch.randn(B, C, generator=g)
        scl = 1 + 0.2 * torch.randn(B, C, generator=g)
        beta = 0.2 * torch.randn(C, generator=g)
        hin = F.silu((h - mean[:, :, None, None]) * scl[:, :, None, None] + beta[None, :, None, None])
    ref = (F.conv2d(hin, w1, b1, padding=1) + F.conv2d(x, w2, b2)) * 0.7071
    hd = G.nhwc(h)
    x1d = G.nhwc(x[:, :X1].contiguous())
    x2d = G.nhwc(x[:, X1:].contiguous()) if X2 else None
    w1p = w1.permute(0, 2, 3, 1).reshape(Cout, 9, C).contiguous().cuda()
    w2p = w2.reshape(Cout, 1, X1 + X2).contiguous().cuda()
    md, sd, bd = ((t.contiguous().cuda() for t in (mean, scl, beta)) if mean is not None else (None, None, None))
    b1d, b2d = b1.cuda(), b2.cuda()                          # (named: a temporary's block would be handed to the next allocation)
    out = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    _lib.check(L.flowse_op_pc16_channel_blocks(blocks))      # 128- / 64-channel blocks (NJ = 2 / 1), whatever the policy picks
    try:
        _lib.check(L.flowse_op_resblock_tail_16(_lib.ptr(hd), C, _lib.ptr(md), _lib.ptr(sd), _lib.ptr(bd), 1, _lib.ptr(w1p),
                                                _lib.ptr(b1d), _lib.ptr(x1d), X1, _lib.ptr(x2d), X2, _lib.ptr(w2p),
                                                _lib.ptr(b2d), _lib.ptr(out), B, H, W, Cout, 0.7071, dt, _lib.ptr(scratch),
                                                scratch.numel(), G.stream()))
    finally:
        _lib.check(L.flowse_op_pc16_channel_blocks(-1))
    torch.cuda.synchronize()
    got = G.nchw(out)
    err = float((got - ref).norm() / ref.norm())
    print(f"resblock_tail_16 {case} dt={dt} blocks={blocks}: rel-L2 vs fp32 torch {err:.3e}")
    if err >= bound:                       # which term is off?
        r1, r2 = F.conv2d(hin, w1, b1, padding=1) * 0.7071, F.conv2d(x, w2, b2) * 0.7071
        print("  vs conv3x3 term alone", float((got - r1).norm() / r1.norm()), " vs shortcut alone", float((got - r2).norm() / r2.norm()))
    assert err < bound

