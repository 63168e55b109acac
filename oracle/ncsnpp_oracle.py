"""CPU oracle: functional restatement of the NCSN++ vector-field network.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU fp32 ops on
a flat ``{reference state_dict key: tensor}`` weight dict; no nn.Module, no
reference import.  Each function cites the reference lines it follows.

Reference: /root/reference/flowmse/backbones/ncsnpp.py (constructor :45-245,
forward :247-404) and ncsnpp_utils/{layerspp,layers,up_or_down_sampling}.py,
op/upfirdn2d.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                   attn_resolutions=(16,), image_size=256, fourier_scale=16)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    cfg["ch_mult"] = tuple(cfg["ch_mult"])
    cfg["attn_resolutions"] = tuple(cfg["attn_resolutions"])
    return cfg


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def fir_kernel_2d():
    """_setup_kernel([1,3,3,1]) -- up_or_down_sampling.py:181-188."""
    k = np.asarray([1, 3, 3, 1], dtype=np.float32)
    k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d_native -- op/upfirdn2d.py:159-200 (pad >= 0 only, as used here).

    x: [N,C,H,W]; zero-insert by `up`, pad (pad0 before, pad1 after) on both
    axes, correlate with the flipped kernel, decimate by `down`.
    """
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    xx = x.reshape(n * c, 1, in_h, in_w)
    if up > 1:
        z = torch.zeros(n * c, 1, in_h * up, in_w * up, dtype=x.dtype)
        z[:, :, ::up, ::up] = xx
        xx = z
    xx = F.pad(xx, [p0, p1, p0, p1])
    w = torch.flip(torch.as_tensor(kernel, dtype=x.dtype), [0, 1]).view(1, 1, kh, kw)
    out = F.conv2d(xx, w)
    out = out[:, :, ::down, ::down]
    return out.reshape(n, c, out.shape[-2], out.shape[-1])


def upsample_2d(x, factor=2):
    """up_or_down_sampling.py:195-224 with k=[1,3,3,1], gain=1."""
    k = fir_kernel_2d() * (factor ** 2)
    p = k.shape[0] - factor
    return upfirdn2d(x, k, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, factor=2):
    """up_or_down_sampling.py:227-257 with k=[1,3,3,1], gain=1."""
    k = fir_kernel_2d()
    p = k.shape[0] - factor
    return upfirdn2d(x, k, down=factor, pad=((p + 1) // 2, p // 2))


def group_norm(x, w, b):
    """nn.GroupNorm(min(C//4,32), C, eps=1e-6) -- layerspp.py:219,231,67."""
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), w, b, eps=1e-6)


def nin(x, W, b):
    """NIN -- layers.py:546-555: y[b,o,h,w] = sum_i x[b,i,h,w] W[i,o] + b[o]."""
    y = torch.einsum("bihw,io->bohw", x, W)
    return y + b[None, :, None, None]


class _W:
    """Accessor for `all_modules.{i}.{name}` keys."""

    def __init__(self, weights, prefix=""):
        self.w = {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
                  for k, v in weights.items()}
        self.prefix = prefix

    def __call__(self, idx, name):
        return self.w[f"{self.prefix}all_modules.{idx}.{name}"]

    def top(self, name):
        return self.w[f"{self.prefix}{name}"]

    def has(self, idx, name):
        return f"{self.prefix}all_modules.{idx}.{name}" in self.w


def resblock(W, i, x, temb, up=False, down=False):
    """ResnetBlockBigGANpp.forward -- layerspp.py:245-274."""
    h = F.silu(group_norm(x, W(i, "GroupNorm_0.weight"), W(i, "GroupNorm_0.bias")))
    if up:
        h = upsample_2d(h)
        x = upsample_2d(x)
    elif down:
        h = downsample_2d(h)
        x = downsample_2d(x)
    h = F.conv2d(h, W(i, "Conv_0.weight"), W(i, "Conv_0.bias"), padding=1)
    h = h + F.linear(F.silu(temb), W(i, "Dense_0.weight"), W(i, "Dense_0.bias"))[:, :, None, None]
    h = F.silu(group_norm(h, W(i, "GroupNorm_1.weight"), W(i, "GroupNorm_1.bias")))
    h = F.conv2d(h, W(i, "Conv_1.weight"), W(i, "Conv_1.bias"), padding=1)
    if W.has(i, "Conv_2.weight"):
        x = F.conv2d(x, W(i, "Conv_2.weight"), W(i, "Conv_2.bias"))
    return (x + h) / np.sqrt(2.0)


def attnblock(W, i, x):
    """AttnBlockpp.forward -- layerspp.py:75-91 (skip_rescale=True)."""
    B, C, H, Wd = x.shape
    h = group_norm(x, W(i, "GroupNorm_0.weight"), W(i, "GroupNorm_0.bias"))
    q = nin(h, W(i, "NIN_0.W"), W(i, "NIN_0.b"))
    k = nin(h, W(i, "NIN_1.W"), W(i, "NIN_1.b"))
    v = nin(h, W(i, "NIN_2.W"), W(i, "NIN_2.b"))
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = torch.reshape(w, (B, H, Wd, H * Wd))
    w = F.softmax(w, dim=-1)
    w = torch.reshape(w, (B, H, Wd, H, Wd))
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = nin(h, W(i, "NIN_3.W"), W(i, "NIN_3.b"))
    return (x + h) / np.sqrt(2.0)


def time_embedding(W, t):
    """GaussianFourierProjection (layerspp.py:39-41) + 2 Linear (ncsnpp.py:256-275)."""
    x = torch.log(t)
    x_proj = x[:, None] * W(0, "W")[None, :] * 2 * np.pi
    temb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    temb = F.linear(temb, W(1, "weight"), W(1, "bias"))
    temb = F.linear(F.silu(temb), W(2, "weight"), W(2, "bias"))
    return temb


# ----------------------------------------------------------------------------
# full forward
# ----------------------------------------------------------------------------
def ncsnpp_forward(weights, cfg, x, time_cond, prefix=""):
    """NCSNpp.forward -- ncsnpp.py:247-404.

    x: complex64 [B,2,F,T] (channel 0 = state x_t, channel 1 = noisy y),
    time_cond: float32 [B].  Returns complex64 [B,1,F,T].
    """
    W = _W(weights, prefix)
    nres = len(cfg["ch_mult"])
    nrb = cfg["num_res_blocks"]
    attn_res = cfg["attn_resolutions"]
    m = 0
    # ncsnpp.py:252-254 feature pack
    x = torch.cat((x[:, [0]].real, x[:, [0]].imag, x[:, [1]].real, x[:, [1]].imag), dim=1)
    temb = time_embedding(W, time_cond)
    m = 3
    input_pyramid = x
    hs = [F.conv2d(x, W(m, "weight"), W(m, "bias"), padding=1)]
    m += 1
    # down path, ncsnpp.py:289-322
    for i_level in range(nres):
        for _ in range(nrb):
            h = resblock(W, m, hs[-1], temb)
            m += 1
            if h.shape[-2] in attn_res:
                h = attnblock(W, m, h)
                m += 1
            hs.append(h)
        if i_level != nres - 1:
            h = resblock(W, m, hs[-1], temb, down=True)
            m += 1
            input_pyramid = downsample_2d(input_pyramid)
            # Combine(method='sum') -- layerspp.py:52-59
            h = F.conv2d(input_pyramid, W(m, "Conv_0.weight"), W(m, "Conv_0.bias")) + h
            m += 1
            hs.append(h)
    # middle, ncsnpp.py:324-330
    h = hs[-1]
    h = resblock(W, m, h, temb); m += 1
    h = attnblock(W, m, h); m += 1
    h = resblock(W, m, h, temb); m += 1
    pyramid = None
    # up path, ncsnpp.py:335-385
    for i_level in reversed(range(nres)):
        for _ in range(nrb + 1):
            h = resblock(W, m, torch.cat([h, hs.pop()], dim=1), temb)
            m += 1
        if h.shape[-2] in attn_res:
            h = attnblock(W, m, h)
            m += 1
        ph = F.silu(group_norm(h, W(m, "weight"), W(m, "bias")))
        m += 1
        ph = F.conv2d(ph, W(m, "weight"), W(m, "bias"), padding=1)
        m += 1
        pyramid = ph if i_level == nres - 1 else upsample_2d(pyramid) + ph
        if i_level != 0:
            h = resblock(W, m, h, temb, up=True)
            m += 1
    assert not hs
    n_modules = 1 + max(int(k[len(prefix):].split(".")[1]) for k in W.w
                        if k.startswith(prefix + "all_modules."))
    assert m == n_modules, (m, n_modules)
    # head, ncsnpp.py:398-403
    h = pyramid / time_cond[:, None, None, None]
    h = F.conv2d(h, W.top("output_layer.weight"), W.top("output_layer.bias"))
    h = h.permute(0, 2, 3, 1).contiguous()
    return torch.view_as_complex(h)[:, None, :, :]


def vf_forward(weights, cfg, x, t, y, prefix=""):
    """VFModel.forward -- model.py:164-170: -dnn(cat([x,y],1), t)."""
    return -ncsnpp_forward(weights, cfg, torch.cat([x, y], dim=1), t, prefix)
