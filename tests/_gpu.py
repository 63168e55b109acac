"""Helpers for the -m gpu tests: ctypes calls into libflowse_hip.so on torch 'cuda' tensors."""
import ctypes as C

import torch

from flowmse_amd import _lib

L = _lib.lib


def nhwc(x):
    """NCHW cpu/any -> NHWC contiguous cuda float32"""
    return x.permute(0, 2, 3, 1).contiguous().cuda().float()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def stream():
    return _lib.current_stream()


def conv2d(x1, w, bias=None, x2=None, bias2=None, res=None, scale=1.0, padding=None, splitk=False):
    """x*: NCHW cpu tensors; w: [Cout, Cin, k, k] (reference layout). Returns NCHW cpu."""
    Cout, Cin, k, _ = w.shape
    taps = k * k
    a1 = nhwc(x1)
    a2 = nhwc(x2) if x2 is not None else None
    B, H, W, C1 = a1.shape
    C2 = a2.shape[3] if a2 is not None else 0
    wp = w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous().cuda()
    bb = bias.contiguous().cuda() if bias is not None else None
    b2 = bias2.contiguous().cuda() if bias2 is not None else None
    rr = nhwc(res) if res is not None else None
    out = torch.empty(B, H, W, Cout, device="cuda")
    scratch = None
    if splitk:
        n = L.flowse_op_conv2d_scratch_floats(B, H, W, C1 + C2, Cout, taps)
        scratch = torch.empty(max(n, 1), device="cuda")
        conv2d.last_split = n > 0
    _lib.check(L.flowse_op_conv2d(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(b2),
                                  b2.shape[1] if b2 is not None else 0, _lib.ptr(rr), _lib.ptr(out), B, H, W, Cout,
                                  taps, float(scale), _lib.ptr(scratch), stream()))
    torch.cuda.synchronize()
    return nchw(out)


def group_norm(x1, gamma, beta, x2=None, silu=True, eps=1e-6):
    a1 = nhwc(x1)
    a2 = nhwc(x2) if x2 is not None else None
    B, H, W, C1 = a1.shape
    C2 = a2.shape[3] if a2 is not None else 0
    n = L.flowse_op_group_norm_scratch_floats(B, H * W, C1 + C2)
    scratch = torch.empty(n, device="cuda")
    out = torch.empty(B, H, W, C1 + C2, device="cuda")
    g, b = gamma.cuda().contiguous(), beta.cuda().contiguous()
    _lib.check(L.flowse_op_group_norm(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(g), _lib.ptr(b), eps, int(silu),
                                      _lib.ptr(out), B, H, W, _lib.ptr(scratch), stream()))
    torch.cuda.synchronize()
    return nchw(out)


def fir(x, up):
    a = nhwc(x)
    B, H, W, Cc = a.shape
    out = torch.empty((B, 2 * H, 2 * W, Cc) if up else (B, H // 2, W // 2, Cc), device="cuda")
    fn = L.flowse_op_fir_up if up else L.flowse_op_fir_down
    _lib.check(fn(_lib.ptr(a), _lib.ptr(out), B, H, W, Cc, stream()))
    torch.cuda.synchronize()
    return nchw(out)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    N, Cc, H, W = x.shape
    kh, kw = kernel.shape
    oh = (H * up + pad[0] + pad[1] - kh) // down + 1
    ow = (W * up + pad[0] + pad[1] - kw) // down + 1
    a = x.contiguous().cuda().float()
    k = torch.as_tensor(kernel, dtype=torch.float32).contiguous().cuda()
    out = torch.empty(N, Cc, oh, ow, device="cuda")
    _lib.check(L.flowse_upfirdn2d(_lib.ptr(a), _lib.ptr(k), N * Cc, H, W, kh, kw, up, up, down, down, pad[0], pad[1],
                                  pad[0], pad[1], _lib.ptr(out), oh, ow, stream()))
    torch.cuda.synchronize()
    return out.cpu()


def attention(q, k, v):
    """q,k,v: [B, C, L] cpu -> [B, C, L] cpu"""
    B, Cc, Lt = q.shape
    qkv = torch.cat([q.permute(0, 2, 1), k.permute(0, 2, 1), v.permute(0, 2, 1)], dim=2).contiguous().cuda()
    out = torch.empty(B, Lt, Cc, device="cuda")
    _lib.check(L.flowse_op_attention(_lib.ptr(qkv), _lib.ptr(out), B, Lt, Cc, stream()))
    torch.cuda.synchronize()
    return out.permute(0, 2, 1).contiguous().cpu()


def gfp(t, W):
    B, E = t.numel(), W.numel()
    td, Wd = t.cuda().contiguous(), W.cuda().contiguous()
    out = torch.empty(B, 2 * E, device="cuda")
    _lib.check(L.flowse_op_gfp(_lib.ptr(td), _lib.ptr(Wd), _lib.ptr(out), B, E, stream()))
    torch.cuda.synchronize()
    return out.cpu()


def conv3x3_gn(x1, gamma, beta, w, bias=None, x2=None, bias2=None, res=None, scale=1.0, silu=True, eps=1e-6):
    """(conv3x3(act(GN(cat[x1,x2]))) + bias + bias2 + res) * scale through the fused halo kernel."""
    Cout, Cin, k, _ = w.shape
    a1 = nhwc(x1)
    a2 = nhwc(x2) if x2 is not None else None
    B, H, W, C1 = a1.shape
    C2 = a2.shape[3] if a2 is not None else 0
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().cuda()
    bb = bias.contiguous().cuda() if bias is not None else None
    b2 = bias2.contiguous().cuda() if bias2 is not None else None
    rr = nhwc(res) if res is not None else None
    g, be = gamma.cuda().contiguous(), beta.cuda().contiguous()
    scratch = torch.empty(L.flowse_op_group_norm_scratch_floats(B, H * W, C1 + C2), device="cuda")
    out = torch.empty(B, H, W, Cout, device="cuda")
    _lib.check(L.flowse_op_conv3x3_gn(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(g), _lib.ptr(be), eps, int(silu),
                                      _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(b2), b2.shape[1] if b2 is not None else 0,
                                      _lib.ptr(rr), _lib.ptr(out), B, H, W, Cout, float(scale), _lib.ptr(scratch),
                                      stream()))
    torch.cuda.synchronize()
    return nchw(out)


def conv3x3_f43(x1, w, gamma=None, beta=None, bias=None, x2=None, bias2=None, res=None, scale=1.0, silu=True,
                eps=1e-6, form="f43"):
    """Same contract as conv3x3_gn (gamma None: no normalisation) through the F(4,3) Winograd kernel (form "f43") or the
    two-dimensional F(4,3) x F(2,3) kernel (form "w2d")."""
    Cout, Cin, k, _ = w.shape
    a1 = nhwc(x1)
    a2 = nhwc(x2) if x2 is not None else None
    B, H, W, C1 = a1.shape
    C2 = a2.shape[3] if a2 is not None else 0
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().cuda()
    bb = bias.contiguous().cuda() if bias is not None else None
    b2 = bias2.contiguous().cuda() if bias2 is not None else None
    rr = nhwc(res) if res is not None else None
    g = gamma.cuda().contiguous() if gamma is not None else None
    be = beta.cuda().contiguous() if beta is not None else None
    scratch = torch.empty(getattr(L, f"flowse_op_conv3x3_{form}_scratch_floats")(B, H, W, C1 + C2, Cout), device="cuda")
    out = torch.empty(B, H, W, Cout, device="cuda")
    fn = getattr(L, f"flowse_op_conv3x3_{form}")
    _lib.check(fn(_lib.ptr(a1), C1, _lib.ptr(a2), C2, _lib.ptr(g), _lib.ptr(be), eps, int(silu),
                                       _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(b2),
                                       b2.shape[1] if b2 is not None else 0, _lib.ptr(rr), _lib.ptr(out), B, H, W, Cout,
                                       float(scale), _lib.ptr(scratch), stream()))
    torch.cuda.synchronize()
    return nchw(out)


def stft_compress(sig, scale=1.0, factor=0.15, exponent=0.5, pad_multiple=64):
    B, Ls = sig.shape
    T = Ls // 128 + 1
    Tpad = ((T + pad_multiple - 1) // pad_multiple) * pad_multiple
    a = sig.contiguous().cuda().float()
    out = torch.empty(B, 1, 256, Tpad, dtype=torch.complex64, device="cuda")
    _lib.check(L.flowse_stft_compress(_lib.ptr(a), B, Ls, float(scale), _lib.ptr(out), T, Tpad, factor, exponent,
                                      stream()))
    torch.cuda.synchronize()
    return out.cpu(), T


def istft_decompress(spec, T, length, scale=1.0, factor=0.15, exponent=0.5):
    a = spec.contiguous().cuda()
    B, _, F, Tpad = a.shape
    out = torch.empty(B, length, device="cuda")
    _lib.check(L.flowse_istft_decompress(_lib.ptr(a), B, T, Tpad, factor, exponent, _lib.ptr(out), length,
                                         float(scale), stream()))
    torch.cuda.synchronize()
    return out.cpu()


class Block:
    """One reference module (ResnetBlockBigGANpp / AttnBlockpp / Combine) behind a single-module C-ABI handle."""
    KINDS = {"resnet": 0, "attn": 1, "combine": 2}

    def __init__(self, kind, in_ch, out_ch, up=False, down=False, temb_dim=64):
        from flowmse_amd.backbones.structure import handle_param_table
        self.h = C.c_void_p()
        _lib.check(L.flowse_block_create(self.KINDS[kind], in_ch, out_ch, int(up), int(down), temb_dim, C.byref(self.h)))
        self.names, self.shapes, self.offsets = handle_param_table(self.h)
        self.kind, self.out_ch, self.up, self.down = kind, out_ch, up, down

    def load(self, weights, precision="fp32"):
        """weights: {module-local reference key (e.g. 'Conv_0.weight'): tensor}; precision as NCSNpp.set_precision"""
        _lib.check(L.flowse_model_set_precision(self.h, {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}[precision]))
        blob = torch.zeros(int(L.flowse_model_blob_numel(self.h)))
        for n, shp, off in zip(self.names, self.shapes, self.offsets):
            w = weights[n[len("all_modules.0."):]].float()
            assert list(w.shape) == shp, (n, tuple(w.shape), shp)
            blob[off:off + w.numel()] = w.reshape(-1)
        _lib.check(L.flowse_model_load_weights(self.h, C.c_void_p(blob.data_ptr()), blob.numel()))
        return self

    def __call__(self, x1, x2=None, temb=None):
        """NCHW cpu tensors in, NCHW cpu out.  temb: raw time embedding [B, temb_dim] (SiLU applied here, as plumbing)."""
        a1 = nhwc(x1)
        a2 = nhwc(x2) if x2 is not None else None
        B, H, W, C1 = a1.shape
        ta = torch.nn.functional.silu(temb.float()).contiguous().cuda() if temb is not None else None
        Ho, Wo = (2 * H, 2 * W) if self.up else (H // 2, W // 2) if self.down else (H, W)
        out = torch.empty(B, Ho, Wo, self.out_ch, device="cuda")
        _lib.check(L.flowse_block_forward(self.h, _lib.ptr(a1), C1, _lib.ptr(a2), _lib.ptr(ta), _lib.ptr(out), B, H, W,
                                          stream()))
        torch.cuda.synchronize()
        return nchw(out)

    def __del__(self):
        try:
            if self.h:
                L.flowse_model_destroy(self.h)
                self.h = None
        except Exception:
            pass
