import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowmse_amd import _lib
L = _lib.lib
def run(dt, B, H, W, C, Cout, taps, gn):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g); w = torch.randn(Cout, C, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) / (C * taps) ** 0.5
    bias = torch.randn(Cout, generator=g)
    a = x.permute(0, 2, 3, 1).contiguous().cuda(); wp = w.permute(0, 2, 3, 1).reshape(Cout, taps, C).contiguous().cuda()
    out = torch.empty(B, H, W, Cout, device="cuda"); scratch = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    mean = scale = beta = None; xin = x
    if gn:
        mean = (0.1 * torch.randn(B, C, generator=g)); scale = 1 + 0.1 * torch.randn(B, C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
        xin = torch.nn.functional.silu((x - mean[:, :, None, None]) * scale[:, :, None, None] + beta[None, :, None, None])
        mean, scale, beta = mean.cuda(), scale.cuda(), beta.cuda()
    ref = torch.nn.functional.conv2d(xin, w, bias, padding=1 if taps == 9 else 0)
    _lib.check(L.flowse_op_conv2d_16(_lib.ptr(a), C, None, 0, _lib.ptr(wp), _lib.ptr(bias.cuda()), None, _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 1,
                                     _lib.ptr(out), B, H, W, Cout, taps, 1.0, dt, _lib.ptr(scratch), scratch.numel(), _lib.current_stream()))
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    print(f"dt={dt} {B}x{H}x{W} C{C}->{Cout} taps{taps} gn={gn}: rel {float((got-ref).norm()/ref.norm()):.3e}")
for dt in (1, 2):
    run(dt, 2, 64, 128, 128, 128, 9, False)
    run(dt, 2, 64, 128, 128, 128, 9, True)
    run(dt, 2, 64, 128, 32, 128, 9, False)
    run(dt, 2, 16, 16, 128, 128, 9, False)
    run(dt, 2, 64, 128, 256, 128, 1, False)
