"""Debug aid: flowse_op_resblock_tail_16 term by term (which of conv3x3 / folded shortcut is off, and where)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
import _gpu as G
from flowmse_amd import _lib
L = _lib.lib

def run(h, x, X1, w1, b1, w2, b2, dt=1):
    B, C, H, W = h.shape
    Cout = w1.shape[0]
    X = x.shape[1]
    X2 = X - X1
    hd = G.nhwc(h); x1d = G.nhwc(x[:, :X1].contiguous()); x2d = G.nhwc(x[:, X1:].contiguous()) if X2 else None
    w1p = w1.permute(0, 2, 3, 1).reshape(Cout, 9, C).contiguous().cuda()
    w2p = w2.reshape(Cout, 1, X).contiguous().cuda()
    b1d, b2d = b1.cuda(), b2.cuda()
    out = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    _lib.check(L.flowse_op_resblock_tail_16(_lib.ptr(hd), C, None, None, None, 1, _lib.ptr(w1p), _lib.ptr(b1d), _lib.ptr(x1d), X1,
                                            _lib.ptr(x2d), X2, _lib.ptr(w2p), _lib.ptr(b2d), _lib.ptr(out), B, H, W, Cout, 1.0, dt,
                                            _lib.ptr(scratch), scratch.numel(), G.stream()))
    torch.cuda.synchronize()
    return G.nchw(out).cpu()

g = torch.Generator().manual_seed(1)
B, C, X, X1, Cout, H, W = 1, 128, 128, 128, 128, 128, 128
h = torch.randn(B, C, H, W, generator=g); x = torch.randn(B, X, H, W, generator=g)
w1 = torch.randn(Cout, C, 3, 3, generator=g) / (C * 9) ** 0.5; w2 = torch.randn(Cout, X, 1, 1, generator=g) / X ** 0.5
z1, z2 = torch.zeros_like(w1), torch.zeros_like(w2)
zb = torch.zeros(Cout)
for name, a1, a2 in (("conv only", w1, z2), ("shortcut only", z1, w2), ("both", w1, w2)):
    got = run(h, x, X1, a1, zb, a2, zb)
    ref = F.conv2d(h, a1, None, padding=1) + F.conv2d(x, a2)
    e = (got - ref)
    print(name, "rel", float(e.norm() / ref.norm()), "got norm", float(got.norm()), "ref norm", float(ref.norm()))
    if e.norm() / ref.norm() > 1e-2:
        # error by position inside the 16x16 tile, and by channel block of 32
        em = (e ** 2).mean(dim=(0, 1)).reshape(H // 16, 16, W // 16, 16).mean(dim=(0, 2))
        print(" err by tile row:", [round(float(v), 3) for v in em.mean(dim=1)])
        print(" err by tile col:", [round(float(v), 3) for v in em.mean(dim=0)])
        ec = (e ** 2).mean(dim=(0, 2, 3)).reshape(-1, 32).mean(dim=1)
        print(" err by channel block:", [round(float(v), 3) for v in ec])
        # is the shortcut output a shifted copy?  correlate with shifted refs
        if name == "shortcut only":
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    r = torch.roll(ref, (dy, dx), (2, 3))
                    print("  shift", dy, dx, float((got - r).norm() / r.norm()))
            # per input-channel group: zero all but one group of x
            for grp in range(X // 32):
                xx = torch.zeros_like(x); xx[:, grp * 32:(grp + 1) * 32] = x[:, grp * 32:(grp + 1) * 32]
                gg = run(h, xx, X1, z1, zb, w2, zb); rr = F.conv2d(xx, w2)
                print("  x group", grp, "rel", float((gg - rr).norm() / rr.norm()), "got/ref norm", float(gg.norm() / rr.norm()))
