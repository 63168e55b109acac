import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _cases as C, _gpu as G
from flowmse_amd.util import synth
from oracle import ncsnpp_oracle as O
def W(keys, prefix): return {k: torch.from_numpy(synth.synth_param(prefix + k, s)) for k, s in keys}
mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
tag, cin, cout, shp = ("halo", 128, 128, (2, 128, 64, 128))
wl = W(C.resblock_keys(cin, cout, 512, None), f"b16.{tag}.")
for zero1 in (False, True):
    w2 = dict(wl)
    if zero1:
        w2["Conv_1.weight"] = torch.zeros_like(w2["Conv_1.weight"])
    blk = G.Block("resnet", cin, cout, temb_dim=512).load(w2, precision=mode)
    x = torch.from_numpy(synth.normal(7, 11, shp)); temb = torch.from_numpy(synth.normal(7, 12, (shp[0], 512)))
    ref = O.resblock(O._W({f"all_modules.0.{k}": v for k, v in w2.items()}), 0, x, temb)
    got = blk(x, temb=temb)
    e = (got - ref)
    print(mode, "zero_conv1" if zero1 else "full", "rel", C.rel_l2(got, ref))
    print("  per-sample", [float(e[b].norm() / ref[b].norm()) for b in range(2)])
    pc = (e.pow(2).sum(dim=(0, 2, 3)).sqrt() / ref.pow(2).sum(dim=(0, 2, 3)).sqrt())
    print("  per-channel err min/max", float(pc.min()), float(pc.max()), "first8", [round(float(v), 3) for v in pc[:8]])
    pr = (e.pow(2).sum(dim=(0, 1, 3)).sqrt() / ref.pow(2).sum(dim=(0, 1, 3)).sqrt())
    print("  per-row err", [round(float(v), 3) for v in pr[:20]])
    pcx = (e.pow(2).sum(dim=(0, 1, 2)).sqrt() / ref.pow(2).sum(dim=(0, 1, 2)).sqrt())
    print("  per-col err", [round(float(v), 3) for v in pcx[:20]])
