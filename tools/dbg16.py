import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _cases as C, _gpu as G
from flowmse_amd.util import synth
from oracle import ncsnpp_oracle as O

def W(keys, prefix): return {k: torch.from_numpy(synth.synth_param(prefix + k, s)) for k, s in keys}
for mode in ("bf16", "fp16"):
    keys = [("Conv_0.weight", (32, 4, 1, 1)), ("Conv_0.bias", (32,))]
    wl = W(keys, "comb.")
    blk = G.Block("combine", 4, 32).load(wl, precision=mode)
    xp = torch.from_numpy(synth.normal(5, 6, (2, 4, 8, 8))); yh = torch.from_numpy(synth.normal(5, 7, (2, 32, 8, 8)))
    ref = torch.nn.functional.conv2d(xp, wl["Conv_0.weight"], wl["Conv_0.bias"]) + yh
    print(mode, "combine", C.rel_l2(blk(xp, yh), ref))
    wl = W(C.attn_keys(256), "b16.attn.")
    blk = G.Block("attn", 256, 256).load(wl, precision=mode)
    x = torch.from_numpy(synth.normal(7, 13, (2, 256, 16, 16)))
    ref = O.attnblock(O._W({f"all_modules.0.{k}": v for k, v in wl.items()}), 0, x)
    print(mode, "attn", C.rel_l2(blk(x), ref))
    for tag, cin, cout, shp in (("small_flat", 256, 256, (2, 256, 8, 8)), ("halo", 128, 128, (2, 128, 64, 128))):
        wl = W(C.resblock_keys(cin, cout, 512, None), f"b16.{tag}.")
        blk = G.Block("resnet", cin, cout, temb_dim=512).load(wl, precision=mode)
        x = torch.from_numpy(synth.normal(7, 11, shp)); temb = torch.from_numpy(synth.normal(7, 12, (shp[0], 512)))
        ref = O.resblock(O._W({f"all_modules.0.{k}": v for k, v in wl.items()}), 0, x, temb)
        got = blk(x, temb=temb)
        print(mode, tag, C.rel_l2(got, ref), float(got.abs().max()), float(ref.abs().max()), bool(torch.isfinite(got).all()))
