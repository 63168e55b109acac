#!/usr/bin/env python3
"""Golden-vector writer: runs the UNMODIFIED reference (seongq/flowmse) on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container where
/root/reference is mounted; its outputs (tests/golden/*.npz) are committed and
are what travels to the GPU box.  The reference itself never travels.

The reference ships no tests / fixtures / known-answer vectors (SURVEY.md
section 4), so parity is pinned by these vectors: (inputs, synthetic weights) are
regenerated deterministically from flowmse_amd.util.synth; only the reference's
OUTPUTS (plus small inputs) are stored.

Safety stubs:
  * torch.utils.cpp_extension.load -> no-op BEFORE importing flowmse.backbones,
    otherwise torch-ROCm would hipify the reference's .cu files in place;
  * pytorch_lightning / torch_ema / torchaudio / pesq / pystoi are absent here
    and are replaced by inert fakes (none of them is on the hot path);
  * sys.dont_write_bytecode so nothing is written under /root/reference.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz and the checkpoint fixture)
        python oracle/gen_golden.py ckpt       (only tests/golden/tiny_ckpt.ckpt + ckpt_forward.npz)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FLOWMSE_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn as nn
import torch.utils.cpp_extension as _cpp

torch.set_num_threads(8)


def _install_stubs():
    _cpp.load = lambda *a, **k: types.SimpleNamespace()       # never JIT / hipify

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    sys.modules["pytorch_lightning"] = pl

    ema = types.ModuleType("torch_ema")

    class ExponentialMovingAverage:
        def __init__(self, params, decay):
            self.collected_params = None

        def store(self, p):
            pass

        def copy_to(self, p):
            pass

        def restore(self, p):
            pass

        def to(self, *a, **k):
            pass

        def update(self, p):
            pass

        def state_dict(self):
            return {}

        def load_state_dict(self, d):
            pass

    ema.ExponentialMovingAverage = ExponentialMovingAverage
    sys.modules["torch_ema"] = ema

    ta = types.ModuleType("torchaudio")
    ta.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchaudio stub"))
    sys.modules["torchaudio"] = ta
    pq = types.ModuleType("pesq")
    pq.pesq = lambda *a, **k: float("nan")
    sys.modules["pesq"] = pq
    ps = types.ModuleType("pystoi")
    ps.stoi = lambda *a, **k: float("nan")
    sys.modules["pystoi"] = ps


_install_stubs()
sys.path.insert(0, REF)

from flowmse.backbones.ncsnpp import NCSNpp                     # noqa: E402
from flowmse.backbones.ncsnpp_utils import layerspp, up_or_down_sampling  # noqa: E402
from flowmse.odes import FLOWMATCHING                           # noqa: E402
from flowmse.sampling import get_white_box_solver               # noqa: E402
from flowmse.util.other import pad_spec                         # noqa: E402

from flowmse_amd.util import synth                              # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
TINY = dict(nf=16, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(16,), image_size=64)
FULL = dict()


def load_synth(module, prefix="", seed=0):
    """Overwrite every parameter of a reference module with synth_param(key)."""
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth.synth_param(prefix + k, tuple(v.shape), seed))
           for k, v in sd.items()}
    module.load_state_dict(new)
    return module


class VF(nn.Module):
    """VFModel.forward (model.py:164-170) without the Lightning constructor."""

    def __init__(self, dnn):
        super().__init__()
        self.dnn = dnn

    def forward(self, x, t, y):
        return -self.dnn(torch.cat([x, y], dim=1), t)


class FixedNoise:
    """ode whose prior_sampling uses an explicit z (odes.py:93-100 otherwise unchanged)."""

    def __init__(self, ode, z):
        self.ode, self.z = ode, z

    def prior_sampling(self, shape, y):
        std = self.ode._std(torch.ones((y.shape[0],)))
        return y + self.z * std[:, None, None, None], self.z


def c64(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KB)")


def param_table(model):
    names, shapes = [], []
    for k, v in model.state_dict().items():
        names.append(k)
        shapes.append(list(v.shape))
    order = [n for n, _ in model.named_parameters()]
    return names, shapes, order


CKPT_CFG = dict(nf=16, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,), image_size=32)


@torch.no_grad()
def make_checkpoint_fixture():
    """A checkpoint in the layout the reference's training writes (train.py:58-66 -> model.py:58,66,81-90):
    Lightning top-level keys, ``state_dict`` of the whole VFModel (``dnn.*``), ``hyper_parameters`` = every
    constructor argument INCLUDING the pickled class reference ``data_module_cls`` =
    flowmse.data_module.SpecsDataModule, and ``ema`` = torch_ema 0.3 ``ExponentialMovingAverage.state_dict()``
    (decay, num_updates, shadow_params over the requires_grad parameters in ``parameters()`` order,
    collected_params).  Weights are synthetic; the EMA shadow differs from the raw weights so that
    ``eval(no_ema=False)`` / ``eval(no_ema=True)`` are distinguishable.  Also stores the reference's outputs for
    both weight sets (tests/golden/ckpt_forward.npz)."""
    from flowmse.data_module import SpecsDataModule
    from flowmse.model import VFModel
    hp = dict(backbone="ncsnpp", ode="flowmatching", lr=1e-4, ema_decay=0.999, t_eps=0.03, T_rev=1.0,
              loss_abs_exponent=0.5, num_eval_files=10, loss_type="mse", data_module_cls=SpecsDataModule,
              sigma_min=0.0, sigma_max=0.487,
              base_dir="/data/WSJ0-CHiME3", format="default", batch_size=8, n_fft=510, hop_length=128,
              num_frames=256, window="hann", num_workers=4, dummy=False, spec_factor=0.15, spec_abs_exponent=0.5,
              normalize="noisy", transform_type="exponent", **CKPT_CFG)
    model = VFModel(**hp)
    load_synth(model.dnn, "", seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert all(k.startswith("dnn.") for k in sd)
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    shadow = [torch.from_numpy(synth.synth_param(n[len("dnn."):], tuple(p.shape), 1)) for n, p in trainable]
    ckpt = {"epoch": 7, "global_step": 4321, "pytorch-lightning_version": "1.6.5",
            "state_dict": sd, "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [],
            "hparams_name": "kwargs", "hyper_parameters": hp,
            "ema": {"decay": 0.999, "num_updates": 4321, "shadow_params": shadow, "collected_params": None}}
    path = os.path.join(OUT, "tiny_ckpt.ckpt")
    torch.save(ckpt, path)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KB)")

    B, Fq, T = 2, 32, 64
    xt = c64(synth.complex_normal(21, 1, (B, 1, Fq, T), 0.5))
    y = c64(synth.synth_spectrogram(7, B, Fq, T))
    t = torch.tensor([0.2725, 0.8])
    model.eval()                                     # EMA stub is inert: weights below are set by hand
    out_raw = model(xt, t, y)
    for (n, p), s in zip(trainable, shadow):         # what ema.copy_to(parameters()) does (model.py:97-99)
        p.copy_(s)
    out_ema = model(xt, t, y)
    save("ckpt_forward", t=t, out_raw=out_raw, out_ema=out_ema)


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    if "ckpt" in sys.argv[1:]:
        make_checkpoint_fixture()
        return
    act = nn.SiLU()

    # ---- parameter tables (key contract) --------------------------------------
    import json
    tables = {}
    for tag, cfg in (("tiny", TINY), ("full", FULL)):
        m = NCSNpp(**cfg)
        names, shapes, order = param_table(m)
        tables[tag] = dict(cfg={k: list(v) if isinstance(v, tuple) else v for k, v in cfg.items()},
                           names=names, shapes=shapes, parameters_order=order,
                           n_modules=len(m.all_modules),
                           n_params=int(sum(p.numel() for p in m.parameters())))
    with open(os.path.join(OUT, "param_tables.json"), "w") as f:
        json.dump(tables, f)
    print("full net params:", tables["full"]["n_params"], "modules:", tables["full"]["n_modules"])

    # ---- per-op fixtures --------------------------------------------------------
    x = torch.from_numpy(synth.normal(5, 1, (2, 8, 16, 32)))
    save("op_fir",
         up=up_or_down_sampling.upsample_2d(x, (1, 3, 3, 1), factor=2),
         down=up_or_down_sampling.downsample_2d(x, (1, 3, 3, 1), factor=2))

    gfp = load_synth(layerspp.GaussianFourierProjection(embedding_size=16, scale=16), "gfp.")
    tt = torch.tensor([0.03, 0.2725, 0.515, 1.0])
    save("op_gfp", t=tt, out=gfp(torch.log(tt)))

    nin = load_synth(layerspp.NIN(32, 48), "nin.")
    xn = torch.from_numpy(synth.normal(5, 2, (2, 32, 8, 4)))
    save("op_nin", out=nin(xn))

    for tag, (C, H, Wd) in (("attn_L64", (32, 8, 8)), ("attn_L256", (64, 16, 16)),
                            ("attn_L16", (256, 4, 4))):
        blk = load_synth(layerspp.AttnBlockpp(channels=C, skip_rescale=True, init_scale=0.), tag + ".")
        xa = torch.from_numpy(synth.normal(5, 3, (2, C, H, Wd)))
        save("op_" + tag, out=blk(xa))

    temb = torch.from_numpy(synth.normal(5, 4, (2, 64)))
    for tag, kw, shp in (("rb_plain", dict(in_ch=32, out_ch=32), (2, 32, 16, 8)),
                         ("rb_widen", dict(in_ch=48, out_ch=32), (2, 48, 16, 8)),
                         ("rb_down", dict(in_ch=32, out_ch=32, down=True), (2, 32, 16, 8)),
                         ("rb_up", dict(in_ch=32, out_ch=32, up=True), (2, 32, 8, 8)),
                         ("rb_gn12", dict(in_ch=384, out_ch=128), (1, 384, 8, 8))):
        blk = load_synth(layerspp.ResnetBlockBigGANpp(act=act, temb_dim=64, dropout=0.0, fir=True,
                                                      fir_kernel=[1, 3, 3, 1], skip_rescale=True,
                                                      init_scale=0., **kw), tag + ".")
        xr = torch.from_numpy(synth.normal(5, 5, shp))
        save("op_" + tag, out=blk(xr, temb[:shp[0]]))

    comb = load_synth(layerspp.Combine(dim1=4, dim2=32, method="sum"), "comb.")
    xp = torch.from_numpy(synth.normal(5, 6, (2, 4, 8, 8)))
    yh = torch.from_numpy(synth.normal(5, 7, (2, 32, 8, 8)))
    save("op_combine", out=comb(xp, yh))

    Yp = c64(synth.synth_spectrogram(0, 1, 256, 501))
    save("op_pad_spec", shape=np.array(pad_spec(Yp).shape), tail=pad_spec(Yp)[..., 501:].abs().sum())

    # ---- spectrogram transforms either side of the path (data_module.py:149-205) -------------
    from flowmse.data_module import SpecsDataModule
    dm = SpecsDataModule(base_dir="")
    sig = torch.from_numpy(synth.normal(5, 8, (1, 4000), 0.1))
    S = dm.stft(sig)
    Sf = dm.spec_fwd(S)
    save("op_spec", stft=S, fwd=Sf, back=dm.spec_back(Sf), istft=dm.istft(dm.spec_back(Sf), 4000))

    # ---- tiny net: forward + sampler --------------------------------------------
    tiny = load_synth(NCSNpp(**TINY)).eval()
    B, Fq, T = 2, 64, 64
    xt = c64(synth.complex_normal(11, 1, (B, 1, Fq, T), 0.5))
    y = c64(synth.synth_spectrogram(0, B, Fq, T))
    t = torch.tensor([0.03, 1.0])
    out = tiny(torch.cat([xt, y], dim=1), t)
    save("tiny_forward", t=t, out=out)
    print("tiny forward |out| rms:", float(out.abs().pow(2).mean().sqrt()))

    ode = FLOWMATCHING()
    vf = VF(tiny)
    z = c64(synth.synth_noise(0, B, Fq, T))
    res = {}
    for N in (1, 5):
        sampler = get_white_box_solver("euler", FixedNoise(ode, z), vf, Y=y, Y_prior=y,
                                       T_rev=1.0, t_eps=0.03, N=N)
        xs, ns = sampler()
        res[f"x_N{N}"] = xs
        assert ns == N
    # non-default grid
    sampler = get_white_box_solver("euler", FixedNoise(ode, z), vf, Y=y, Y_prior=y,
                                   T_rev=0.8, t_eps=0.05, N=3)
    res["x_N3_T08_e005"] = sampler()[0]
    res["timesteps_N5"] = torch.linspace(1.0, 0.03, 5)
    save("tiny_sampler", **res)

    # wide tiny net, non-square, T not a power of two (exercises W=192 -> 48 -> 12 ...)
    cfgw = dict(nf=32, ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(16,), image_size=64)
    wide = load_synth(NCSNpp(**cfgw)).eval()
    xt = c64(synth.complex_normal(12, 1, (1, 1, 64, 192), 0.5))
    yw = c64(synth.synth_spectrogram(3, 1, 64, 192))
    tw = torch.tensor([0.515])
    save("wide_forward", t=tw, out=wide(torch.cat([xt, yw], dim=1), tw))

    # ---- full net: forward at [1,2,256,64] ---------------------------------------
    full = load_synth(NCSNpp(**FULL)).eval()
    xt = c64(synth.complex_normal(13, 1, (1, 1, 256, 64), 0.5))
    yf = c64(synth.synth_spectrogram(5, 1, 256, 64))
    tf = torch.tensor([0.2725])
    outf = full(torch.cat([xt, yf], dim=1), tf)
    save("full_forward_T64", t=tf, out=outf)
    print("full forward |out| rms:", float(outf.abs().pow(2).mean().sqrt()))

    make_checkpoint_fixture()


if __name__ == "__main__":
    main()
