"""Checkpoint ingestion (BASELINE config[2]): a committed checkpoint in the REFERENCE's layout
(tests/golden/tiny_ckpt.ckpt, written by oracle/gen_golden.py with the unmodified reference's VFModel:
Lightning keys, state_dict 'dnn.*', hyper_parameters incl. the pickled class reference
flowmse.data_module.SpecsDataModule, ema = torch_ema state_dict) must load WITHOUT flowmse / pytorch_lightning /
torch_ema importable, and eval(no_ema=...) must select the weight set the reference selects (model.py:81-103,
evaluate.py:64-72)."""
import os
import sys

import pytest
import torch

import _cases as C
from flowmse_amd.util import synth

CKPT = os.path.join(C.GOLDEN, "tiny_ckpt.ckpt")
CKPT_CFG = dict(nf=16, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,), image_size=32)


def ckpt_inputs():
    B, F, T = 2, 32, 64
    xt = C.c64(synth.complex_normal(21, 1, (B, 1, F, T), 0.5))
    y = C.c64(synth.synth_spectrogram(7, B, F, T))
    return xt, y


def _load():
    from flowmse_amd.model import VFModel
    # exactly evaluate.py:64-67's call
    return VFModel.load_from_checkpoint(CKPT, base_dir="", batch_size=8, num_workers=4, kwargs=dict(gpu=False))


def test_reference_packages_are_absent_here():
    """The point of the fixture: none of the packages its pickle refers to may be importable in the test process."""
    for mod in ("flowmse", "pytorch_lightning", "torch_ema"):
        assert mod not in sys.modules
        with pytest.raises(ImportError):
            __import__(mod)
    with pytest.raises(ModuleNotFoundError):          # what a plain torch.load does with this file
        torch.load(CKPT, map_location="cpu", weights_only=False)


def test_load_reference_layout_checkpoint():
    from flowmse_amd.model import read_checkpoint
    raw = read_checkpoint(CKPT)
    assert {"state_dict", "hyper_parameters", "ema", "pytorch-lightning_version"} <= set(raw)
    cls = raw["hyper_parameters"]["data_module_cls"]
    assert cls.__module__ == "flowmse.data_module" and cls.__name__ == "SpecsDataModule"     # placeholder class
    m = _load()
    assert (m.dnn.nf, m.dnn.ch_mult, m.dnn.image_size) == (16, (1, 2), 32)
    assert m.ode.sigma_max == 0.487 and m.t_eps == 0.03 and m.data_module.n_fft == 510
    assert {"base_dir", "batch_size", "num_workers", "kwargs", "format", "normalize"} <= set(m.ignored_hparams)
    # raw weights = synth seed 0 under the reference's state_dict keys
    for n, p in m.dnn.named_parameters():
        assert torch.equal(p.detach(), torch.from_numpy(synth.synth_param(n, tuple(p.shape), 0))), n
    # EMA: torch_ema keeps the requires_grad parameters only (the Fourier frequencies W are frozen, layerspp.py:37)
    trainable = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    assert len(m._ema_shadow) == len(trainable) == len(list(m.parameters())) - 1
    for (n, p), s in zip(trainable, m._ema_shadow):
        assert torch.equal(s, torch.from_numpy(synth.synth_param(n[len("dnn."):], tuple(p.shape), 1))), n


def test_eval_swaps_ema_like_the_reference():
    m = _load()
    name, p = next((n, p) for n, p in m.dnn.named_parameters() if n.endswith("Conv_0.weight"))
    raw = torch.from_numpy(synth.synth_param(name, tuple(p.shape), 0))
    ema = torch.from_numpy(synth.synth_param(name, tuple(p.shape), 1))
    m.eval(no_ema=False)                      # evaluate.py:72
    assert torch.equal(p.detach(), ema)
    m.eval(no_ema=False)                      # idempotent: the backup is not overwritten by EMA weights
    m.train(True)
    assert torch.equal(p.detach(), raw)
    m.eval(no_ema=True)
    assert torch.equal(p.detach(), raw)


def test_oracle_on_checkpoint_weights_matches_reference():
    """Pins the oracle on this (2-level) configuration and both weight sets of the file."""
    from oracle import ncsnpp_oracle as O
    g = C.gold("ckpt_forward")
    xt, y = ckpt_inputs()
    t = torch.from_numpy(g["t"])
    m = _load()
    cfg = O.make_cfg(**CKPT_CFG)
    for key, no_ema in (("out_raw", True), ("out_ema", False)):
        m.eval(no_ema=no_ema)
        sd = {n: p.detach().clone() for n, p in m.dnn.named_parameters()}
        out = O.vf_forward(sd, cfg, xt, t, y)
        assert C.rel_l2(out, g[key]) < 1e-5, key
        m.train(True)
    assert C.rel_l2(g["out_raw"], g["out_ema"]) > 0.1       # the two weight sets are really different


@pytest.mark.gpu
def test_checkpoint_forward_on_gpu_matches_reference():
    """load -> eval(no_ema=False) -> cuda -> VF equals the reference's output with the EMA weights;
    eval(no_ema=True) equals its output with the state_dict weights."""
    g = C.gold("ckpt_forward")
    xt, y = ckpt_inputs()
    t = torch.from_numpy(g["t"]).cuda()
    m = _load()
    m.eval(no_ema=False)
    m.cuda()
    out = m(xt.cuda(), t, y.cuda())
    err_ema = C.rel_l2(out.cpu(), g["out_ema"])
    m.eval(no_ema=True)
    out = m(xt.cuda(), t, y.cuda())
    err_raw = C.rel_l2(out.cpu(), g["out_raw"])
    print("checkpoint forward rel-L2: ema", err_ema, "raw", err_raw)
    assert err_ema < 1e-4 and err_raw < 1e-4
