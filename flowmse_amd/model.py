"""VFModel facade: the surface of ``flowmse.model.VFModel`` that ``evaluate.py`` touches, without Lightning.

Reference: flowmse/model.py:19-206.  What evaluate.py uses (evaluate.py:64-132): ``load_from_checkpoint``,
``eval(no_ema=False)`` (= swap the EMA shadow weights in, model.py:92-103), ``.cuda()``, ``.ode``, ``_stft``,
``_forward_transform``, ``to_audio`` and the model itself as ``VF_fn(x, t, y)`` (= ``-dnn(cat[x,y], t)``,
model.py:164-170).  Training (loss, optimizer, Lightning hooks, dataloaders) is out of scope.
"""
import pickle
import types
import warnings

import torch
import torch.nn as nn

from flowmse_amd.backbones import BackboneRegistry
from flowmse_amd.data_module import SpecTransform
from flowmse_amd.odes import ODERegistry


# ---------------------------------------------------------------------------------------------- checkpoint reader
# A reference checkpoint pickles class references from packages that are absent on an inference box:
# hyper_parameters['data_module_cls'] = flowmse.data_module.SpecsDataModule (train.py:58,66 -> save_hyperparameters,
# model.py:66), and Lightning may wrap hyper_parameters in its AttributeDict.  None of them carries information the
# sampler needs, so unknown globals from those packages unpickle to inert placeholders instead of failing.
_FOREIGN_PREFIXES = ("flowmse", "sgmse", "pytorch_lightning", "lightning", "lightning_fabric", "torch_ema",
                     "torchmetrics", "wandb", "omegaconf")


class _ForeignDict(dict):
    """Placeholder for dict-like foreign containers (e.g. pytorch_lightning AttributeDict)."""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)


def _foreign_placeholder(module, name):
    if name in ("AttributeDict", "DictConfig"):
        return _ForeignDict

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    return type(name, (), {"__module__": module, "__init__": __init__, "__setstate__": __setstate__,
                           "_flowse_foreign": True, "__reduce_ex__": object.__reduce_ex__})


class _CkptUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if module.split(".")[0] in _FOREIGN_PREFIXES:
                return _foreign_placeholder(module, name)
            raise


_ckpt_pickle = types.SimpleNamespace(__name__="flowmse_amd_ckpt_pickle", Unpickler=_CkptUnpickler, load=pickle.load,
                                     loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps,
                                     UnpicklingError=pickle.UnpicklingError, PickleError=pickle.PickleError)


def read_checkpoint(path, map_location="cpu"):
    """torch.load for a reference (Lightning) checkpoint that works without flowmse / pytorch_lightning /
    torch_ema installed: class references into those packages become placeholders."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_ckpt_pickle)


def _is_foreign(v):
    return getattr(v, "_flowse_foreign", False) or getattr(type(v), "_flowse_foreign", False)


# constructor keywords the facade understands (VFModel + NCSNpp + FLOWMATCHING + SpecTransform); every other
# hyper-parameter of a training run (dataset paths, loader settings, ...) is ignored with no effect
_KNOWN_HPARAMS = {
    "backbone", "ode", "lr", "ema_decay", "t_eps", "T_rev", "loss_abs_exponent", "num_eval_files", "loss_type",
    # NCSNpp (ncsnpp.py:45-67)
    "scale_by_sigma", "nonlinearity", "nf", "ch_mult", "num_res_blocks", "attn_resolutions", "resamp_with_conv",
    "conditional", "fir", "fir_kernel", "skip_rescale", "resblock_type", "progressive", "progressive_input",
    "progressive_combine", "init_scale", "fourier_scale", "image_size", "embedding_type", "dropout",
    # FLOWMATCHING (odes.py:66-68)
    "sigma_min", "sigma_max",
    # spectrogram transform half of SpecsDataModule (data_module.py:96-105)
    "n_fft", "hop_length", "window", "spec_factor", "spec_abs_exponent", "transform_type",
}


class VFModel(nn.Module):
    def __init__(self, backbone="ncsnpp", ode="flowmatching", lr=1e-4, ema_decay=0.999, t_eps=0.03, T_rev=1.0,
                 loss_abs_exponent=0.5, num_eval_files=10, loss_type="mse", data_module_cls=None, **kwargs):
        super().__init__()
        dnn_cls = BackboneRegistry.get_by_name(backbone)
        self.dnn = dnn_cls(**kwargs)
        ode_cls = ODERegistry.get_by_name(ode)
        self.ode = ode_cls(**kwargs)
        self.lr = lr
        self.ema_decay = ema_decay
        self.t_eps = t_eps
        self.T_rev = T_rev
        self.ode.T_rev = T_rev
        self.loss_type = loss_type
        self.num_eval_files = num_eval_files
        self.loss_abs_exponent = loss_abs_exponent
        self.data_module = SpecTransform(**kwargs) if data_module_cls is None else data_module_cls(**kwargs)
        # EMA state (torch_ema layout: list of tensors in self.parameters() order)
        self._ema_shadow = None
        self._ema_backup = None
        self._error_loading_ema = False

    # ------------------------------------------------------------------ checkpoint / EMA (model.py:81-106)
    def load_ema_shadow(self, shadow_params):
        params = [p for p in self.parameters()]
        trainable = [p for p in params if p.requires_grad]
        if len(shadow_params) == len(params):
            target = params
        elif len(shadow_params) == len(trainable):          # torch_ema keeps requires_grad params only
            target = trainable
        else:
            raise ValueError(f"EMA shadow has {len(shadow_params)} tensors, model has {len(params)} parameters "
                             f"({len(trainable)} trainable)")
        for s, p in zip(shadow_params, target):
            if tuple(s.shape) != tuple(p.shape):
                raise ValueError(f"EMA shadow shape {tuple(s.shape)} != parameter shape {tuple(p.shape)}")
        self._ema_shadow = [s.detach().clone().float() for s in shadow_params]
        self._ema_target = target

    @classmethod
    def load_from_checkpoint(cls, checkpoint_file, map_location="cpu", **overrides):
        """Parse a Lightning checkpoint of the reference without Lightning / torch_ema
        (layout: SURVEY.md section 5: hyper_parameters, state_dict['dnn.*'], ema['shadow_params'])."""
        ckpt = read_checkpoint(checkpoint_file, map_location)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)                      # evaluate.py:64-67 passes base_dir / batch_size / num_workers / kwargs
        hp.pop("data_module_cls", None)           # class reference (placeholder here); SpecTransform stands in
        dropped = sorted(k for k in hp if k not in _KNOWN_HPARAMS)
        hp = {k: v for k, v in hp.items() if k in _KNOWN_HPARAMS and not _is_foreign(v)}
        model = cls(**hp)
        model.ignored_hparams = dropped
        sd = {k[len("dnn."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("dnn.")}
        if not sd:
            raise KeyError("checkpoint state_dict has no 'dnn.*' entries (not a flowmse VFModel checkpoint?)")
        model.dnn.load_state_dict(sd)
        ema = ckpt.get("ema")
        if ema is not None and "shadow_params" in ema:
            model.load_ema_shadow(ema["shadow_params"])
        else:
            model._error_loading_ema = True
            warnings.warn("EMA state_dict not found in checkpoint!")
        return model

    def train(self, mode=True, no_ema=False):
        res = super().train(mode)
        if not self._error_loading_ema and self._ema_shadow is not None:
            with torch.no_grad():
                if mode is False and not no_ema:
                    if self._ema_backup is None:
                        self._ema_backup = [p.detach().clone() for p in self._ema_target]
                        for p, s in zip(self._ema_target, self._ema_shadow):
                            p.copy_(s.to(p.device))
                        self.dnn.mark_dirty()
                elif self._ema_backup is not None:
                    for p, b in zip(self._ema_target, self._ema_backup):
                        p.copy_(b)
                    self._ema_backup = None
                    self.dnn.mark_dirty()
        return res

    def eval(self, no_ema=False):
        return self.train(False, no_ema=no_ema)

    # ------------------------------------------------------------------ vector field
    def forward(self, x, t, y):
        """-dnn(cat([x, y], 1), t) (model.py:164-170); x, y complex64 [B,1,F,T] on 'cuda', t float32 [B]."""
        return self.dnn.vf_call(x, t, y, 1)

    def weights_frozen(self):
        """Context manager of the backbone: no in-place parameter updates inside (see NCSNpp.weights_frozen)."""
        return self.dnn.weights_frozen()

    def euler_sample_(self, x, y, timesteps, stepsizes):
        """Fused N-step Euler loop, in place on x (used by sampling.get_white_box_solver)."""
        return self.dnn.euler_sample(x, y, timesteps, stepsizes)

    def rk_sample_(self, x, y, timesteps, stepsizes, tableau):
        """Fused N-step fixed-step loop ('euler' | 'heun' | 'rk4'), in place on x: one library call, no host sync."""
        return self.dnn.rk_sample(x, y, timesteps, stepsizes, tableau)

    # ------------------------------------------------------------------ spectrogram helpers (model.py:190-203)
    def to_audio(self, spec, length=None):
        return self._istft(self._backward_transform(spec), length)

    def _forward_transform(self, spec):
        return self.data_module.spec_fwd(spec)

    def _backward_transform(self, spec):
        return self.data_module.spec_back(spec)

    def _stft(self, sig):
        return self.data_module.stft(sig)

    def _istft(self, spec, length=None):
        return self.data_module.istft(spec, length)
