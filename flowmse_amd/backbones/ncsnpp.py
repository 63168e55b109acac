"""NCSN++ vector-field backbone: the reference's constructor / call signature over the HIP library.

Drop-in for ``flowmse.backbones.ncsnpp.NCSNpp`` (reference flowmse/backbones/ncsnpp.py:36-404):

* same constructor keywords (``NCSNpp.__init__`` :45-67) and registry name ``"ncsnpp"`` (:36);
* same ``state_dict`` keys / shapes / ``parameters()`` order (checked against the reference in
  tests/test_host_logic.py), so ``load_state_dict`` of a flowmse checkpoint and the torch_ema
  ``shadow_params`` list apply unchanged;
* ``forward(x, time_cond)``: x complex64 ``[B,2,F,T]`` (channel 0 = x_t, channel 1 = y), time_cond float32
  ``[B]`` -> complex64 ``[B,1,F,T]`` (:247-404).

The nn.Module here is only a parameter container (PyTorch = tensor container / weight loader); every
arithmetic operation of the forward pass runs in hand-written HIP kernels behind the C ABI
(include/flowse_hip.h).  There is no PyTorch / CPU fallback: CPU tensors raise.
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from flowmse_amd import _lib
from .shared import BackboneRegistry
from .structure import create_handle, handle_param_table

_FIXED = dict(scale_by_sigma=True, nonlinearity="swish", resamp_with_conv=True, conditional=True, fir=True,
              fir_kernel="song", skip_rescale=True, resblock_type="biggan", progressive="output_skip",
              progressive_input="input_skip", progressive_combine="sum", embedding_type="fourier")


def _set_nested(root, dotted, param):
    """Register `param` under a dotted reference key, creating container modules on the way."""
    parts = dotted.split(".")
    mod = root
    for i, p in enumerate(parts[:-1]):
        if p.isdigit():
            mod = mod[int(p)]
        else:
            if not hasattr(mod, p):
                nxt = parts[i + 1]
                setattr(mod, p, nn.ModuleList() if nxt.isdigit() else nn.Module())
            mod = getattr(mod, p)
        if isinstance(mod, nn.ModuleList) and i + 1 < len(parts) - 1 and parts[i + 1].isdigit():
            while len(mod) <= int(parts[i + 1]):
                mod.append(nn.Module())
    mod.register_parameter(parts[-1], param)


class _ShapeOnly:
    """Stands in for a [B] time tensor in shape checks (the fused samplers build their times inside the library)."""

    def __init__(self, shape):
        self.shape = shape


@BackboneRegistry.register("ncsnpp")
class NCSNpp(nn.Module):
    @staticmethod
    def add_argparse_args(parser):
        return parser

    def __init__(self, scale_by_sigma=True, nonlinearity="swish", nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2),
                 num_res_blocks=2, attn_resolutions=(16,), resamp_with_conv=True, conditional=True, fir=True,
                 fir_kernel="song", skip_rescale=True, resblock_type="biggan", progressive="output_skip",
                 progressive_input="input_skip", progressive_combine="sum", init_scale=0., fourier_scale=16,
                 image_size=256, embedding_type="fourier", dropout=.0, **unused_kwargs):
        super().__init__()
        given = dict(scale_by_sigma=scale_by_sigma, nonlinearity=nonlinearity, resamp_with_conv=resamp_with_conv,
                     conditional=conditional, fir=fir, fir_kernel=fir_kernel, skip_rescale=skip_rescale,
                     resblock_type=str(resblock_type).lower(), progressive=str(progressive).lower(),
                     progressive_input=str(progressive_input).lower(),
                     progressive_combine=str(progressive_combine).lower(),
                     embedding_type=str(embedding_type).lower())
        for k, v in _FIXED.items():
            # scale_by_sigma / fir_kernel / resamp_with_conv are never read by the reference forward
            if k in ("scale_by_sigma", "fir_kernel", "resamp_with_conv"):
                continue
            if given[k] != v:
                raise NotImplementedError(
                    f"NCSNpp({k}={given[k]!r}): the HIP hot path implements the released FlowSE configuration "
                    f"({k}={v!r}) only")
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is a training feature; the sampler path is inference only")
        self.nf = nf
        self.ch_mult = tuple(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.num_resolutions = len(self.ch_mult)
        self.all_resolutions = [image_size // (2 ** i) for i in range(self.num_resolutions)]
        self.image_size = image_size
        self.init_scale = init_scale
        self.fourier_scale = fourier_scale
        self.scale_by_sigma = scale_by_sigma
        self.conditional = conditional
        self.skip_rescale = skip_rescale
        self.resblock_type = given["resblock_type"]
        self.progressive = given["progressive"]
        self.progressive_input = given["progressive_input"]
        self.embedding_type = given["embedding_type"]

        self._handle = create_handle(dict(nf=nf, ch_mult=self.ch_mult, num_res_blocks=num_res_blocks,
                                          attn_resolutions=self.attn_resolutions, image_size=image_size))
        names, shapes, offsets = handle_param_table(self._handle)
        self._param_names, self._param_shapes, self._param_offsets = names, shapes, offsets
        self._blob_numel = int(_lib.lib.flowse_model_blob_numel(self._handle))
        # output_layer first, then all_modules.* : the reference's registration order (ncsnpp.py:97,245)
        self.output_layer = nn.Module()
        self.all_modules = nn.ModuleList(
            [nn.Module() for _ in range(_lib.lib.flowse_model_num_modules(self._handle))])
        for name, shape in zip(names, shapes):
            p = nn.Parameter(torch.zeros(*shape), requires_grad=not name.endswith("all_modules.0.W"))
            _set_nested(self, name, p)
        self.reset_parameters()
        self._uploaded_versions = None
        self._uploaded_device = None
        self._frozen_depth = 0
        self._frozen_checked = False
        self.precision = "fp32"

    # ------------------------------------------------------------------ initialisation (reference rules)
    @torch.no_grad()
    def reset_parameters(self):
        """DDPM variance-scaling init as in the reference (layers.py:54-91: fan_avg, uniform; init_scale=0
        tensors get scale 1e-10), zero biases, unit GroupNorm, W ~ N(0, fourier_scale^2)."""
        for name, p in self.named_parameters():
            leaf = name.split(".")[-1]
            parent = name.split(".")[-2] if "." in name else ""
            if name == "all_modules.0.W":
                p.copy_(torch.randn(p.shape) * self.fourier_scale)
            elif leaf in ("bias", "b"):
                p.zero_()
            elif p.dim() == 1:                                    # GroupNorm weight
                p.fill_(1.0)
            else:
                scale = 1.0
                if parent == "Conv_1" or parent == "NIN_3":
                    scale = self.init_scale                       # layerspp.py:71,232
                elif parent.startswith("NIN"):
                    scale = 0.1                                   # layers.py:547
                elif p.dim() == 4 and p.shape[0] == 4 and p.shape[-1] == 3:
                    scale = self.init_scale                       # pyramid heads, ncsnpp.py:212,224
                scale = 1e-10 if scale == 0 else scale
                shape = p.shape
                rf = 1.0 if leaf == "W" else float(np.prod(shape)) / shape[0] / shape[1]
                fan_in, fan_out = shape[1] * rf, shape[0] * rf
                var = scale / ((fan_in + fan_out) / 2)
                p.copy_((torch.rand(*shape) * 2. - 1.) * np.sqrt(3 * var))

    # ------------------------------------------------------------------ weights -> library
    def _params_in_order(self):
        """Parameter objects in the library's table order.  The list is cached (no module-tree walk per call) and
        rebuilt whenever a parameter object OR a submodule on the way to one was replaced (``_apply`` with tensor
        swapping, ``register_parameter``, direct assignment, parametrize / prune, ``dnn.output_layer = ...``, swapping an
        entry of ``all_modules``): the cache is validated by identity -- every link ``parent._modules[name] is child``
        of the module tree and every ``owner._parameters[leaf] is p`` -- a dict lookup each, the same order of work as
        the version scan."""
        cached = self.__dict__.get("_plist")
        if cached is not None:
            plist, owners, links = cached
            if all(d.get(k) is m for d, k, m in links) and all(o[0].get(o[1]) is q for o, q in zip(owners, plist)):
                return plist
        mods = dict(self.named_modules())
        links = []
        for name, m in mods.items():
            if name:
                parent, _, key = name.rpartition(".")
                links.append((mods[parent]._modules, key, m))
        plist, owners = [], []
        for n in self._param_names:
            mod_name, _, leaf = n.rpartition(".")
            m = mods[mod_name]
            owners.append((m._parameters, leaf))
            plist.append(m._parameters[leaf])
        self.__dict__["_plist"] = (plist, owners, links)
        return plist

    def canonical_blob(self):
        """Flat float32 CPU tensor of all parameters in reference order / layout."""
        with torch.no_grad():
            return torch.cat([p.detach().to("cpu", torch.float32).reshape(-1) for p in self._params_in_order()])

    def upload_weights(self, device=None):
        """(Re)pack the current parameter values into the library's device-side layouts."""
        if device is not None:
            torch.cuda.set_device(device)
        blob = self.canonical_blob().contiguous()
        assert blob.numel() == self._blob_numel
        _lib.check(_lib.lib.flowse_model_load_weights(self._handle, C.c_void_p(blob.data_ptr()), blob.numel()))
        self._uploaded_versions = [p._version for p in self._params_in_order()]
        self._uploaded_device = torch.cuda.current_device()

    def mark_dirty(self):
        """Force a weight re-upload before the next call (set by load_state_dict / .to() / precision changes; call it
        yourself after mutating parameters through ``.data`` -- such writes do not bump ``Parameter._version``)."""
        self._uploaded_versions = None
        self._frozen_checked = False

    def _ensure_uploaded(self, device):
        # inside `weights_frozen()` the O(#parameters) identity + version scan runs once, on the first call; later calls
        # of the same loop only look at the dirty flag (mark_dirty / load_state_dict / _apply / set_precision reset it)
        if self._frozen_depth > 0 and self._frozen_checked and self._uploaded_versions is not None \
                and self._uploaded_device == device.index:
            return
        if self._uploaded_versions is None or self._uploaded_device != device.index or \
                self._uploaded_versions != [p._version for p in self._params_in_order()]:
            self.upload_weights(device)
        self._frozen_checked = self._frozen_depth > 0

    @contextlib.contextmanager
    def weights_frozen(self):
        """Promise that no parameter is mutated in place inside the block (a sampler loop: eval mode, ``no_grad``).
        The per-call scan of all 647 ``Parameter._version`` counters then runs once instead of once per network
        evaluation -- the per-NFE host cost of the plugin solver loop and of the black-box RK45 right-hand side.
        Explicit invalidation (``mark_dirty``, ``load_state_dict``, ``.to()``, ``set_precision``, the EMA swap of
        ``VFModel.eval``) is still honoured inside the block.  The solver loops of ``flowmse_amd.sampling`` (plugin
        ``update_fn`` loop, black-box RK45 right-hand side) enter this block for every VF_fn that offers it -- an
        ``update_fn`` that changes parameters in place during sampling must call ``mark_dirty()`` itself; a violation is
        detected on leaving the block (warning + re-upload on the next call), not inside it."""
        self._frozen_depth += 1
        self._frozen_checked = False
        try:
            yield self
        finally:
            self._frozen_depth -= 1
            # the promise is checked once on the way out: an in-place parameter update inside the block (test-time
            # adaptation in a plugin update_fn, ...) was NOT seen by the evaluations after the first one
            if self._frozen_depth == 0 and self._uploaded_versions is not None and \
                    self._uploaded_versions != [p._version for p in self._params_in_order()]:
                self.mark_dirty()
                import warnings
                warnings.warn("NCSNpp.weights_frozen(): parameters were modified in place inside the block; the network "
                              "evaluations after the first one of this block used the weights uploaded before the change "
                              "(they are re-uploaded on the next call). Do not mutate parameters inside a sampler loop, or "
                              "call dnn.mark_dirty() after each change.", RuntimeWarning, stacklevel=3)
            self._frozen_checked = False

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__.pop("_plist", None)
        self.mark_dirty()
        return r

    # ------------------------------------------------------------------ calls
    def _check_io(self, x, y, t):
        if not x.is_cuda:
            raise RuntimeError("flowmse_amd.NCSNpp runs on the MI355X HIP kernels only: move inputs to 'cuda' "
                               "(there is no CPU fallback)")
        if x.dtype != torch.complex64 or y.dtype != torch.complex64:
            raise TypeError("x and y must be complex64")
        if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected x, y of shape [B,1,F,T], got {tuple(x.shape)} / {tuple(y.shape)}")
        if t.shape != (x.shape[0],):
            raise ValueError(f"t must have shape [{x.shape[0]}], got {tuple(t.shape)}")

    def vf_call(self, x, t, y, mode):
        """mode 0: dnn(cat[x,y], t); mode 1: -dnn(...) (VFModel.forward)."""
        self._check_io(x, y, t)
        self._ensure_uploaded(x.device)
        x = x.contiguous()
        y = y.contiguous()
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        B, _, F, T = x.shape
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.flowse_vf_forward(self._handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(t), _lib.ptr(out),
                                                  B, F, T, mode, _lib.current_stream()))
        return out

    def forward(self, x, time_cond):
        if x.dim() != 4 or x.shape[1] != 2:
            raise ValueError(f"expected complex input [B,2,F,T], got {tuple(x.shape)}")
        return self.vf_call(x[:, 0:1], time_cond, x[:, 1:2], 0)

    TABLEAUS = {"euler": 0, "heun": 1, "rk4": 2}

    def rk_sample(self, x, y, ts, dts, tableau="euler"):
        """In-place N-step fixed-step integration on x: one C-ABI call for the whole loop (flowse_rk_sample in
        include/flowse_hip.h; 'euler' = the reference's solver, 'heun' / 'rk4' the plugin solvers)."""
        if x.dim() != 4:
            raise ValueError(f"expected x of shape [B,1,F,T], got {tuple(x.shape)}")
        self._check_io(x, y, _ShapeOnly((x.shape[0],)))
        if not (x.is_contiguous() and y.is_contiguous()):
            raise ValueError("x and y must be contiguous")
        self._ensure_uploaded(x.device)
        N = len(ts)
        ts_a = (C.c_float * N)(*[float(v) for v in ts])
        dts_a = (C.c_float * N)(*[float(v) for v in dts])
        B, _, F, T = x.shape
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.flowse_rk_sample(self._handle, _lib.ptr(x), _lib.ptr(y), ts_a, dts_a, N,
                                                 self.TABLEAUS[tableau], B, F, T, _lib.current_stream()))
        return x

    def euler_sample(self, x, y, ts, dts):
        """In-place N-step Euler integration on x (see flowse_euler_sample in include/flowse_hip.h)."""
        return self.rk_sample(x, y, ts, dts, "euler")

    def graph_launches(self):
        """hipGraph launches this handle has made so far (0 = every evaluation ran as plain launches)."""
        return int(_lib.lib.flowse_model_graph_launches(self._handle))

    def reserve(self, B, F, T):
        self._ensure_uploaded(torch.device("cuda", torch.cuda.current_device()))
        nbytes = C.c_int64()
        _lib.check(_lib.lib.flowse_model_reserve(self._handle, B, F, T, C.byref(nbytes)))
        return int(nbytes.value)

    PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}

    def set_precision(self, mode):
        """'fp32' (default, exact fp32 MFMA) | 'bf16x3' (split-bf16, fp32-class) | 'bf16' (BASELINE config 3)."""
        _lib.check(_lib.lib.flowse_model_set_precision(self._handle, self.PRECISIONS[mode]))
        self.precision = mode
        self.mark_dirty()
        return self

    def profile_begin(self, mode=0):
        """Bracket launches with HIP events (0: dominant conv kernel only, 1: every op)."""
        _lib.check(_lib.lib.flowse_profile_begin(self._handle, mode))

    def profile_end(self):
        import json
        buf = C.create_string_buffer(1 << 20)
        _lib.check(_lib.lib.flowse_profile_end(self._handle, buf, len(buf)))
        return json.loads(buf.value.decode())

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.__dict__.pop("_plist", None)
        self.mark_dirty()
        return r

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                _lib.lib.flowse_model_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
