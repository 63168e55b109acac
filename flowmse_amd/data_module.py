"""Spectrogram transforms on either side of the sampler.

Mirror of the transform half of the reference's ``SpecsDataModule`` (flowmse/data_module.py:149-205): STFT with
n_fft 510 (256 bins), hop 128, periodic Hann, centred frames, and the magnitude compression
``c * |z|^e * exp(j arg z)`` (e = 0.5, c = 0.15) with its inverse.  These are the "next" rows of the hot-path scope
(SURVEY.md section 8(f)); they run as torch ops (plumbing) on the device of the signal.  Dataset / dataloader code
of the reference is training infrastructure and is not reproduced.
"""
import torch


def _window(kind, n_fft):
    base = torch.hann_window(n_fft, periodic=True)
    if kind == "hann":
        return base
    if kind == "sqrthann":
        return base.sqrt()
    raise NotImplementedError(f"Window type {kind} not implemented!")


class SpecTransform:
    def __init__(self, n_fft=510, hop_length=128, window="hann", spec_factor=0.15, spec_abs_exponent=0.5,
                 transform_type="exponent", **_unused):
        if transform_type not in ("exponent", "log", "none"):
            raise ValueError(f"unknown transform_type {transform_type!r}")
        self.n_fft, self.hop_length = n_fft, hop_length
        self.window = _window(window, n_fft)
        self._win_cache = {}
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.transform_type = transform_type

    def _win(self, ref):
        if ref.device not in self._win_cache:
            self._win_cache[ref.device] = self.window.to(ref.device)
        return self._win_cache[ref.device]

    # ---- magnitude warping, phase preserved ---------------------------------------------------
    @staticmethod
    def _remag(spec, new_mag):
        return torch.polar(new_mag, spec.angle())

    def spec_fwd(self, spec):
        kind, c = self.transform_type, self.spec_factor
        if kind == "none":
            return spec
        if kind == "log":
            return c * self._remag(spec, torch.log1p(spec.abs()))
        e = self.spec_abs_exponent
        return c * (spec if e == 1 else self._remag(spec, spec.abs().pow(e)))

    def spec_back(self, spec):
        kind = self.transform_type
        if kind == "none":
            return spec
        spec = spec / self.spec_factor
        if kind == "log":
            return self._remag(spec, torch.expm1(spec.abs()))
        e = self.spec_abs_exponent
        return spec if e == 1 else self._remag(spec, spec.abs().pow(1.0 / e))

    # ---- fused HIP path (GPU tensors, reference STFT geometry) ------------------------------------
    def fused_ok(self, ref):
        """The HIP kernels cover the released configuration: n_fft 510, hop 128, hann, 'exponent' transform."""
        return (ref.is_cuda and self.n_fft == 510 and self.hop_length == 128 and self.transform_type == "exponent"
                and bool(torch.equal(self.window, torch.hann_window(510, periodic=True))))

    def analyze(self, sig, scale=1.0, pad_multiple=64):
        """pad_spec(spec_fwd(stft(sig * scale)))[:, None] in one kernel: sig float32 [B, L] on 'cuda' ->
        complex64 [B, 1, 256, Tpad] (frames beyond L // 128 + 1 are the zero padding)."""
        from flowmse_amd import _lib
        sig = sig.contiguous().float()
        B, L = sig.shape
        T = L // self.hop_length + 1
        Tpad = ((T + pad_multiple - 1) // pad_multiple) * pad_multiple
        out = torch.empty(B, 1, 256, Tpad, dtype=torch.complex64, device=sig.device)
        with torch.cuda.device(sig.device):
            _lib.check(_lib.lib.flowse_stft_compress(_lib.ptr(sig), B, L, float(scale), _lib.ptr(out), T, Tpad,
                                                     float(self.spec_factor), float(self.spec_abs_exponent),
                                                     _lib.current_stream()))
        return out

    def synthesize(self, spec, length, scale=1.0):
        """istft(spec_back(spec), length) * scale in one kernel: spec complex64 [B, 1, 256, Tpad] -> [B, length].
        Like the reference (model.py:190-191 on the padded sample), ALL Tpad frames take part in the overlap-add:
        the frames past length // 128 + 1 still reach the last samples of the waveform."""
        from flowmse_amd import _lib
        spec = spec.contiguous()
        B, _, F, Tpad = spec.shape
        T = Tpad
        out = torch.empty(B, length, dtype=torch.float32, device=spec.device)
        with torch.cuda.device(spec.device):
            _lib.check(_lib.lib.flowse_istft_decompress(_lib.ptr(spec), B, T, Tpad, float(self.spec_factor),
                                                        float(self.spec_abs_exponent), _lib.ptr(out), length,
                                                        float(scale), _lib.current_stream()))
        return out

    # ---- STFT pair ------------------------------------------------------------------------------
    def _stft_args(self, ref):
        return dict(n_fft=self.n_fft, hop_length=self.hop_length, window=self._win(ref), center=True)

    def stft(self, sig):
        return torch.stft(sig, return_complex=True, **self._stft_args(sig))

    def istft(self, spec, length=None):
        return torch.istft(spec, length=length, **self._stft_args(spec))
