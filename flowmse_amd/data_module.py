"""Spectrogram transforms either side of the sampler (reference: flowmse/data_module.py:149-205).

STFT n_fft 510 / hop 128 / periodic hann / center=True, and the magnitude compression
``|z|^e exp(j angle z) * factor`` with its inverse.  These are the "next" rows of the hot-path scope table
(SURVEY.md section 8(f)); they run as plain torch ops (plumbing) on whatever device the signal lives on.
The dataset / dataloader side of SpecsDataModule is training infrastructure and out of scope.
"""
import torch


class SpecTransform:
    def __init__(self, n_fft=510, hop_length=128, window="hann", spec_factor=0.15, spec_abs_exponent=0.5,
                 transform_type="exponent", **ignored):
        self.n_fft = n_fft
        self.hop_length = hop_length
        if window == "hann":
            self.window = torch.hann_window(n_fft, periodic=True)
        elif window == "sqrthann":
            self.window = torch.sqrt(torch.hann_window(n_fft, periodic=True))
        else:
            raise NotImplementedError(f"Window type {window} not implemented!")
        self.windows = {}
        self.spec_factor = spec_factor
        self.spec_abs_exponent = spec_abs_exponent
        self.transform_type = transform_type

    def _get_window(self, x):
        w = self.windows.get(x.device)
        if w is None:
            w = self.window.to(x.device)
            self.windows[x.device] = w
        return w

    def spec_fwd(self, spec):
        if self.transform_type == "exponent":
            if self.spec_abs_exponent != 1:
                e = self.spec_abs_exponent
                spec = spec.abs() ** e * torch.exp(1j * spec.angle())
            spec = spec * self.spec_factor
        elif self.transform_type == "log":
            spec = torch.log(1 + spec.abs()) * torch.exp(1j * spec.angle())
            spec = spec * self.spec_factor
        return spec

    def spec_back(self, spec):
        if self.transform_type == "exponent":
            spec = spec / self.spec_factor
            if self.spec_abs_exponent != 1:
                e = self.spec_abs_exponent
                spec = spec.abs() ** (1 / e) * torch.exp(1j * spec.angle())
        elif self.transform_type == "log":
            spec = spec / self.spec_factor
            spec = (torch.exp(spec.abs()) - 1) * torch.exp(1j * spec.angle())
        return spec

    def stft(self, sig):
        return torch.stft(sig, n_fft=self.n_fft, hop_length=self.hop_length, window=self._get_window(sig),
                          center=True, return_complex=True)

    def istft(self, spec, length=None):
        return torch.istft(spec, n_fft=self.n_fft, hop_length=self.hop_length, window=self._get_window(spec),
                           center=True, length=length)
