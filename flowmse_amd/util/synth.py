"""Deterministic, library-version-independent synthetic data.

Everything here is a pure function of (seed, stream, element index) through a
splitmix64 hash evaluated with numpy uint64 arithmetic, so fixtures written in
one container reproduce bit-for-bit in another regardless of the torch / numpy
RNG implementation.  Used for

* synthetic NCSN++ weights (bench.py, smoke(), golden fixtures): every tensor
  is non-degenerate -- the reference's default initialisation leaves every
  ``Conv_1`` / ``NIN_3`` / pyramid conv at ~1e-10 (reference:
  flowmse/backbones/ncsnpp_utils/layers.py:88-91, ncsnpp.py:61), which would
  make random-init fixtures blind to half of the graph;
* synthetic spectrograms ``Y ~ CN(0, 0.1^2)`` and prior noise ``z ~ CN(0, 1)``
  (SURVEY.md section 8(d)).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, n):
    """n doubles in [0,1): hash of (seed, stream, i)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(base ^ _splitmix64(idx))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, stream, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, stream, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, stream, shape, std=1.0):
    """Box-Muller on two hashed uniforms (float64 math, rounded to float32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, 2 * stream + 0x5151, n)
    u2 = uniform01(seed, 2 * stream + 0x5152, n)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (std * r * np.cos(2.0 * np.pi * u2)).astype(np.float32).reshape(shape)


def complex_normal(seed, stream, shape, std=1.0):
    """CN(0, std^2): Re, Im ~ N(0, std^2 / 2) (matches torch.randn_like on complex64)."""
    s = std / np.sqrt(2.0)
    re = normal(seed, 2 * stream, shape, s)
    im = normal(seed, 2 * stream + 1, shape, s)
    return (re + 1j * im).astype(np.complex64)


def _stream_of(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def synth_param(name, shape, seed=0):
    """Synthetic value for the parameter called `name` (reference state_dict key)."""
    shape = tuple(int(s) for s in shape)
    st = _stream_of(name)
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if parent.startswith("GroupNorm") or (leaf in ("weight", "bias") and len(shape) == 1
                                           and "GN" in parent):
        if leaf == "weight":
            return 1.0 + uniform(seed, st, shape, -0.25, 0.25)
        return uniform(seed, st, shape, -0.15, 0.15)
    if leaf == "W" and len(shape) == 1:          # Gaussian Fourier frequencies
        return normal(seed, st, shape, 16.0)
    if leaf in ("bias", "b"):
        return uniform(seed, st, shape, -0.1, 0.1)
    if leaf == "W" and len(shape) == 2:          # NIN: [in, out]
        fan_in = shape[0]
    elif len(shape) == 4:                        # conv: [out, in, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
    elif len(shape) == 2:                        # linear: [out, in]
        fan_in = shape[1]
    else:
        fan_in = max(1, int(np.prod(shape)))
    a = float(np.sqrt(3.0 / fan_in))             # unit-gain uniform
    return uniform(seed, st, shape, -a, a)


def synth_spectrogram(index, B, F, T, std=0.1, seed=1234):
    """Y ~ CN(0, std^2), complex64 [B,1,F,T]; seed = 1234 + utterance index."""
    return complex_normal(seed + index, 7, (B, 1, F, T), std)


def synth_noise(index, B, F, T, seed=4321):
    """z ~ CN(0,1), complex64 [B,1,F,T]; seed = 4321 + utterance index."""
    return complex_normal(seed + index, 11, (B, 1, F, T), 1.0)
