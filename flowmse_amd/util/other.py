"""Shape helpers around the hot path (reference: flowmse/util/other.py:83-90)."""
import torch


def pad_spec(Y):
    """Zero-pad the frame axis up to a multiple of 64 (6 stride-2 levels in NCSN++)."""
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))


def si_sdr(s, s_hat):
    """Scale-invariant SDR in dB of the estimate ``s_hat`` against the target ``s`` (1-D numpy arrays);
    reference: flowmse/util/other.py:71-75."""
    import numpy as np
    target = (np.dot(s_hat, s) / np.dot(s, s)) * s
    return 10.0 * np.log10(np.sum(target ** 2) / np.sum((target - s_hat) ** 2))


def read_wav(path):
    """float32 tensor [1, samples] and the sample rate (scipy reader; torchaudio is optional here)."""
    import numpy as np
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    if data.ndim > 1:
        data = data[:, 0]
    return torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))[None], sr
