"""Shape helpers around the hot path (reference: flowmse/util/other.py:83-90)."""
import torch


def pad_spec(Y):
    """Zero-pad the frame axis up to a multiple of 64 (6 stride-2 levels in NCSN++)."""
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))
