"""Validation hook: enhance a few validation files with the fast sampler and score them.

Same call as the reference's ``evaluate_model(model, num_eval_files, inference_N)`` (flowmse/util/inference.py:15-71,
called from ``VFModel.validation_step``, model.py:139-150): files are picked uniformly over
``model.data_module.valid_set.{clean,noisy}_files``, each is normalised by its peak, transformed, padded, run through
the white-box Euler solver (one fused ``flowse_euler_sample`` call when ``model`` is the HIP-backed VFModel) and
resynthesised; returns the means ``(pesq, si_sdr, estoi)``.  PESQ / ESTOI need the optional ``pesq`` / ``pystoi``
packages and are NaN without them (the reference imports them unconditionally).
"""
import torch

from flowmse_amd.util.other import read_wav, si_sdr

sr = 16000
N = 5


def evaluate_model(model, num_eval_files, inference_N=N, odesolver="euler", VF_fn=None):
    from flowmse_amd.evaluate import enhance_waveform       # late: evaluate.py imports the sampling package
    try:
        from pesq import pesq
    except ImportError:
        pesq = None
    try:
        from pystoi import stoi
    except ImportError:
        stoi = None
    T_rev, t_eps = model.T_rev, model.t_eps
    model.ode.T_rev = T_rev
    valid = model.data_module.valid_set
    picks = torch.linspace(0, len(valid.clean_files) - 1, num_eval_files, dtype=torch.int).tolist()
    device = torch.device("cpu") if VF_fn is not None else next(model.parameters()).device
    tot_pesq = tot_sdr = tot_estoi = 0.0
    for i in picks:
        x, _ = read_wav(valid.clean_files[i])
        y, _ = read_wav(valid.noisy_files[i])
        x_hat = enhance_waveform(model, y.to(device), N=inference_N, T_rev=T_rev, t_eps=t_eps, odesolver=odesolver,
                                 VF_fn=VF_fn, device=device)
        x = x.squeeze().numpy()
        tot_sdr += si_sdr(x, x_hat)
        try:
            tot_pesq += pesq(sr, x, x_hat, "wb") if pesq else float("nan")
        except Exception:                                   # pesq raises on utterances without speech
            tot_pesq += float("nan")
        tot_estoi += stoi(x, x_hat, sr, extended=True) if stoi else float("nan")
    return tot_pesq / num_eval_files, tot_sdr / num_eval_files, tot_estoi / num_eval_files
