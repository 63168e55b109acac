#!/usr/bin/env python3
"""Inference CLI: counterpart of the reference's ``evaluate.py`` (evaluate.py:27-194) on the HIP sampler.

    python -m flowmse_amd.evaluate --test_dir DATA --folder_destination OUT --ckpt MODEL.ckpt [--N 5]

Same arguments and outputs (``files/*.wav``, ``_results.csv``, ``_avg_results.txt``, ``_settings.txt``).  Per
utterance it follows evaluate.py:101-136: load wav -> normalise by max|y| -> STFT -> magnitude compression
(spec_fwd) -> pad frames to a multiple of 64 -> white-box Euler sampler -> spec_back -> iSTFT -> rescale.
Wave I/O uses scipy (torchaudio / soundfile are optional), PESQ / ESTOI are reported when the ``pesq`` / ``pystoi``
packages are importable, SI-SDR / SI-SIR / SI-SAR always (utils.py:10-35).  ``--synthetic`` runs with synthetic
weights and synthetic noisy/clean pairs (no checkpoint or dataset needed) as an end-to-end smoke demo.
"""
import argparse
import csv
import glob
import os
import re
import time

import numpy as np
import torch

from flowmse_amd.sampling import get_white_box_solver
from flowmse_amd.util.other import pad_spec, read_wav as _read_wav


def energy_ratios(s_hat, s, n):
    """SI-SDR, SI-SIR, SI-SAR in dB (reference utils.py:10-35)."""
    alpha_s = np.dot(s_hat, s) / np.linalg.norm(s) ** 2
    s_target = alpha_s * s
    alpha_n = np.dot(s_hat, n) / np.linalg.norm(n) ** 2
    e_noise = alpha_n * n
    e_art = s_hat - s_target - e_noise
    si_sdr = 10 * np.log10(np.linalg.norm(s_target) ** 2 / np.linalg.norm(e_noise + e_art) ** 2)
    si_sir = 10 * np.log10(np.linalg.norm(s_target) ** 2 / np.linalg.norm(e_noise) ** 2)
    si_sar = 10 * np.log10(np.linalg.norm(s_target) ** 2 / np.linalg.norm(e_art) ** 2)
    return si_sdr, si_sir, si_sar


def mean_std(data):
    a = np.asarray([d for d in data if np.isfinite(d)], dtype=np.float64)
    return f"{a.mean():.2f} ± {a.std():.2f}" if a.size else "nan"


def enhance_waveform(model, y, N=5, T_rev=1.0, t_eps=0.03, odesolver="euler", z=None, VF_fn=None, device=None):
    """One utterance, evaluate.py:107-136.  y: float tensor [1, samples].  Returns the enhanced waveform (numpy)."""
    device = device or y.device
    T_orig = y.size(1)
    norm_factor = y.abs().max().item()
    dm = model.data_module
    fused = VF_fn is None and hasattr(dm, "fused_ok") and dm.fused_ok(y.to(device))
    if fused:                      # STFT + compression + frame padding as one HIP kernel
        Y = dm.analyze(y.to(device) / norm_factor)       # y / max|y| as the reference computes it (evaluate.py:111)
    else:
        y = y / norm_factor
        Y = torch.unsqueeze(model._forward_transform(model._stft(y.to(device))), 0)
        Y = pad_spec(Y)
    sampler = get_white_box_solver(odesolver, model.ode, VF_fn if VF_fn is not None else model, Y=Y, Y_prior=Y,
                                   T_rev=T_rev, t_eps=t_eps, N=N, z=z)
    sample, _ = sampler()
    if fused:                      # decompression + iSTFT + rescale as one HIP kernel
        return dm.synthesize(sample, T_orig, norm_factor).squeeze().cpu().numpy()
    x_hat = model.to_audio(sample.squeeze(), T_orig)
    return (x_hat * norm_factor).squeeze().cpu().numpy()


def enhance_batch(model, ys, N=5, T_rev=1.0, t_eps=0.03, odesolver="euler"):
    """Several utterances whose padded frame counts agree, as ONE sampler call (the reference enhances one file at
    a time, evaluate.py:97; trajectories are independent, so batching changes nothing but throughput).
    ys: list of float tensors [1, samples_i] on the target device.  Returns a list of numpy waveforms."""
    norms = [y.abs().max().item() for y in ys]
    dm = model.data_module
    fused = hasattr(dm, "fused_ok") and all(dm.fused_ok(y) for y in ys)
    if fused:                      # STFT + compression + frame padding: one HIP kernel per utterance
        specs = [dm.analyze(y / n) for y, n in zip(ys, norms)]
    else:
        specs = [pad_spec(torch.unsqueeze(model._forward_transform(model._stft(y / n)), 0)) for y, n in zip(ys, norms)]
    Y = torch.cat(specs, dim=0)
    sample, _ = get_white_box_solver(odesolver, model.ode, model, Y=Y, Y_prior=Y, T_rev=T_rev, t_eps=t_eps, N=N)()
    if fused:                      # decompression + iSTFT + rescale: one HIP kernel per utterance
        return [dm.synthesize(sample[i:i + 1], y.size(1), n).squeeze().cpu().numpy()
                for i, (y, n) in enumerate(zip(ys, norms))]
    return [(model.to_audio(sample[i, 0], y.size(1)) * n).squeeze().cpu().numpy()
            for i, (y, n) in enumerate(zip(ys, norms))]


def _write_wav(path, x, sr=16000):
    """16-bit PCM WAV like the reference's ``soundfile.write(path, x_hat, 16000)`` (evaluate.py:147; libsndfile's
    default subtype for .wav is PCM_16, float samples scaled by 0x7FFF and rounded to nearest).  soundfile itself is
    used when importable, so output trees diff cleanly against the reference's.  Without it scipy writes samples formed
    the way libsndfile forms them: the product ``x * 32767`` in float32 (libsndfile scales in the sample's own type
    before ``lrintf``), round half to even, and -- libsndfile does NOT clip by default (SFC_SET_CLIPPING off) -- a
    sample with |x| > 1, possible after the rescale by max|y|, keeps the low 16 bits of its rounded value instead of
    saturating.  Non-finite samples (undefined in libsndfile's float -> int conversion) are written as 0.  The fallback
    aims at the same file for finite input; it has not been diffed against libsndfile here (soundfile is not in
    this image)."""
    x = np.asarray(x, dtype=np.float32)
    try:
        import soundfile
        soundfile.write(path, x, sr)
        return
    except ImportError:
        pass
    from scipy.io import wavfile
    scaled = np.rint(x * np.float32(32767.0))                      # float32 product, like libsndfile
    scaled = np.where(np.isfinite(scaled), scaled, np.float32(0.0))
    pcm = scaled.astype(np.int64).astype(np.uint16).astype(np.int16)
    wavfile.write(path, sr, pcm)


def _synthetic_pairs(n, seconds=2.0, sr=16000, seed=0):
    g = np.random.default_rng(seed)
    t = np.arange(int(seconds * sr)) / sr
    out = []
    for i in range(n):
        clean = 0.3 * np.sin(2 * np.pi * (200 + 60 * i) * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t))
        noisy = clean + 0.05 * g.standard_normal(t.shape)
        out.append((f"synthetic_{i:02d}.wav", clean.astype(np.float32), noisy.astype(np.float32)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--test_dir", type=str, default=None, help="directory with test/clean and test/noisy")
    ap.add_argument("--odesolver_type", type=str, choices=("white",), default="white")
    ap.add_argument("--odesolver", type=str, default="euler")
    ap.add_argument("--reverse_starting_point", type=float, default=1.0)
    ap.add_argument("--last_eval_point", type=float, default=0.03)
    ap.add_argument("--folder_destination", type=str, required=True)
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--N", type=int, default=5)
    ap.add_argument("--N_mid", type=int, default=0, help="accepted for command-line compatibility (evaluate.py:42: "
                                                         "'not related to FlowSE'); must be 0")
    ap.add_argument("--synthetic", type=int, default=0, help="run on this many synthetic pairs with synthetic weights")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16", "fp16"])
    ap.add_argument("--batch", type=int, default=1,
                    help="enhance up to this many utterances of equal padded length per sampler call (1 = reference behaviour)")
    args = ap.parse_args(argv)
    if args.N_mid != 0:
        raise ValueError("N_mid should be 0.")          # evaluate.py:124-125

    from flowmse_amd.model import VFModel
    if args.synthetic:
        from flowmse_amd.util import synth
        model = VFModel(backbone="ncsnpp", ode="flowmatching")
        model.dnn.load_state_dict({n: torch.from_numpy(synth.synth_param(n, tuple(p.shape)))
                                   for n, p in model.dnn.named_parameters()})
        pairs = _synthetic_pairs(args.synthetic)
    else:
        if not args.ckpt or not args.test_dir:
            ap.error("--ckpt and --test_dir are required unless --synthetic is given")
        model = VFModel.load_from_checkpoint(args.ckpt, base_dir="", batch_size=8, num_workers=4,
                                             kwargs=dict(gpu=False))
        clean_dir = os.path.join(args.test_dir, "test", "clean")
        noisy_dir = os.path.join(args.test_dir, "test", "noisy")
        pairs = []
        for f in sorted(glob.glob(os.path.join(noisy_dir, "*.wav"))):
            name = os.path.basename(f)
            pairs.append((name, _read_wav(os.path.join(clean_dir, name))[0][0].numpy(), _read_wav(f)[0][0].numpy()))
    model.eval(no_ema=False)
    model.cuda()
    model.dnn.set_precision(args.precision)
    m = re.search(r"epoch=(\d+)", args.ckpt or "")
    epoch = m.group(1) if m else "n/a"

    target_dir = args.folder_destination.rstrip("/") + "/"
    os.makedirs(target_dir + "files/", exist_ok=True)
    try:
        from pesq import pesq
    except Exception:
        pesq = None
    try:
        from pystoi import stoi
    except Exception:
        stoi = None
    data = {"filename": [], "pesq": [], "estoi": [], "si_sdr": [], "si_sir": [], "si_sar": []}
    sr = 16000
    frames, t0 = 0, time.time()
    from flowmse_amd.parallel import batches_by_length
    enhanced = {}
    if args.batch > 1:                 # group by padded frame count, largest first
        lens = [(((p[2].shape[0] // 128 + 1) + 63) // 64) * 64 for p in pairs]
        for _, ids in batches_by_length(range(len(pairs)), lens, args.batch):
            outs = enhance_batch(model, [torch.from_numpy(pairs[i][2])[None].cuda() for i in ids], N=args.N,
                                 T_rev=args.reverse_starting_point, t_eps=args.last_eval_point,
                                 odesolver=args.odesolver)
            enhanced.update(dict(zip(ids, outs)))
    for idx, (name, x, y) in enumerate(pairs):
        if idx in enhanced:
            x_hat = enhanced[idx]
        else:
            x_hat = enhance_waveform(model, torch.from_numpy(y)[None].cuda(), N=args.N,
                                     T_rev=args.reverse_starting_point, t_eps=args.last_eval_point,
                                     odesolver=args.odesolver)
        frames += y.shape[0] // 128 + 1
        n = y - x
        _write_wav(target_dir + "files/" + name, x_hat, sr)
        data["filename"].append(name)
        try:
            p = pesq(sr, x, x_hat, "wb") if pesq else float("nan")
        except Exception:
            p = float("nan")
        data["pesq"].append(p)
        data["estoi"].append(stoi(x, x_hat, sr, extended=True) if stoi else float("nan"))
        r = energy_ratios(x_hat, x, n)
        data["si_sdr"].append(r[0]); data["si_sir"].append(r[1]); data["si_sar"].append(r[2])
    dt = time.time() - t0

    with open(os.path.join(target_dir, "_results.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(list(data.keys()))
        for i in range(len(data["filename"])):
            w.writerow([data[k][i] for k in data])
    with open(os.path.join(target_dir, "_avg_results.txt"), "w") as f:
        f.write("PESQ: {} \n".format(mean_std(data["pesq"])))
        f.write("ESTOI: {} \n".format(mean_std(data["estoi"])))
        f.write("SI-SDR: {} \n".format(mean_std(data["si_sdr"])))
        f.write("SI-SIR: {} \n".format(mean_std(data["si_sir"])))
        f.write("SI-SAR: {} \n".format(mean_std(data["si_sar"])))
    with open(os.path.join(target_dir, "_settings.txt"), "w") as f:
        f.write(f"epoch: {epoch}\ncheckpoint file: {args.ckpt}\nodesolver_type: {args.odesolver_type}\n")
        f.write(f"odesolver: {args.odesolver}\nReverse starting point: {args.reverse_starting_point}\n")
        f.write(f"Last evaluated point: {args.last_eval_point}\ndata: {args.test_dir}\node: FLOWMATCHING\n")
        f.write(f"sigma_min: {model.ode.sigma_min}\nsigma_max: {model.ode.sigma_max}\nN: {args.N}\n")
        f.write(f"precision: {args.precision}\n")
    print(f"enhanced {len(pairs)} utterances ({frames} frames) in {dt:.2f} s -> {target_dir}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
