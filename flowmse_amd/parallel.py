"""Per-utterance data parallelism: one process per GPU, weights replicated, no collective in the solver loop.

The reference enhances utterances one by one on one GPU (evaluate.py:97); each trajectory depends only on its
own (Y, z) and the shared read-only weights (sampling/__init__.py:36-60; GroupNorm and attention are
per-sample), so the path shards by utterance.  The only exchange step is the final gather of the enhanced
spectrograms to rank 0 -- one RCCL (backend "nccl" on ROCm) or gloo collective on padded buffers.
"""
import torch
import torch.distributed as dist


def shard_utterances(lengths, world_size):
    """Greedy longest-processing-time assignment of utterances (by padded frame count) to ranks.

    Returns a list (one entry per rank) of utterance-index lists; deterministic for equal lengths."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return shards


def batches_by_length(indices, lengths, max_batch):
    """Group a rank's utterances into batches of equal padded length (the kernels take one T per call)."""
    by_len = {}
    for i in indices:
        by_len.setdefault(int(lengths[i]), []).append(i)
    out = []
    for T in sorted(by_len, reverse=True):
        ids = by_len[T]
        for k in range(0, len(ids), max_batch):
            out.append((T, ids[k:k + max_batch]))
    return out


def gather_spectrograms(local, local_ids, n_total, group=None):
    """Gather per-utterance complex spectrograms [1,F,T_i] from all ranks to rank 0.

    local: list of complex64 tensors [F, T_i] on this rank's device; local_ids: their global utterance indices.
    Returns on rank 0 a list of n_total tensors (CPU), elsewhere None.  Two collectives: lengths/ids (int64)
    then one padded payload all_gather (payload is O(MB) per utterance, negligible on xGMI)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        out = [None] * n_total
        for i, s in zip(local_ids, local):
            out[i] = s.cpu()
        return out
    dev = local[0].device if local else torch.device("cuda", torch.cuda.current_device()) \
        if torch.cuda.is_available() and dist.get_backend(group) == "nccl" else torch.device("cpu")
    F = local[0].shape[0] if local else 0
    meta = torch.tensor([len(local), F, max([s.shape[1] for s in local], default=0)], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    n_max = int(max(m[0] for m in metas))
    F = int(max(m[1] for m in metas))
    t_max = int(max(m[2] for m in metas))
    ids = torch.full((n_max, 2), -1, dtype=torch.int64, device=dev)
    payload = torch.zeros((n_max, F, t_max, 2), dtype=torch.float32, device=dev)
    for k, (i, s) in enumerate(zip(local_ids, local)):
        ids[k, 0] = i
        ids[k, 1] = s.shape[1]
        payload[k, :, :s.shape[1]] = torch.view_as_real(s)
    all_ids = [torch.zeros_like(ids) for _ in range(world)]
    all_pay = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(all_ids, ids, group=group)
    dist.all_gather(all_pay, payload, group=group)
    if rank != 0:
        return None
    out = [None] * n_total
    for r in range(world):
        for k in range(n_max):
            i, t = int(all_ids[r][k, 0]), int(all_ids[r][k, 1])
            if i >= 0:
                out[i] = torch.view_as_complex(all_pay[r][k, :, :t].contiguous()).cpu()
    return out


def enhance_sharded(sample_fn, specs, max_batch=8, group=None, pad_multiple=64, keep_padding=False):
    """Enhance a ragged set of utterances data-parallel over the ranks of `group` (BASELINE config 4).

    specs: list of complex64 spectrograms [F, T_i] (every rank holds the same list, or at least the entries of its
    own shard); sample_fn(Y, ids) -> X maps a zero-padded batch Y [b,1,F,T] of equal padded length to the enhanced
    batch (ids = global utterance indices of the rows, e.g. to pick reproducible noise).  Utterances are dealt to
    ranks by padded length (LPT), batched by equal padded length, enhanced and gathered to rank 0 with ONE exchange at
    the very end.  Returns on rank 0 the list of enhanced spectrograms, None elsewhere:

    * keep_padding=False: each cropped back to its own [F, T_i] (the spectrogram of the utterance);
    * keep_padding=True: the whole padded [F, Tpad_i] sample.  This is what the reference feeds to the iSTFT
      (evaluate.py:132 ``model.to_audio(sample, T_orig)`` on the padded sample): the zero-padded frames are no longer
      zero after enhancement and, through the overlap-add and the window envelope, reach the last ~127 samples of the
      waveform.  Use it whenever waveforms must equal the per-utterance path's (``SpecTransform.synthesize``).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    true_len = [int(s.shape[-1]) for s in specs]
    padded = [((t + pad_multiple - 1) // pad_multiple) * pad_multiple for t in true_len]
    mine = shard_utterances(padded, world)[rank]
    out_local, ids_local = [], []
    for T, ids in batches_by_length(mine, padded, max_batch):
        Y = torch.stack([torch.nn.functional.pad(specs[i], (0, T - true_len[i])) for i in ids])[:, None]
        X = sample_fn(Y.contiguous(), ids)
        for row, i in enumerate(ids):
            out_local.append(X[row, 0] if keep_padding else X[row, 0, :, :true_len[i]])
            ids_local.append(i)
    return gather_spectrograms(out_local, ids_local, len(specs), group=group)
