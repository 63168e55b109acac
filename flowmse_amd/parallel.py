"""Per-utterance data parallelism: one process per GPU, weights replicated, no collective in the solver loop.

The reference enhances utterances one by one on one GPU (evaluate.py:97); each trajectory depends only on its
own (Y, z) and the shared read-only weights (sampling/__init__.py:36-60; GroupNorm and attention are
per-sample), so the path shards by utterance.  The only exchange step is the final gather of the enhanced
spectrograms to rank 0: a few small metadata collectives, then ONE ``dist.gather`` to rank 0 per distinct padded
length (RCCL -- backend "nccl" on ROCm -- on device tensors, gloo on CPU tensors).  Nothing lands on the other ranks.
"""
import time

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


def _pad_to(t, multiple):
    return ((int(t) + multiple - 1) // multiple) * multiple


def gather_spectrograms(local, local_ids, n_total, group=None, pad_multiple=64):
    """Gather per-utterance complex spectrograms [F, T_i] from all ranks to rank 0.

    local: list of complex64 tensors [F, T_i] on this rank's device; local_ids: their global utterance indices.
    Returns on rank 0 a list of n_total tensors (CPU, each with its own T_i), elsewhere None.

    Exchange: (1) one all_gather of the per-rank utterance counts and one of the (id, T_i) tables -- a few hundred
    bytes; (2) per distinct PADDED length (T_i rounded up to `pad_multiple`, i.e. the lengths the sampler batches by)
    one ``dist.gather`` to rank 0 of a [k, F, Tpad] slab, k = the largest count any rank holds at that length.  Only
    rank 0 receives payload: world x k x F x Tpad x 8 bytes per bucket, against the n_max x F x t_max slab an
    all_gather of one padded buffer would land on EVERY rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        out = [None] * n_total
        for i, s in zip(local_ids, local):
            out[i] = s.cpu()
        return out
    on_gpu = dist.get_backend(group) == "nccl"
    dev = (local[0].device if local else torch.device("cuda", torch.cuda.current_device())) if on_gpu \
        else torch.device("cpu")
    # ---- (1) metadata: which rank holds which utterance at which length
    cnt = torch.tensor([len(local), local[0].shape[0] if local else 0], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    n_max = max(int(c[0]) for c in cnts)
    F = max(int(c[1]) for c in cnts)
    table = torch.full((max(n_max, 1), 2), -1, dtype=torch.int64)
    for k, (i, s) in enumerate(zip(local_ids, local)):
        table[k, 0] = int(i)
        table[k, 1] = int(s.shape[1])
    table = table.to(dev)
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table, group=group)
    tables = [t.cpu().tolist() for t in tables]                    # [(id, T_i)] per rank, in the holder's order
    # ---- (2) payload: one gather per distinct padded length, every rank walks the buckets in the same order
    buckets = {}                                                   # Tpad -> per rank list of (position on the rank, id, T_i)
    for r in range(world):
        for k, (i, t) in enumerate(tables[r]):
            if i >= 0:
                buckets.setdefault(_pad_to(t, pad_multiple), [[] for _ in range(world)])[r].append((k, i, t))
    out = [None] * n_total if rank == 0 else None
    for Tp in sorted(buckets, reverse=True):
        per_rank = buckets[Tp]
        kmax = max(len(v) for v in per_rank)
        slab = torch.zeros((kmax, F, Tp, 2), dtype=torch.float32, device=dev)
        for row, (k, _, t) in enumerate(per_rank[rank]):
            slab[row, :, :t] = torch.view_as_real(local[k]).to(dev)
        recv = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
        dist.gather(slab, recv, dst=0, group=group)
        if rank == 0:
            for r in range(world):
                got = recv[r].cpu()
                for row, (_, i, t) in enumerate(per_rank[r]):
                    out[i] = torch.view_as_complex(got[row, :, :t].contiguous())
    return out


def enhance_sharded(sample_fn, specs, max_batch=8, group=None, pad_multiple=64, keep_padding=False, stats=None):
    """Enhance a ragged set of utterances data-parallel over the ranks of `group` (BASELINE config 4; the reference's
    loop over the test set is evaluate.py:97-136, one utterance per sampler call on one GPU).

    specs: list of complex64 spectrograms [F, T_i] (every rank holds the same list, or at least the entries of its
    own shard -- only the LENGTHS of the others are read); sample_fn(Y, ids) -> X maps a zero-padded batch Y [b,1,F,T]
    of equal padded length to the enhanced batch (ids = global utterance indices of the rows, e.g. to pick
    reproducible noise).  Utterances are dealt to ranks by padded length (LPT), batched by equal padded length,
    enhanced and gathered to rank 0 with ONE exchange step at the very end (gather_spectrograms).  Returns on rank 0
    the list of enhanced spectrograms, None elsewhere:

    * keep_padding=False: each cropped back to its own [F, T_i] (the spectrogram of the utterance);
    * keep_padding=True: the whole padded [F, Tpad_i] sample.  This is what the reference feeds to the iSTFT
      (evaluate.py:132 ``model.to_audio(sample, T_orig)`` on the padded sample): the zero-padded frames are no longer
      zero after enhancement and, through the overlap-add and the window envelope, reach the last ~127 samples of the
      waveform.  Use it whenever waveforms must equal the per-utterance path's (``SpecTransform.synthesize``).

    stats (optional dict) receives this rank's share of the job: ``utterances``, ``batches``, ``frames`` (sum of the
    padded frame counts it enhanced), ``true_frames``, ``sample_s`` (wall time of its sampler calls, device drained)
    and ``gather_s`` (the exchange).  Load imbalance = the spread of ``frames`` / ``sample_s`` over the ranks.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    true_len = [int(s.shape[-1]) for s in specs]
    padded = [_pad_to(t, pad_multiple) for t in true_len]
    mine = shard_utterances(padded, world)[rank]
    out_local, ids_local = [], []
    t0 = time.perf_counter()
    batches = batches_by_length(mine, padded, max_batch)
    for T, ids in batches:
        Y = torch.stack([torch.nn.functional.pad(specs[i], (0, T - true_len[i])) for i in ids])[:, None]
        X = sample_fn(Y.contiguous(), ids)
        for row, i in enumerate(ids):
            out_local.append(X[row, 0] if keep_padding else X[row, 0, :, :true_len[i]])
            ids_local.append(i)
    if stats is not None:
        if torch.cuda.is_available() and out_local and out_local[0].is_cuda:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
    res = gather_spectrograms(out_local, ids_local, len(specs), group=group, pad_multiple=pad_multiple)
    if stats is not None:
        stats.update(utterances=len(mine), batches=len(batches), frames=sum(padded[i] for i in mine),
                     true_frames=sum(true_len[i] for i in mine), sample_s=t1 - t0,
                     gather_s=time.perf_counter() - t1)
    return res
