"""Per-utterance data parallelism: one process per GPU, weights replicated, no collective in the solver loop.

Host-only planning (no GPU, no process group): ``plan_shards`` / ``plan_summary`` -- ``python bench.py --workload vbdmd
--plan --gpus N`` prints them for BASELINE config[3].

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


# Cost model of one sampler call on a batch of b utterances padded to T frames: T * (COST_ALPHA + COST_BETA * b).
# Fitted to the round-6 MI355X measurements of the N=5 fp32 sampler ([1,1,256,256]: 11.3 k frames/s, [8,1,256,256]:
# 21.8 k frames/s => 88 us / frame at b = 1, 46 us / frame / utterance at b = 8; round 3's fit was 65 / 48).  Only the RATIO
# matters (1.35 -> 1.2: a batch slot costs relatively more than it did): the planner uses it to cut the length-sorted set into
# batches and to decide whether padding a straggler up to a longer batch is cheaper than running it in an under-filled one.
COST_ALPHA, COST_BETA = 48.0, 40.0


def batch_cost(T, b):
    """Modelled time (us) of one N=5 sampler call on b utterances of T padded frames."""
    return T * (COST_ALPHA + COST_BETA * b) if b else 0.0


def plan_batches(indices, lengths, max_batch, promote=False):
    """Cost-optimal batching of a set of utterances: [(T, ids)] with T = the batch's padded length (its longest member).

    The utterances are sorted by length (descending); every batch is a run of <= max_batch consecutive ones.  A dynamic
    programme over the sorted list picks the cut points that minimise the modelled time sum batch_cost(T, b).

    promote=False (default): a batch only holds utterances of ONE padded length, so every utterance runs at exactly the
    length pad_spec gives it in the reference (util/other.py:83-90) and its result equals the per-utterance path's.
    promote=True: the stragglers of one length may join the next longer batch, zero-padded up to its T, when the sampler
    call they save (COST_ALPHA * T') outweighs the frames they add (COST_BETA * (T - T') each).  NOT result-preserving:
    the network is not padding-invariant (GroupNorm statistics and the attention at the 16-bin level run over all
    frames of a sample, ncsnpp.py:289-330), so a promoted utterance comes out as the reference would compute it at the
    LONGER padding -- a valid enhancement of the same input, but not bit-comparable with the reference's own call."""
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i))
    n = len(order)
    best = [0.0] + [float("inf")] * n
    cut = [0] * (n + 1)
    for k in range(1, n + 1):
        for b in range(1, min(max_batch, k) + 1):
            if not promote and int(lengths[order[k - b]]) != int(lengths[order[k - 1]]):
                break
            c = best[k - b] + batch_cost(int(lengths[order[k - b]]), b)
            if c < best[k] - 1e-9:
                best[k], cut[k] = c, b
    out, k = [], n
    while k > 0:
        b = cut[k]
        out.append((int(lengths[order[k - b]]), order[k - b:k]))
        k -= b
    out.reverse()
    return out


def _load(batches):
    return sum(batch_cost(T, len(ids)) for T, ids in batches)


def plan_shards(lengths, world_size, max_batch, promote=False):
    """Deal a ragged utterance set to `world_size` ranks as ready-made batches: [[(T, ids), ...] per rank].

    lengths: padded frame counts (multiples of the model's 64-frame granule).  (1) plan_batches() over the WHOLE set:
    batches are formed before they are dealt, so a rank receives full batches wherever the set allows it (dealing
    single utterances first and batching per rank afterwards -- round 3 -- left every rank with one under-filled
    sampler call per distinct length: at 8 ranks the 512 / 576 / 640-frame buckets ran at batch 1..6 rates).  (2) The
    batches go, most expensive first, each to the currently least-loaded rank (LPT on modelled time; deterministic).
    (3) Levelling: while moving part of a batch from the most to the least loaded rank shortens the longer of the
    two, do so (this is what splits a lone batch over idle ranks when there are fewer batches than ranks)."""
    batches = plan_batches(range(len(lengths)), lengths, max_batch, promote)
    batches.sort(key=lambda b: (-batch_cost(b[0], len(b[1])), -b[0], b[1][0]))
    loads = [0.0] * world_size
    plan = [[] for _ in range(world_size)]
    for T, ids in batches:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        plan[r].append((T, list(ids)))
        loads[r] += batch_cost(T, len(ids))
    for _ in range(4 * world_size):                                # levelling moves (bounded)
        hi = max(range(world_size), key=lambda k: (loads[k], -k))
        lo = min(range(world_size), key=lambda k: (loads[k], k))
        best = None                                                # (new makespan of the pair, batch index, count moved)
        for bi, (T, ids) in enumerate(plan[hi]):
            for m in range(1, len(ids) + 1):                       # move the m SHORTEST members (the tail: ids are
                moved = ids[len(ids) - m:]                         # sorted long -> short) as a batch of their own
                Tm = int(lengths[moved[0]])
                new_hi = loads[hi] - batch_cost(T, len(ids)) + batch_cost(T, len(ids) - m)
                new_lo = loads[lo] + batch_cost(Tm, m)
                span = max(new_hi, new_lo)
                if span < loads[hi] - 1e-6 and (best is None or span < best[0] - 1e-9):
                    best = (span, bi, m)
        if best is None:
            break
        _, bi, m = best
        T, ids = plan[hi][bi]
        moved, kept = ids[len(ids) - m:], ids[:len(ids) - m]
        if kept:
            plan[hi][bi] = (T, kept)
        else:
            del plan[hi][bi]
        plan[lo].append((int(lengths[moved[0]]), moved))
        loads[hi], loads[lo] = _load(plan[hi]), _load(plan[lo])
    for r in range(world_size):
        plan[r].sort(key=lambda b: (-b[0], b[1][0]))
    return plan


def plan_summary(plan, lengths, max_batch):
    """Per-rank figures of a plan_shards() result (host only): utterances, batches, padded frames as dealt (`frames`),
    frames as run incl. promotion padding (`frames_run`), batch fill = sum(b T) / sum(max_batch T), modelled time."""
    ranks = []
    for batches in plan:
        frames = sum(int(lengths[i]) for _, ids in batches for i in ids)
        run = sum(T * len(ids) for T, ids in batches)
        cap = sum(T * max_batch for T, _ in batches)
        ranks.append({"utterances": sum(len(ids) for _, ids in batches), "batches": len(batches), "frames": frames,
                      "frames_run": run, "batch_fill": (run / cap) if cap else 1.0,
                      "promoted": sum(1 for T, ids in batches for i in ids if int(lengths[i]) != T),
                      "model_ms": sum(batch_cost(T, len(ids)) for T, ids in batches) / 1e3})
    mean_f = sum(r["frames"] for r in ranks) / max(len(ranks), 1)
    mean_t = sum(r["model_ms"] for r in ranks) / max(len(ranks), 1)
    run, cap = sum(r["frames_run"] for r in ranks), sum(T * max_batch for b in plan for T, _ in b)
    return {"world": len(plan), "max_batch": max_batch, "per_rank": ranks,
            "frame_imbalance_max_over_mean": (max(r["frames"] for r in ranks) / mean_f) if mean_f else 1.0,
            "model_time_imbalance_max_over_mean": (max(r["model_ms"] for r in ranks) / mean_t) if mean_t else 1.0,
            "batch_fill": (run / cap) if cap else 1.0,
            "promotion_padding_frames": run - sum(r["frames"] for r in ranks)}


def _pad_to(t, multiple):
    return ((int(t) + multiple - 1) // multiple) * multiple


# ---- rank 0's device -> host tail.  One grow-only page-locked staging buffer per process (page-locking ~0.5 GB per
# pass cost more than the copies it serves); results are VIEWS into it, so a buffer is only reused once nothing returned
# from an earlier call references its storage any more (torch's storage use count) -- a caller that keeps two result
# sets alive simply owns two buffers.
_STAGE_POOL = []
_STAGE_POOL_MAX = 2                                                # unreferenced buffers kept for reuse (a live result set owns its own)
_SIDE_STREAMS = {}                                                 # device index -> the drain's copy stream


def _use_count(buf):
    """References to the storage of `buf` (torch's storage use count), or None where this build of torch has no such query."""
    fn = getattr(torch._C, "_storage_Use_Count", None)
    if fn is None:
        return None
    try:
        return int(fn(buf.untyped_storage()._cdata))
    except Exception:
        return None


def _storage_free(buf):
    n = _use_count(buf)
    return n is not None and n <= 2                                # `buf` itself + the temporary


def _new_stage(nfloats):
    buf = torch.empty(max(int(nfloats * 1.125), 1), dtype=torch.float32)
    if torch.cuda.is_available():                                  # (plain memory in the CPU-only unit test)
        buf = buf.pin_memory()
    return buf


def _acquire_stage(nfloats):
    """A pinned fp32 buffer of >= nfloats elements that no live tensor views; grown (not shrunk) on demand.  The pool keeps
    at most _STAGE_POOL_MAX unreferenced buffers (the largest ones); where the use count cannot be queried nothing is pooled
    at all -- a fresh buffer per call is slow (it is page-locked each time) but can never alias a result."""
    if _use_count(torch.empty(1)) is None:
        return _new_stage(nfloats)
    free = sorted((k for k, buf in enumerate(_STAGE_POOL) if _storage_free(buf)), key=lambda k: -_STAGE_POOL[k].numel())
    hit = next((k for k in free if _STAGE_POOL[k].numel() >= nfloats), None)
    keep = set(free[:_STAGE_POOL_MAX]) | ({hit} if hit is not None else set())
    if hit is None and free:
        keep.discard(free[0])                                      # too small and unreferenced: replaced below
    buf = _STAGE_POOL[hit] if hit is not None else _new_stage(nfloats)
    _STAGE_POOL[:] = [b for k, b in enumerate(_STAGE_POOL) if k in keep or not _storage_free(b)]
    if hit is None:
        _STAGE_POOL.append(buf)
    return buf


def stage_pool_stats():
    """(buffers, total bytes) of the pinned staging pool -- test / bench hook."""
    return len(_STAGE_POOL), sum(b.numel() * 4 for b in _STAGE_POOL)


class _HostDrain:
    """Rows of device slabs -> one pinned staging buffer, asynchronously on a side stream; one synchronise at the end."""

    def __init__(self, F, rows_total_floats, dev):
        self.F = F
        self.stage = _acquire_stage(rows_total_floats)
        self.off = 0
        self.views = []
        self.keep = []                                             # device slabs stay alive until finish()
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
        self.side = _SIDE_STREAMS[key]

    def add(self, rows):
        """rows: [(device tensor [F, >= t, 2] fp32, id, t)] that are complete in the CURRENT stream's order."""
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            for src, i, t in rows:
                n = self.F * t * 2
                v = self.stage[self.off:self.off + n].view(self.F, t, 2)
                v.copy_(src[:, :t], non_blocking=True)
                self.views.append((i, v))
                self.off += n
                self.keep.append(src)

    def finish(self, out):
        self.side.synchronize()
        self.keep.clear()
        for i, v in self.views:
            out[i] = torch.view_as_complex(v)
        return out


def gather_spectrograms(local, local_ids, n_total, group=None, pad_multiple=64, to_host=True):
    """Gather per-utterance complex spectrograms [F, T_i] from all ranks to rank 0.

    local: list of complex64 tensors [F, T_i] on this rank's device; local_ids: their global utterance indices.
    Returns on rank 0 a list of n_total tensors (each with its own T_i), elsewhere None.  to_host=True: CPU tensors --
    on GPUs these are views into a process-wide pinned staging buffer that is recycled once nothing references it (no
    per-call page-locking, no per-row copy on the host).  to_host=False (RCCL only): the rows stay on rank 0's device as
    views of the received slabs (the iSTFT runs there anyway); no device -> host traffic at all.

    Exchange: (1) one all_gather of the per-rank utterance counts and one of the (id, T_i) tables -- a few hundred
    bytes; (2) per distinct PADDED length (T_i rounded up to `pad_multiple`, i.e. the lengths the sampler batches by)
    one ``dist.gather`` to rank 0 of a [k, F, Tpad] slab, k = the largest count any rank holds at that length.  Only
    rank 0 receives payload: world x k x F x Tpad x 8 bytes per bucket, against the n_max x F x t_max slab an
    all_gather of one padded buffer would land on EVERY rank.  Rank 0 starts a bucket's device -> host copies on a side
    stream as soon as its gather is ordered, so they run under the next bucket's collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        out = [None] * n_total
        if not to_host:
            for i, s in zip(local_ids, local):
                out[i] = s
        elif local and local[0].is_cuda:
            F = local[0].shape[0]
            drain = _HostDrain(F, sum(F * s.shape[1] * 2 for s in local), local[0].device)
            drain.add([(torch.view_as_real(s), i, s.shape[1]) for i, s in zip(local_ids, local)])
            drain.finish(out)
        else:
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
    drain = None
    if rank == 0 and on_gpu and to_host:
        drain = _HostDrain(F, sum(F * t * 2 for tb in tables for i, t in tb if i >= 0), dev)
    for Tp in sorted(buckets, reverse=True):
        per_rank = buckets[Tp]
        kmax = max(len(v) for v in per_rank)
        slab = torch.zeros((kmax, F, Tp, 2), dtype=torch.float32, device=dev)
        for row, (k, _, t) in enumerate(per_rank[rank]):
            slab[row, :, :t] = torch.view_as_real(local[k]).to(dev)
        recv = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
        dist.gather(slab, recv, dst=0, group=group)
        if rank == 0:
            rows = [(recv[r][row], i, t) for r in range(world) for row, (_, i, t) in enumerate(per_rank[r])]
            if drain is not None:
                drain.add(rows)                                    # async, under the next bucket's collective
            elif on_gpu:                                           # to_host=False: device views of the received slabs
                for src, i, t in rows:
                    out[i] = torch.view_as_complex(src[:, :t])
            else:                                                  # gloo: the slabs are host memory already
                for src, i, t in rows:
                    out[i] = torch.view_as_complex(src[:, :t].contiguous())
    if drain is not None:
        drain.finish(out)
    return out


def enhance_sharded(sample_fn, specs, max_batch=8, group=None, pad_multiple=64, keep_padding=False, stats=None,
                    promote=False, plan=None, to_host=True):
    """Enhance a ragged set of utterances data-parallel over the ranks of `group` (BASELINE config 4; the reference's
    loop over the test set is evaluate.py:97-136, one utterance per sampler call on one GPU).

    specs: list of complex64 spectrograms [F, T_i] (every rank holds the same list, or at least the entries of its
    own shard -- only the LENGTHS of the others are read); sample_fn(Y, ids) -> X maps a zero-padded batch Y [b,1,F,T]
    of equal padded length to the enhanced batch (ids = global utterance indices of the rows, e.g. to pick
    reproducible noise).  The set is cut into equal-length batches first and the batches are dealt to the ranks by
    modelled time (plan_shards; `plan` = a precomputed plan_shards() result, e.g. one rank's share of a larger world;
    `promote`: see plan_batches -- off by default because it changes the promoted utterances' results), enhanced and
    gathered to rank 0 with ONE exchange step at the very end (gather_spectrograms).  Returns on rank 0
    the list of enhanced spectrograms (to_host: see gather_spectrograms), None elsewhere:

    * keep_padding=False: each cropped back to its own [F, T_i] (the spectrogram of the utterance);
    * keep_padding=True: the whole padded [F, Tpad_i] sample.  This is what the reference feeds to the iSTFT
      (evaluate.py:132 ``model.to_audio(sample, T_orig)`` on the padded sample): the zero-padded frames are no longer
      zero after enhancement and, through the overlap-add and the window envelope, reach the last ~127 samples of the
      waveform.  Use it whenever waveforms must equal the per-utterance path's (``SpecTransform.synthesize``).

    stats (optional dict) receives this rank's share of the job: ``utterances``, ``batches``, ``frames`` (sum of the
    padded frame counts it enhanced), ``frames_run`` (the same incl. promotion padding), ``batch_fill`` (frames_run /
    sum(max_batch * T) over its sampler calls), ``true_frames``, ``sample_s`` (wall time of its sampler calls, device
    drained) and ``gather_s`` (the exchange).  Load imbalance = the spread of ``frames`` / ``sample_s`` over the ranks.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    true_len = [int(s.shape[-1]) for s in specs]
    padded = [_pad_to(t, pad_multiple) for t in true_len]
    batches = plan_shards(padded, world, max_batch, promote)[rank] if plan is None else plan[rank]
    mine = [i for _, ids in batches for i in ids]
    out_local, ids_local = [], []
    t0 = time.perf_counter()
    for T, ids in batches:
        Y = torch.stack([torch.nn.functional.pad(specs[i], (0, T - true_len[i])) for i in ids])[:, None]
        X = sample_fn(Y.contiguous(), ids)
        for row, i in enumerate(ids):
            # keep_padding: the utterance's OWN padded length (what pad_spec gives it), also inside a promoted batch
            out_local.append(X[row, 0, :, :padded[i]] if keep_padding else X[row, 0, :, :true_len[i]])
            ids_local.append(i)
    if stats is not None:
        if torch.cuda.is_available() and out_local and out_local[0].is_cuda:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
    res = gather_spectrograms(out_local, ids_local, len(specs), group=group, pad_multiple=pad_multiple, to_host=to_host)
    if stats is not None:
        run = sum(T * len(ids) for T, ids in batches)
        cap = sum(T * max_batch for T, _ in batches)
        stats.update(utterances=len(mine), batches=len(batches), frames=sum(padded[i] for i in mine),
                     frames_run=run, batch_fill=(run / cap) if cap else 1.0,
                     true_frames=sum(true_len[i] for i in mine), sample_s=t1 - t0,
                     gather_s=time.perf_counter() - t1)
    return res
