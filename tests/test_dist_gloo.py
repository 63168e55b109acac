"""CPU, world_size 2 (gloo): the multi-GPU path's host logic -- sharding and the final gather collective."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flowmse_amd.parallel import gather_spectrograms, shard_utterances
    lengths = [128, 64, 192, 64, 256]
    shards = shard_utterances(lengths, world)
    mine = shards[rank]
    # "enhanced spectrogram" of utterance i: value i + 1j*t at frame t
    local = []
    for i in mine:
        s = torch.zeros(8, lengths[i], dtype=torch.complex64)
        s.real += i
        s.imag += torch.arange(lengths[i])[None, :]
        local.append(s)
    out = gather_spectrograms(local, mine, len(lengths))
    if rank == 0:
        ok = all(o is not None and o.shape == (8, lengths[i]) and float(o.real.mean()) == i
                 and float(o.imag[0, -1]) == lengths[i] - 1 for i, o in enumerate(out))
        q.put(ok)
    else:
        assert out is None
    dist.destroy_process_group()


def _worker_sharded(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flowmse_amd.parallel import enhance_sharded
    g = torch.Generator().manual_seed(0)
    lens = [70, 128, 501, 64, 300, 129, 640, 65, 256]
    specs = [torch.view_as_complex(torch.randn(8, t, 2, generator=g)) for t in lens]
    calls = []

    def fake_sampler(Y, ids):                       # stands in for the HIP sampler: must see equal padded lengths
        assert Y.shape[-1] % 64 == 0 and Y.shape[0] <= 3
        calls.append((Y.shape[-1], tuple(ids)))
        return Y * (2.0 + 0j) + 1.0

    st = {}
    out = enhance_sharded(fake_sampler, specs, max_batch=3, stats=st)
    padded = enhance_sharded(fake_sampler, specs, max_batch=3, keep_padding=True)
    # every utterance enhanced exactly once over the two ranks; this rank's share is what stats reports
    assert st["utterances"] == sum(len(ids) for _, ids in calls) // 2 and st["frames"] % 64 == 0
    assert st["batches"] == len(calls) // 2 and st["sample_s"] >= 0 and st["gather_s"] >= 0
    if rank == 0:
        ok = all(o.shape == s.shape and torch.allclose(o, s * 2 + 1) for o, s in zip(out, specs))
        # keep_padding: whole padded samples (the zero-padded frames come back as 0 * 2 + 1 = 1)
        ok = ok and all(p.shape[1] % 64 == 0 and torch.allclose(p[:, :s.shape[1]], s * 2 + 1) and
                        torch.allclose(p[:, s.shape[1]:], torch.ones_like(p[:, s.shape[1]:]))
                        for p, s in zip(padded, specs))
        q.put(ok)
    else:
        assert out is None and padded is None
    dist.destroy_process_group()


def _worker_uneven(rank, world, port, q):
    """Three ranks, one of them with nothing to send, several utterances per length bucket on the others."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flowmse_amd.parallel import gather_spectrograms
    holds = {0: [(4, 64), (1, 100)], 1: [], 2: [(0, 128), (2, 70), (3, 128), (5, 1)]}[rank]
    local = [torch.full((4, t), float(i), dtype=torch.complex64) + 1j * torch.arange(t)[None, :] for i, t in holds]
    out = gather_spectrograms(local, [i for i, _ in holds], 6)
    if rank == 0:
        want = {4: 64, 1: 100, 0: 128, 2: 70, 3: 128, 5: 1}
        q.put(all(out[i].shape == (4, t) and float(out[i].real.mean()) == i and float(out[i].imag[0, -1]) == t - 1
                  for i, t in want.items()))
    else:
        assert out is None
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_enhance_sharded_world2_gloo():
    """Ragged utterances -> LPT shards -> equal-length batches -> sampler -> crop -> single final gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.timeout(120)
def test_gather_world3_uneven_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_uneven, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.timeout(120)
def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
