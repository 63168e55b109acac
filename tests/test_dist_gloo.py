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

    out = enhance_sharded(fake_sampler, specs, max_batch=3)
    if rank == 0:
        ok = all(o.shape == s.shape and torch.allclose(o, s * 2 + 1) for o, s in zip(out, specs))
        q.put(ok)
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
