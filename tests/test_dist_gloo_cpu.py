"""CPU, world_size 2 (gloo): the image-sharded evaluation loop and its single all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ttl_amd.driver import evaluate_sharded, shard_indices, topk_hits

N_ITEMS, K = 37, 10


def _predict(i):
    g = torch.Generator().manual_seed(i)
    return torch.randn(1, K, generator=g)


def _label(i):
    return (i * 7) % K


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = evaluate_sharded(_predict, N_ITEMS, _label, rank, world, "cpu")
    q.put((rank, r))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shards_partition_the_dataset():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in shard_indices(N_ITEMS, r, world))
        assert seen == list(range(N_ITEMS))
        sizes = [len(shard_indices(N_ITEMS, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_topk_hits_semantics():
    z = torch.tensor([[0.1, 0.9, 0.3, 0.2, 0.0, -1.0, 0.5]])
    h1, h5 = topk_hits(z, torch.tensor([1]))
    assert int(h1) == 1 and int(h5) == 1
    h1, h5 = topk_hits(z, torch.tensor([5]))
    assert int(h1) == 0 and int(h5) == 0
    h1, h5 = topk_hits(z, torch.tensor([3]))
    assert int(h1) == 0 and int(h5) == 1


def test_two_rank_allreduce_equals_single_rank():
    single = evaluate_sharded(_predict, N_ITEMS, _label, 0, 1, "cpu")
    assert single["count"] == N_ITEMS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == single            # identical on every rank and equal to the unsharded run
