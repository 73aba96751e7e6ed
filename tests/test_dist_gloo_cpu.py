"""CPU, world_size 2 (gloo): the image-sharded evaluation loop and its single all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ttl_amd.driver import ImageShard, evaluate_sharded, shard_indices, topk_hits

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
    shard = ImageShard(rank, world)           # the object bench.py and eval.test_time_adapt_eval use
    r["ranks_seen"] = shard.ranks_seen()
    r["max"] = float(shard.max(torch.tensor([float(rank + 1)], dtype=torch.float64)).item())
    r["owned"] = [i for i in range(N_ITEMS) if shard.owns(i)] == list(shard.indices(N_ITEMS))
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
    for r in res.values():
        assert r.pop("ranks_seen") == 2 and r.pop("max") == 2.0 and r.pop("owned") is True
    assert res[0] == res[1] == single            # identical on every rank and equal to the unsharded run


def test_view_boxes_depend_on_seed_and_global_index_only():
    """GpuAugMixAugmenter(seed=...): the crop boxes of test item i are a function of (seed, i) — the same whether one
    rank walks the whole dataset or several ranks walk their shards in any order."""
    from ttl_amd.views import GpuAugMixAugmenter
    aug = GpuAugMixAugmenter(n_views=7, seed=11)
    whole = [aug.boxes(375, 500, i) for i in range(10)]
    for world in (2, 3):
        for rank in range(world):
            other = GpuAugMixAugmenter(n_views=7, seed=11)
            for i in reversed(list(ImageShard(rank, world).indices(10))):      # another order, another object
                assert torch.equal(other.boxes(375, 500, i), whole[i])
    assert not torch.equal(whole[0], whole[1])
    assert not torch.equal(GpuAugMixAugmenter(n_views=7, seed=12).boxes(375, 500, 0), whole[0])
    assert all(int(b[0, 4]) == 2 for b in whole)                               # view 0 stays the base view
