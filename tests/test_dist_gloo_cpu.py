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


# ---- 8 ranks on one host: every rank finds ITS GPU's NUMA node from a fixture sysfs tree (no KFD: the drm / PCI-order fallback)
PCI = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0", "0000:85:00.0", "0000:95:00.0", "0000:e5:00.0", "0000:f5:00.0"]
NODE_OF = [3, 2, 1, 0, 7, 6, 5, 4]            # GPU i (PCI order) hangs off NUMA node NODE_OF[i] (NPS4 x 2 sockets, scrambled)


def make_fixture_sysfs(root, with_kfd=False):
    """An 8-GPU, 8-NUMA-node /sys: devices/pci.../<addr>/{vendor,class,numa_node}, class/drm/card<k> and renderD<128+k> symlinked to
    them (card numbers deliberately NOT in PCI order), devices/system/node/node<n>/cpulist = 16 CPUs each; plus two non-GPU cards."""
    import os
    for i, addr in enumerate(PCI):
        d = os.path.join(root, "devices/pci0000:00", addr)
        os.makedirs(d)
        for name, val in (("vendor", "0x1002"), ("class", "0x120000"), ("numa_node", str(NODE_OF[i]))):
            with open(os.path.join(d, name), "w") as f:
                f.write(val + "\n")
    extra = {"0000:01:00.0": ("0x1a03", "0x030000", "0"), "0000:02:00.0": ("0x1002", "0x040300", "0")}   # BMC VGA; an AMD audio function
    for addr, (ven, cls, node) in extra.items():
        d = os.path.join(root, "devices/pci0000:00", addr)
        os.makedirs(d)
        for name, val in (("vendor", ven), ("class", cls), ("numa_node", node)):
            with open(os.path.join(d, name), "w") as f:
                f.write(val + "\n")
    drm = os.path.join(root, "class/drm")
    os.makedirs(drm)
    order = [0, 5, 2, 7, 4, 1, 6, 3]              # card k+1 -> GPU order[k]
    os.makedirs(os.path.join(drm, "card0"))
    os.symlink(os.path.join(root, "devices/pci0000:00", "0000:01:00.0"), os.path.join(drm, "card0", "device"))
    os.makedirs(os.path.join(drm, "card0-VGA-1"))
    for k, g in enumerate(order):
        for nm in (f"card{k + 1}", f"renderD{128 + k}"):
            os.makedirs(os.path.join(drm, nm))
            os.symlink(os.path.join(root, "devices/pci0000:00", PCI[g]), os.path.join(drm, nm, "device"))
    os.makedirs(os.path.join(drm, "card9"))
    os.symlink(os.path.join(root, "devices/pci0000:00", "0000:02:00.0"), os.path.join(drm, "card9", "device"))
    for n in range(8):
        d = os.path.join(root, f"devices/system/node/node{n}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as f:
            f.write(f"{16 * n}-{16 * n + 7},{128 + 16 * n}-{128 + 16 * n + 7}\n")
    if with_kfd:                                   # KFD lists the GPUs in PCI order too, behind two CPU nodes
        base = os.path.join(root, "class/kfd/kfd/topology/nodes")
        for n in range(2):
            os.makedirs(os.path.join(base, str(n)))
            with open(os.path.join(base, str(n), "properties"), "w") as f:
                f.write("simd_count 0\ndrm_render_minor 0\n")
        for g in range(8):
            os.makedirs(os.path.join(base, str(2 + g)))
            with open(os.path.join(base, str(2 + g), "properties"), "w") as f:
                f.write(f"simd_count 1024\ndrm_render_minor {128 + order.index(g)}\n")
    return root


def _numa_worker(rank, world, port, sysfs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ttl_amd.driver import gpu_numa_cpus, pin_to_gpu_numa_node
    node, cpus, src = gpu_numa_cpus(rank, sysfs, env={}, with_source=True)        # BEFORE the process group, like bench.py does
    rec = pin_to_gpu_numa_node(rank, world, sysfs, env={}, apply=False)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = ImageShard(rank, world)
    nodes = shard.gather(torch.tensor([float(node), float(min(cpus))], dtype=torch.float64))
    q.put((rank, dict(node=node, src=src, first_cpu=min(cpus), n_cpus=len(cpus), rec=rec, all_nodes=nodes[:, 0].tolist(),
                      all_first=nodes[:, 1].tolist(), seen=shard.ranks_seen())))
    dist.destroy_process_group()


def test_eight_ranks_pick_eight_distinct_numa_nodes_without_kfd(tmp_path):
    """one process per GPU, 8 per node.  On a box whose container hides /sys/class/kfd (the driver's own bench box did)
    the rank -> NUMA node map comes from /sys/class/drm/card*/device ordered by PCI address; 8 gloo ranks each resolve their own
    GPU and the gathered result is 8 distinct nodes / 8 disjoint CPU sets — the map NODE_OF, not card-number order."""
    sysfs = make_fixture_sysfs(str(tmp_path / "sys"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_numa_worker, args=(r, 8, port, sysfs, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(8):
        assert res[r]["node"] == NODE_OF[r] and res[r]["src"] == "drm-pci-order" and res[r]["seen"] == 8
        assert res[r]["first_cpu"] == 16 * NODE_OF[r] and res[r]["n_cpus"] == 16
        assert res[r]["rec"]["numa_node"] == NODE_OF[r] and res[r]["rec"]["applied"] is False
        assert res[r]["rec"].get("source") == "drm-pci-order" or "reason" in res[r]["rec"]      # (this container has < 2 of those CPUs)
        assert res[r]["all_nodes"] == [float(n) for n in NODE_OF]                                # identical on every rank
    assert len({res[r]["node"] for r in range(8)}) == 8 and len({res[r]["first_cpu"] for r in range(8)}) == 8


def test_visible_device_lists_compose_and_unresolvable_ones_do_not_pin(tmp_path):
    """ROCR_VISIBLE_DEVICES filters the physical list, HIP_ / CUDA_VISIBLE_DEVICES index INTO that subset (round-3 advisor: reading
    only one of them pins a rank to another GPU's socket while the record says applied).  UUID entries, out-of-range indices or
    HIP / CUDA lists that disagree -> no pin, with the reason in the record.  KFD, when readable, answers before the drm fallback."""
    from ttl_amd.driver import gpu_numa_cpus, pin_to_gpu_numa_node
    sysfs = make_fixture_sysfs(str(tmp_path / "sys"), with_kfd=True)
    node = lambda lr, env: gpu_numa_cpus(lr, sysfs, env=env, with_source=True)
    assert node(0, {})[0] == NODE_OF[0] and node(0, {})[2] == "kfd"
    assert node(1, {"ROCR_VISIBLE_DEVICES": "4,5,6,7"})[0] == NODE_OF[5]
    assert node(0, {"HIP_VISIBLE_DEVICES": "3"})[0] == NODE_OF[3]
    assert node(0, {"CUDA_VISIBLE_DEVICES": "6,2"})[0] == NODE_OF[6] and node(1, {"CUDA_VISIBLE_DEVICES": "6,2"})[0] == NODE_OF[2]
    assert node(1, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "2,3"})[0] == NODE_OF[7]       # composed, not either alone
    assert node(0, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "2,3", "CUDA_VISIBLE_DEVICES": "2,3"})[0] == NODE_OF[6]
    for env in ({"ROCR_VISIBLE_DEVICES": "GPU-deadbeef00000000"}, {"HIP_VISIBLE_DEVICES": "0,9"}, {"HIP_VISIBLE_DEVICES": "0", "CUDA_VISIBLE_DEVICES": "1"},
                {"ROCR_VISIBLE_DEVICES": "0,1", "HIP_VISIBLE_DEVICES": "2"}):
        n, cpus, why = node(0, env)
        assert n is None and cpus is None and "VISIBLE_DEVICES" in why, (env, why)
        rec = pin_to_gpu_numa_node(0, 1, sysfs, env=env)
        assert rec["applied"] is False and rec["numa_node"] is None and "VISIBLE_DEVICES" in rec["reason"]
    assert node(2, {"HIP_VISIBLE_DEVICES": "0,1"})[0] is None          # more local ranks than visible GPUs
    # numa_node == -1 everywhere (single-socket VM): nothing to pin to, and the record says which sources were tried
    for addr in PCI:
        with open(os.path.join(sysfs, "devices/pci0000:00", addr, "numa_node"), "w") as f:
            f.write("-1\n")
    n, cpus, why = node(0, {})
    assert n is None and "kfd: numa_node -1" in why and "drm-pci-order: numa_node -1" in why


def test_gather_is_identity_for_one_rank():
    t = torch.tensor([1.5, 2.5], dtype=torch.float64)
    assert torch.equal(ImageShard(0, 1).gather(t), t.reshape(1, 2))


def _run_bench_stub(extra, env=None, n=8):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TTL_BENCH_STUB_MS="3")
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):       # bench.py must start its own ranks
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--stub-pipeline", "--steps", "12", "--warmup", "2",
                        "--repeats", "3"] + extra, env=e, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr[-30000:]


def test_bench_control_flow_of_an_eight_rank_run():
    """bench.py --gpus 8 end to end on CPU (round-4 review item 6b): the real self-spawn (torch.distributed.run started by bench.py
    itself), the ranks_seen all-reduce, timed blocks bracketed by barriers with the max over ranks, per-rank rates gathered through
    ImageShard, rank_balance and the ONE JSON line of rank 0 — with a host-only stand-in for the episode pipeline (--stub-pipeline:
    a sleep per image) and gloo.  Only the headline leg runs by default when N > 1."""
    rc, j, err = _run_bench_stub([])
    assert rc == 0 and j is not None, err
    assert j["stub"] is True and j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["scaling"] == "weak"
    assert j["protocol"]["legs"] == ["fp16"] and list(j["legs"]) == ["fp16"]           # N > 1: the headline build alone
    assert len(j["per_rank_value"]) == 8 and j["rank_balance"]["ok"] is True
    assert j["config"]["backend"] == "gloo" and "image-sharded x8" in j["config"]["parallelism"]
    # 8 ranks x 12 steps of 3 ms: whole-job rate = 8 x one rank's; every image of the last timed block is in the accumulator
    assert 0.5 * 8 / 3e-3 < j["value"] <= 8 / 3e-3
    assert abs(j["value"] - sum(j["per_rank_value"])) < 0.25 * j["value"]
    assert j["accuracy_accumulator"]["images"] == 8 * 12
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k


def test_bench_refuses_ranks_with_different_libraries_and_a_wrong_rank_count():
    """The first multi-GPU run must be boring (round-5 review item 7): the N > 1 line repeats the library's sha256 per rank and a run in
    which one rank loaded another file ends non-zero before anything is timed; so does a launch whose rank count is not --gpus."""
    import subprocess
    import sys
    rc, j, err = _run_bench_stub([])
    assert rc == 0 and j["protocol"]["lib_sha256_15_per_rank"] == ["0" * 15] * 8, (rc, err)
    rc, j, err = _run_bench_stub([], {"TTL_BENCH_STUB_LIBSHA": "3:00000000deadbeef"})
    assert rc != 0 and j is None and "different fp16 libraries" in err, (rc, err)
    # launched by hand with 2 ranks while --gpus says 3: every rank refuses (WORLD_SIZE check; the ranks_seen all-reduce is the
    # second line of defence behind it)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "3", "--stub-pipeline", "--steps", "4",
                        "--warmup", "1"], env=dict(e, TTL_BENCH_STUB_MS="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "--gpus 3 but WORLD_SIZE=2" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_flags_a_slow_rank_and_strict_balance_exits_3():
    """One rank 20 % slower than the others: rank_balance.ok is false in the line; with --strict-balance bench.py exits 3 (the
    code travels from rank 0 through the self-spawn), without it the run still succeeds and reports."""
    slow = {"TTL_BENCH_STUB_SLOW": "5:1.2"}
    rc, j, err = _run_bench_stub([], slow)
    assert rc == 0 and j["rank_balance"]["ok"] is False and j["rank_balance"]["slowest_over_median"] < 0.9, (rc, j and j["rank_balance"], err)
    assert min(range(8), key=lambda r: j["per_rank_value"][r]) == 5
    rc, j, err = _run_bench_stub(["--strict-balance"], slow)
    assert rc == 3 and j["rank_balance"]["ok"] is False, (rc, err)
