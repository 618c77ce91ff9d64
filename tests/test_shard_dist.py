"""N>1 path on CPU: two gloo ranks shard one synthetic batch, evaluate their slabs with the CPU oracle (the
checker — the GPU engine is exercised by the -m gpu tests) and all-reduce the action counters."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pingoo_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_every_batch_size():
    for n in [0, 1, 63, 64, 65, 127, 128, 1000, 4096, 100001]:
        for world in [1, 2, 3, 4, 8]:
            edges = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            for lo, hi in edges:
                assert lo % 64 == 0 or lo == n
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 64 or n < 64 * world
    with pytest.raises(ValueError):
        shard.shard_bounds(10, 2, 2)


def test_library_slab_split_equals_the_python_one():
    """pwaf_node_evaluate_batch (one process, N devices) cuts a batch exactly like the process-per-GPU mode does."""
    from pingoo_amd.engine import node_shard_bounds

    for n in [0, 1, 63, 64, 65, 127, 128, 1000, 4096, 100001, 10_000_000, 2 ** 32 - 1]:
        for world in [1, 2, 3, 4, 8]:
            for r in range(world):
                assert node_shard_bounds(n, r, world) == shard.shard_bounds(n, r, world), (n, r, world)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import pyoracle
    from synth import pysynth

    r, w, _ = shard.init_process_group("gloo")
    assert (r, w) == (rank, world)
    wl = pysynth.Workload(0)
    lo, hi = shard.shard_bounds(n, rank, world)
    batch = wl.batch(lo, hi - lo, threads=1)  # each rank generates only its own slab of the global request stream
    v = pyoracle.Oracle(wl.rules, wl.lists, wl.geoip).evaluate(batch)
    counts = torch.from_numpy(np.bincount(v["action"], minlength=4).astype(np.int64))
    local = counts.clone()
    shard.allreduce_counts(counts)
    dist.barrier()
    q.put((rank, lo, hi, local.tolist(), counts.tolist()))
    dist.destroy_process_group()


def test_two_rank_gloo_counter_allreduce_matches_single_process():
    from oracle import pyoracle
    from synth import pysynth

    n, world = 3000, 2
    wl = pysynth.Workload(0)
    whole = pyoracle.Oracle(wl.rules, wl.lists, wl.geoip).evaluate(wl.batch(0, n, threads=1))
    want = np.bincount(whole["action"], minlength=4).tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n
    assert [a + b for a, b in zip(res[0][3], res[1][3])] == want
    assert res[0][4] == want and res[1][4] == want
    assert want[0] > 0 and sum(want[1:]) > 0


def test_bench_self_launch_becomes_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` started plainly re-executes itself under torch.distributed.run on 127.0.0.1 (the driver's own
    command line, so that both ways of starting it run the same N ranks)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(bench.os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv))
    bench.self_launch(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4)
    argv = seen["argv"]
    assert seen["exe"] == sys.executable and argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    script = argv.index(os.path.abspath(bench.__file__))
    assert argv[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def _single_rank_worker(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from pingoo_amd import shard

    before = shard.collective_world()
    shard.init_process_group("gloo")  # what torch.distributed.run --nproc-per-node 1 sets up: a group of one rank
    t = torch.tensor([5, 6, 7, 8], dtype=torch.int64)
    shard.allreduce_counts(t)
    q.put((before, shard.collective_world(), dist.is_initialized(), t.tolist()))
    dist.destroy_process_group()


def test_world_size_one_under_a_launcher_still_runs_the_collective():
    """with RANK in the environment (torch.distributed.run --nproc-per-node 1) a process group exists and the
    counters go through dist.all_reduce — `rccl_ranks: 1` then means a collective ran; a plain process has none (collective_world 0)."""
    from pingoo_amd import shard

    assert shard.collective_world() == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(29900 + (os.getpid() % 90), q))
    p.start()
    before, world, inited, vals = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert before == 0 and world == 1 and inited and vals == [5, 6, 7, 8]


def test_bench_slabs_weak_and_strong():
    """bench.py --scaling strong (the default when --gpus > 1) splits ONE batch (BASELINE configs[3]: "10M-request batch ... sharded
    2/4/8 GPUs") into the same 64-aligned slabs the node API uses; weak gives every rank its own batch of the global seeded stream."""
    sys.path.insert(0, ROOT)
    import bench

    for world in (1, 2, 4, 8):
        for n_arg in (10_000_000, 1_000_003, 65, 64):
            slabs = [bench.slab_of("strong", n_arg, r, world) for r in range(world)]
            assert all(t == n_arg for _, t, _ in slabs)
            assert slabs[0][2] == 0 and sum(n for n, _, _ in slabs) == n_arg
            for (n0, _, f0), (_, _, f1) in zip(slabs, slabs[1:]):
                assert f0 + n0 == f1 and f1 % 64 == 0  # contiguous, whole 64-request groups
            assert [(n, f) for n, _, f in slabs] == [(hi - lo, lo) for lo, hi in (shard.shard_bounds(n_arg, r, world) for r in range(world))]
            weak = [bench.slab_of("weak", n_arg, r, world) for r in range(world)]
            assert weak == [(n_arg, n_arg * world, r * n_arg) for r in range(world)]
