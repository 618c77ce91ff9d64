"""Multi-GPU sharding of request batches: one process per GPU, no data-path collective.

Requests are independent and the compiled tables are immutable (pingoo/server.rs:40-47,76: rules, lists and
GeoIP are loaded once and shared read-only), so every rank holds a full replica of the tables and evaluates a
contiguous slab of the request stream. The only exchange is the all-reduce of the four action counters
(RCCL over xGMI through torch.distributed's "nccl" backend on GPUs; "gloo" in CPU tests). Verdict arrays are
NOT gathered: each shard's verdicts go back to its own host buffers.
"""
from __future__ import annotations

import os
from typing import Tuple

GROUP = 64  # requests per bit-column word on the device: slabs are aligned to it so no group straddles two ranks


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of rank `rank` out of `world`: balanced to within one 64-request group."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    groups = (n + GROUP - 1) // GROUP
    lo_g = groups * rank // world
    hi_g = groups * (rank + 1) // world
    return min(n, lo_g * GROUP), min(n, hi_g * GROUP)


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment; (0, 1, 0) when absent."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str | None = None, always: bool = False):
    """Initialises torch.distributed from the env. nccl == RCCL on ROCm. A process group is created whenever the process was started
    by torch.distributed.run (RANK is set) — also at world size 1, so that the counters' all-reduce is a collective that really runs —
    or when `always` asks for one (a plain single process: rank 0 of a world of 1 on a free local port)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank()
    if dist.is_initialized():
        return rank, world, local
    if world == 1 and "RANK" not in os.environ and not always:
        return rank, world, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world == 1:
            import socket

            with socket.socket() as s:  # (a single process without a launcher picks its own free port)
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        else:
            # ranks started by hand (RANK / WORLD_SIZE set, no launcher) must agree on ONE port — a random one per rank never meets, a fixed
            # default lets two jobs on one host collide or cross-join: derived from a job id every rank shares, else refused
            job = os.environ.get("PWAF_JOB_ID") or os.environ.get("SLURM_JOB_ID") or os.environ.get("TORCHELASTIC_RUN_ID")
            if not job:
                raise RuntimeError(f"rank {rank} of {world} was started without a launcher and without MASTER_PORT: set MASTER_PORT (the same on every rank), "
                                   "or PWAF_JOB_ID to derive one, or start the ranks with `python -m torch.distributed.run --master-port P`")
            import zlib

            os.environ["MASTER_PORT"] = str(20000 + zlib.crc32(job.encode()) % 20000)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_counts(counts):
    """Sums the per-rank action counters (int64 tensor [4] on the rank's device) in place across ranks: one collective per call
    whenever a process group exists (world size 1 included: the sum of one rank's counters is those counters)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def collective_world() -> int:
    """The world size the counters' all-reduce runs over; 0 = no process group, no collective is called."""
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0
