"""torch.distributed plumbing for the read-sharded multi-GPU path (one process per GPU; backend nccl == RCCL on
ROCm, gloo in the CPU tests).  Reads are independent (GanonClassify.cpp:676-831), the filter is replicated, so the
data path needs no collective: ranks only agree on the timing window and add up their counters
(the reference sums per-thread Total/Rep the same way, GanonClassify.cpp:197-246,475-490)."""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous shard [lo, hi) of n items for `rank` (keeps input order when shards are concatenated)"""
    return (n * rank) // world, (n * (rank + 1)) // world


def init(backend: str, device=None) -> None:
    import socket
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        # stand-alone process (no launcher): a world of one on a free port -- the collectives still go through RCCL
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, **kw)


def group_size() -> int:
    """ranks in the initialised process group (1 without one): what a record's n_gpus is taken from"""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier() -> None:
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: int, device="cpu") -> int:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(x)
    t = torch.tensor([x], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def all_gather_float(x: float, device="cpu"):
    """every rank's value, by rank"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def all_gather_text(text: str, device="cpu", width: int = 96):
    """every rank's short string, by rank (fixed-width byte tensors: the same call on RCCL and gloo, no pickling)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [text]
    raw = text.encode()[:width].ljust(width, b"\0")
    t = torch.tensor(list(raw), dtype=torch.uint8, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [bytes(o.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for o in out]
