"""Bin-range partitioning of a flat IBF that does not fit one GPU (BASELINE.json configs[4]).

The IBF is cut by technical-bin range AT TARGET BOUNDARIES into column slices: rank g keeps words
[word_lo, word_hi) of every row (`S x W_g`, rows re-laid-out contiguously) and owns the targets whose bins lie
inside.  Every rank hashes every read (recomputing 18 minimisers is cheaper than broadcasting them), counts only its
own columns and applies the per-read cutoff locally -- valid because a target's bins never straddle two ranks and
the cutoff of select_matches is per target (/root/reference/src/ganon-classify/GanonClassify.cpp:516-527).
The only exchange step is one variable-size all-to-all of sparse (read, target, count) records to the rank that
owns the read (contiguous read ranges), over RCCL/xGMI with the `nccl` backend (gloo in the CPU tests).
A boundary word shared by two ranks is simply held by both (8 bytes per row); the foreign bins in it are mapped to
"no target" locally.  HIBFs cannot be column-sliced this way: replicas only.

`torch.distributed` is plumbing here (rank discovery + the collective); the hot path stays behind the C ABI.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

NO_TARGET = 0xFFFFFFFF
MATCH_DTYPE = np.dtype([("read", "<u4"), ("target", "<u4"), ("count", "<u4")])


@dataclass
class Slice:
    rank: int
    word_lo: int            # first 64-bin word held by this rank
    word_hi: int            # one past the last word
    bins_local: int         # bin count of the local IBF
    bin2target_local: np.ndarray  # local bin -> local target id (NO_TARGET for foreign/unassigned bins)
    targets_global: np.ndarray    # local target id -> caller's (global) target id


def plan_partition(bin2target: np.ndarray, n_bins: int, world: int) -> List[Slice]:
    """Split bins [0, n_bins) into `world` contiguous ranges of roughly equal width whose borders fall between
    targets.  Raises if a target's bins are not contiguous enough to be owned by one rank."""
    bin2target = np.ascontiguousarray(bin2target, dtype=np.uint32)
    assert len(bin2target) == n_bins
    # candidate cut points: bins b where no target spans (b-1, b)
    first, last = {}, {}
    for b, t in enumerate(bin2target.tolist()):
        if t == NO_TARGET:
            continue
        first.setdefault(t, b)
        last[t] = b
    open_until = np.zeros(n_bins + 1, dtype=np.int64)  # a cut at b is illegal if some target has first < b <= last
    for t, f in first.items():
        if last[t] > f:
            open_until[f + 1] += 1
            open_until[last[t] + 1] -= 1
    illegal = np.cumsum(open_until)[: n_bins + 1] > 0
    cuts = [0]
    for g in range(1, world):
        want = (n_bins * g) // world
        lo, hi = want, want
        while lo > cuts[-1] and illegal[lo]:
            lo -= 1
        while hi < n_bins and illegal[hi]:
            hi += 1
        cand = [c for c in (lo, hi) if cuts[-1] <= c <= n_bins and not illegal[c]]
        if not cand:
            raise ValueError("no legal bin-range cut: a target's bins span the whole filter")
        cuts.append(min(cand, key=lambda c: abs(c - want)))
    cuts.append(n_bins)
    slices = []
    for g in range(world):
        b_lo, b_hi = cuts[g], cuts[g + 1]
        word_lo, word_hi = b_lo // 64, (b_hi + 63) // 64 if b_hi > b_lo else b_lo // 64
        if word_hi == word_lo:  # empty range: keep one word so that the local filter is well formed
            word_hi = min(word_lo + 1, (n_bins + 63) // 64)
            word_lo = word_hi - 1
        bins_local = min(n_bins, word_hi * 64) - word_lo * 64
        local = np.full(bins_local, NO_TARGET, dtype=np.uint32)
        owned = bin2target[b_lo:b_hi]
        tg = np.unique(owned[owned != NO_TARGET])
        remap = {int(t): i for i, t in enumerate(tg.tolist())}
        for b in range(b_lo, b_hi):
            t = int(bin2target[b])
            if t != NO_TARGET:
                if first[t] < b_lo or last[t] >= b_hi:
                    raise ValueError(f"target {t} straddles the cut at bin {b_lo}/{b_hi}")
                local[b - word_lo * 64] = remap[t]
        slices.append(Slice(g, word_lo, word_hi, bins_local, local, tg.astype(np.uint32)))
    return slices


def slice_rows(rows: np.ndarray, bin_words: int, sl: Slice) -> np.ndarray:
    """rows: uint64 [S, W] -> contiguous [S, W_g] column slice of this rank."""
    rows = rows.reshape(-1, bin_words)
    return np.ascontiguousarray(rows[:, sl.word_lo:sl.word_hi])


def read_owner_ranges(n_reads: int, world: int) -> np.ndarray:
    """contiguous read ranges: rank g owns reads [r[g], r[g+1])"""
    return np.array([(n_reads * g) // world for g in range(world + 1)], dtype=np.int64)


def exchange_matches(local: np.ndarray, n_reads: int, rank: int, world: int, device: str = "cpu", group=None) -> np.ndarray:
    """local: MATCH_DTYPE records with GLOBAL target ids, grouped by read (ascending).  One variable-size
    all-to-all sends every record to the owner of its read; returns the records of the reads this rank owns, sorted
    by (read, target)."""
    import torch
    import torch.distributed as dist

    ranges = read_owner_ranges(n_reads, world)
    reads = local["read"].astype(np.int64)
    bounds = np.searchsorted(reads, ranges, side="left")  # records are already ordered by read
    send_counts = np.diff(bounds).astype(np.int64)
    flat = np.ascontiguousarray(local).view(np.uint32).reshape(-1, 3).astype(np.int64)
    send = torch.from_numpy(flat).to(device)
    sc = torch.from_numpy(send_counts).to(device)
    rc = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.cpu().numpy()
    recv = torch.empty((int(recv_counts.sum()), 3), dtype=torch.int64, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=[int(x) for x in recv_counts],
                           input_split_sizes=[int(x) for x in send_counts], group=group)
    got = recv.cpu().numpy()
    out = np.zeros(len(got), dtype=MATCH_DTYPE)
    if len(got):
        out["read"], out["target"], out["count"] = got[:, 0], got[:, 1], got[:, 2]
        out = out[np.lexsort((out["target"], out["read"]))]
    return out


LocalClassify = Callable[[np.ndarray, int, int, int, np.ndarray, int, np.ndarray, np.ndarray, Optional[np.ndarray], int, int, float],
                         Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]


def hip_local_classify(device_index: int) -> LocalClassify:
    """The product's local step: this rank's column slice behind the C ABI."""
    from . import HipFilter, HipStream

    cache = {}

    def run(rows, bins, bin_size, hash_funs, bin2target, n_targets, bases, off1, off2, k, w, rel_cutoff):
        key = id(rows)
        if key not in cache:
            cache[key] = HipFilter.ibf(rows.reshape(-1), bins, bin_size, hash_funs, bin2target, n_targets, device=device_index)
        st = HipStream(cache[key], len(off1) - 1, max(int(bases.size), 1))
        st.submit(bases, off1, off2, k, w, rel_cutoff)
        out = st.fetch()
        st.destroy()
        return out

    return run


class PartitionedIbf:
    """One rank's share of a bin-range partitioned flat IBF."""

    def __init__(self, rows: np.ndarray, bins: int, bin_size: int, hash_funs: int, bin2target: np.ndarray, rank: int,
                 world: int, local_classify: LocalClassify, comm_device: str = "cpu", group=None):
        self.rank, self.world = rank, world
        self.bin_size, self.hash_funs = bin_size, hash_funs
        self.plan = plan_partition(bin2target, bins, world)
        self.slice = self.plan[rank]
        W = (bins + 63) >> 6
        self.rows_local = slice_rows(rows, W, self.slice)  # in production each rank reads only its columns from disk
        self.local_classify = local_classify
        self.comm_device = comm_device
        self.group = group

    def classify(self, bases: np.ndarray, off1: np.ndarray, off2: Optional[np.ndarray], k: int, w: int, rel_cutoff: float):
        """-> (read_lo, read_hi, n_hashes[all reads], status[all reads], matches of the owned reads)"""
        sl = self.slice
        n_local_targets = max(1, len(sl.targets_global))
        nh, status, mo, m = self.local_classify(self.rows_local, sl.bins_local, self.bin_size, self.hash_funs,
                                                sl.bin2target_local, n_local_targets, bases, off1, off2, k, w, rel_cutoff)
        glob = np.zeros(len(m), dtype=MATCH_DTYPE)
        if len(m):
            glob["read"], glob["count"] = m["read"], m["count"]
            glob["target"] = sl.targets_global[m["target"]]
        n_reads = len(off1) - 1
        mine = exchange_matches(glob, n_reads, self.rank, self.world, self.comm_device, self.group)
        ranges = read_owner_ranges(n_reads, self.world)
        return int(ranges[self.rank]), int(ranges[self.rank + 1]), nh, status, mine
