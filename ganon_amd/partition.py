"""Bin-range partitioning of a flat IBF that does not fit one GPU (BASELINE.json configs[4]).

The IBF is cut by technical-bin range AT TARGET BOUNDARIES into column slices: rank g keeps words
[word_lo, word_hi) of every row (`S x W_g`, rows re-laid-out contiguously) and owns the targets whose bins lie
inside.  Every rank hashes every read (recomputing 18 minimisers is cheaper than broadcasting them), counts only its
own columns and applies the per-read cutoff locally -- valid because a target's bins never straddle two ranks and
the cutoff of select_matches is per target (/root/reference/src/ganon-classify/GanonClassify.cpp:516-527).
The only exchange step is one variable-size all-to-all of sparse (read, target, count) records (12 bytes each) to
the rank that owns the read (contiguous read ranges), over RCCL/xGMI with the `nccl` backend (gloo in the CPU
tests).  With a HIP local filter the records never visit the host on the way: the library's device match buffer
is wrapped as a tensor, target ids are made global and the per-owner split points are found on the device, RCCL
sends straight from HBM, and the owner sorts what it received on the device.
A boundary word shared by two ranks is simply held by both (8 bytes per row); the foreign bins in it are mapped to
"no target" locally.  HIBFs cannot be column-sliced this way: replicas only.

A rank never needs the whole matrix: `PartitionedIbf` takes an already-local filter (loaded column-wise with
gn_filter_write_rows, or filled on the device); `from_host_rows` slices a host matrix for small tests.
`torch.distributed` is plumbing here (rank discovery + the collective); the hot path stays behind the C ABI.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

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
    # first / last bin of every target (vectorised: filters of config 5 have 10^5..10^6 bins)
    valid = np.nonzero(bin2target != NO_TARGET)[0]
    tv = bin2target[valid].astype(np.int64)
    n_t = int(tv.max()) + 1 if len(tv) else 0
    first = np.full(n_t, n_bins, dtype=np.int64)
    last = np.full(n_t, -1, dtype=np.int64)
    np.minimum.at(first, tv, valid)
    np.maximum.at(last, tv, valid)
    open_until = np.zeros(n_bins + 2, dtype=np.int64)  # a cut at b is illegal if some target has first < b <= last
    multi = np.nonzero(last > first)[0]
    np.add.at(open_until, first[multi] + 1, 1)
    np.add.at(open_until, last[multi] + 1, -1)
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
        ov = owned != NO_TARGET
        tg = np.unique(owned[ov])
        if len(tg) and (first[tg].min() < b_lo or last[tg].max() >= b_hi):
            bad = tg[(first[tg] < b_lo) | (last[tg] >= b_hi)][0]
            raise ValueError(f"target {int(bad)} straddles the cut at bin {b_lo}/{b_hi}")
        local[b_lo - word_lo * 64: b_hi - word_lo * 64][ov] = np.searchsorted(tg, owned[ov]).astype(np.uint32)
        slices.append(Slice(g, word_lo, word_hi, bins_local, local, tg.astype(np.uint32)))
    return slices


def slice_rows(rows: np.ndarray, bin_words: int, sl: Slice) -> np.ndarray:
    """rows: uint64 [S, W] -> contiguous [S, W_g] column slice of this rank."""
    rows = rows.reshape(-1, bin_words)
    return np.ascontiguousarray(rows[:, sl.word_lo:sl.word_hi])


def read_owner_ranges(n_reads: int, world: int) -> np.ndarray:
    """contiguous read ranges: rank g owns reads [r[g], r[g+1])"""
    return np.array([(n_reads * g) // world for g in range(world + 1)], dtype=np.int64)


def _all_to_all_records(send, send_counts, world: int, group):
    """send: int32 tensor [m, 3] ordered by destination rank; send_counts: int64 tensor [world] on the comm device"""
    import torch
    import torch.distributed as dist

    rc = torch.empty(world, dtype=torch.int64, device=send.device)
    dist.all_to_all_single(rc, send_counts.to(send.device), group=group)
    recv_counts = [int(x) for x in rc.cpu().tolist()]
    recv = torch.empty((sum(recv_counts), 3), dtype=torch.int32, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts,
                           input_split_sizes=[int(x) for x in send_counts.cpu().tolist()], group=group)
    return recv


def _sorted_records(recv) -> np.ndarray:
    """int32 [m, 3] tensor -> MATCH_DTYPE records sorted by (read, target); the sort runs where the tensor lives"""
    import torch

    if recv.shape[0]:
        key = (recv[:, 0].to(torch.int64) & 0xFFFFFFFF) << 32 | (recv[:, 1].to(torch.int64) & 0xFFFFFFFF)
        recv = recv[torch.argsort(key)]
    got = np.ascontiguousarray(recv.cpu().numpy())
    return got.view(np.uint32).reshape(-1, 3).view(MATCH_DTYPE).reshape(-1).copy()


def exchange_matches(local: np.ndarray, n_reads: int, rank: int, world: int, device: str = "cpu", group=None) -> np.ndarray:
    """local: MATCH_DTYPE records with GLOBAL target ids, grouped by read (ascending).  One variable-size
    all-to-all sends every record (12 bytes, as three int32) to the owner of its read; returns the records of the
    reads this rank owns, sorted by (read, target)."""
    import torch

    ranges = read_owner_ranges(n_reads, world)
    reads = local["read"].astype(np.int64)
    bounds = np.searchsorted(reads, ranges, side="left")  # records are already ordered by read
    send_counts = torch.from_numpy(np.diff(bounds).astype(np.int64))
    flat = np.ascontiguousarray(local).view(np.uint32).reshape(-1, 3).view(np.int32)
    send = torch.from_numpy(flat).to(device)
    return _sorted_records(_all_to_all_records(send, send_counts, world, group))


def exchange_matches_device(records, targets_global, n_reads: int, rank: int, world: int, group=None) -> np.ndarray:
    """records: int32 DEVICE tensor [m, 3] (read, LOCAL target, count) ordered by read -- the library's own match
    buffer; targets_global: int32 device tensor (local target -> global).  Remap, split and exchange on the device."""
    import torch

    rec = records.clone()
    if rec.shape[0]:
        rec[:, 1] = targets_global[rec[:, 1].to(torch.int64)]
    ranges = torch.from_numpy(read_owner_ranges(n_reads, world)).to(rec.device)
    bounds = torch.searchsorted(rec[:, 0].contiguous().to(torch.int64), ranges)
    return _sorted_records(_all_to_all_records(rec, bounds[1:] - bounds[:-1], world, group))


class LocalFilter:
    """One rank's column slice as a classifier.  classify() -> (n_hashes, status, match_off, matches with LOCAL
    target ids, grouped by read).  device_records() (optional) -> the same matches as an int32 device tensor."""

    def classify(self, bases, off1, off2, k, w, rel_cutoff):  # pragma: no cover - interface
        raise NotImplementedError

    def device_records(self):
        return None

    def close(self):
        pass


class HipLocalFilter(LocalFilter):
    """The product's local step: this rank's column slice behind the C ABI (an existing HipFilter, or host rows)."""

    def __init__(self, flt, device_index: int = 0, own: bool = False):
        self.flt, self.dev, self.own = flt, device_index, own
        self.st = None
        self._cap = (0, 0)
        self._resident = None

    @classmethod
    def from_rows(cls, rows, bins, bin_size, hash_funs, bin2target, n_targets, device_index: int = 0):
        from . import HipFilter
        return cls(HipFilter.ibf(np.ascontiguousarray(rows).reshape(-1), bins, bin_size, hash_funs, bin2target, n_targets,
                                 device=device_index), device_index, own=True)

    def classify(self, bases, off1, off2, k, w, rel_cutoff):
        from . import HipStream
        n, nb = len(off1) - 1, max(int(bases.size), 1)
        if self.st is None or self._cap[0] < n or self._cap[1] < nb:
            if self.st is not None:
                self.st.destroy()
            self.st = HipStream(self.flt, max(n, 1), nb)
            self._cap = (max(n, 1), nb)
            self._resident = None
        if self._resident is not bases:  # a batch that is already in HBM is not uploaded again
            self.st.upload(bases, off1, off2)
            self._resident = bases
        self.st.classify(k, w, rel_cutoff)
        self.st.sync()
        return self.st

    def device_records(self):
        return self.st.device_records(self.dev)

    def close(self):
        if self.st is not None:
            self.st.destroy()
            self.st = None
        if self.own:
            self.flt.free()


MakeLocal = Callable[[np.ndarray, int, int, int, np.ndarray, int], LocalFilter]


class PartitionedIbf:
    """One rank's share of a bin-range partitioned flat IBF."""

    def __init__(self, sl: Slice, rank: int, world: int, local: LocalFilter, comm_device: str = "cpu", group=None):
        self.rank, self.world = rank, world
        self.slice = sl
        self.local = local
        self.comm_device = comm_device
        self.group = group
        self._tg_dev = None

    @classmethod
    def from_host_rows(cls, rows: np.ndarray, bins: int, bin_size: int, hash_funs: int, bin2target: np.ndarray, rank: int,
                       world: int, make_local: MakeLocal, comm_device: str = "cpu", group=None) -> "PartitionedIbf":
        """small filters / tests: slice a host matrix (a production rank loads only its columns)"""
        sl = plan_partition(bin2target, bins, world)[rank]
        rows_local = slice_rows(rows, (bins + 63) >> 6, sl)
        local = make_local(rows_local, sl.bins_local, bin_size, hash_funs, sl.bin2target_local, max(1, len(sl.targets_global)))
        return cls(sl, rank, world, local, comm_device, group)

    def classify(self, bases: np.ndarray, off1: np.ndarray, off2: Optional[np.ndarray], k: int, w: int, rel_cutoff: float):
        """-> (read_lo, read_hi, n_hashes[all reads], status[all reads], matches of the owned reads)"""
        sl = self.slice
        n_reads = len(off1) - 1
        out = self.local.classify(bases, off1, off2, k, w, rel_cutoff)
        ranges = read_owner_ranges(n_reads, self.world)
        rec = self.local.device_records() if self.comm_device != "cpu" else None
        if rec is not None:
            import torch
            if self._tg_dev is None:
                tg = sl.targets_global if len(sl.targets_global) else np.zeros(1, np.uint32)
                self._tg_dev = torch.from_numpy(tg.view(np.int32).copy()).to(rec.device)
            mine = exchange_matches_device(rec, self._tg_dev, n_reads, self.rank, self.world, self.group)
            nh, status = out.fetch_read_info()
        else:
            nh, status, mo, m = out.fetch() if hasattr(out, "fetch") else out
            glob = np.zeros(len(m), dtype=MATCH_DTYPE)
            if len(m):
                glob["read"], glob["count"] = m["read"], m["count"]
                glob["target"] = sl.targets_global[m["target"]]
            mine = exchange_matches(glob, n_reads, self.rank, self.world, self.comm_device, self.group)
        return int(ranges[self.rank]), int(ranges[self.rank + 1]), nh, status, mine
