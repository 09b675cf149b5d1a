"""Bin-range partitioning of a flat IBF that does not fit one GPU (BASELINE.json configs[4]).

The IBF is cut by technical-bin range AT TARGET BOUNDARIES into column slices: rank g keeps words
[word_lo, word_hi) of every row (`S x W_g`, rows re-laid-out contiguously) and owns the targets whose bins lie
inside.  Every rank hashes every read (recomputing 18 minimisers is cheaper than broadcasting them), counts only its
own columns and applies the per-read cutoff locally -- valid because a target's bins never straddle two ranks and
the cutoff of select_matches is per target (/root/reference/src/ganon-classify/GanonClassify.cpp:516-527).
The only exchange step is one variable-size all-to-all of sparse (read, target, count) records (12 bytes each) to
the rank that owns the read (contiguous read ranges), over RCCL/xGMI with the `nccl` backend (gloo in the CPU
tests), next to a fixed-size all-to-all of the owners' per-read offsets.  With a HIP local filter the records never
visit the host: RCCL sends straight from the library's device match buffer (wrapped as a tensor, no copy), the
per-owner split points are read off the library's device offsets, and the owner concatenates what it received per
read, source rank after source rank -- ascending target, because slices ascend with the rank -- with the same kernel
the single-process product uses (gn_gather_run_buffers), which also rewrites part-local target ids.  One host
synchronisation per batch: the world x world count matrix (torch wants split sizes as Python ints).
This module is the ONE-PROCESS-PER-GPU form of the partition (bench.py under torchrun); `ganon-classify` does the same
inside one process with hipMemcpyPeerAsync between its devices (host/backend_hip.cpp, gn_gather_run).
A boundary word shared by two ranks is simply held by both (8 bytes per row); the foreign bins in it are mapped to
"no target" locally.  HIBFs cannot be column-sliced this way: replicas only.

A rank never needs the whole matrix: `PartitionedIbf` takes an already-local filter (loaded column-wise with
gn_filter_write_rows, or filled on the device); `from_host_rows` slices a host matrix for small tests.
`torch.distributed` is plumbing here (rank discovery + the collective); the hot path stays behind the C ABI.
"""
from __future__ import annotations

import time
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


def merge_parts_numpy(offs, recs, maps, read_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """numpy statement of gn_gather (csrc/gn_gather.hip): part i holds, for every owned read r, the records
    recs[i][offs[i][r]-offs[i][0] : offs[i][r+1]-offs[i][0]] (grouped by read, ascending target, part-local target ids);
    the result is read after read the parts' segments behind each other, target ids mapped through maps[i].
    -> (match_off u64[n+1], records MATCH_DTYPE)"""
    n = len(offs[0]) - 1
    moff = np.zeros(n + 1, dtype=np.uint64)
    chunks = []
    for o, r, mp in zip(offs, recs, maps):
        o = np.asarray(o, dtype=np.int64) - int(o[0])
        moff += o.astype(np.uint64)
        g = np.zeros(len(r), dtype=MATCH_DTYPE)
        if len(r):
            g["read"], g["count"] = r["read"], r["count"]
            g["target"] = r["target"] if mp is None else np.asarray(mp, dtype=np.uint32)[r["target"]]
        chunks.append(g)
    allr = np.concatenate(chunks) if chunks else np.zeros(0, MATCH_DTYPE)
    order = np.argsort(allr["read"], kind="stable")  # (parts were appended in order: a stable sort keeps part order per read)
    return moff, allr[order]


def exchange_grouped(off, rec, n_reads: int, rank: int, world: int, group=None):
    """The exchange step.  off: int64 tensor [n_reads+1], rec: int32 tensor [m, 3] = this rank's matches of ALL reads grouped
    by read (part-local target ids), both on the communication device.  Every record goes to the rank that owns its read
    (contiguous read ranges); the owner gets, per source rank, the records and the per-read offsets of its reads.
      1. counts: one all-gather of the world x world matrix "records from rank s for the reads of rank d" -- its copy to
         the host is the ONLY host synchronisation of the step (torch's all-to-all takes its split sizes as Python ints);
      2. offsets: all-to-all of off[lo_d .. hi_d] per owner d (sizes follow from the read ranges alone);
      3. records: all-to-all straight from the caller's buffer (records are ordered by read, hence by owner).
    -> (parts_off [world tensors of n_own+1], parts_rec [world tensors of c_s x 3])"""
    import torch
    import torch.distributed as dist

    ranges = read_owner_ranges(n_reads, world)
    dev = off.device
    bounds = off[torch.from_numpy(ranges).to(dev)]
    cnt = (bounds[1:] - bounds[:-1]).contiguous()
    if world > 1:
        rows = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(rows, cnt, group=group)
        mat = torch.stack(rows).cpu().numpy()          # [source][owner]
    else:
        mat = cnt.cpu().numpy().reshape(1, 1)
    send_counts = [int(x) for x in mat[rank]]
    recv_counts = [int(x) for x in mat[:, rank]]
    n_own = int(ranges[rank + 1] - ranges[rank])
    send_off = torch.cat([off[int(ranges[d]):int(ranges[d + 1]) + 1] for d in range(world)]).contiguous()
    recv_off = torch.empty(world * (n_own + 1), dtype=off.dtype, device=dev)
    recv = torch.empty((sum(recv_counts), 3), dtype=rec.dtype, device=dev)
    if dist.is_initialized():  # (a world of one still goes through the collective: same calls, same buffers)
        dist.all_to_all_single(recv_off, send_off, output_split_sizes=[n_own + 1] * world,
                               input_split_sizes=[int(ranges[d + 1] - ranges[d]) + 1 for d in range(world)], group=group)
        dist.all_to_all_single(recv, rec.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
    else:
        recv_off.copy_(send_off)
        recv.copy_(rec)
    parts_off = [recv_off[s * (n_own + 1):(s + 1) * (n_own + 1)] for s in range(world)]
    starts = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    parts_rec = [recv[int(starts[s]):int(starts[s + 1])] for s in range(world)]
    return parts_off, parts_rec


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

    def device_offsets(self):
        return self.st.device_offsets(self.dev)

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
        self.exchange_ms: List[float] = []   # per classify(): the exchange step (collectives + the sync before the merge), wall clock
        self.merge_ms: List[float] = []      # per classify(): merge of what arrived on the owner
        self._maps = None     # every rank's local -> global target table (exchanged once)
        self._gather = None   # device merge (gn_gather) of what the exchange delivered
        self._last = None

    @classmethod
    def from_host_rows(cls, rows: np.ndarray, bins: int, bin_size: int, hash_funs: int, bin2target: np.ndarray, rank: int,
                       world: int, make_local: MakeLocal, comm_device: str = "cpu", group=None) -> "PartitionedIbf":
        """small filters / tests: slice a host matrix (a production rank loads only its columns)"""
        sl = plan_partition(bin2target, bins, world)[rank]
        rows_local = slice_rows(rows, (bins + 63) >> 6, sl)
        local = make_local(rows_local, sl.bins_local, bin_size, hash_funs, sl.bin2target_local, max(1, len(sl.targets_global)))
        return cls(sl, rank, world, local, comm_device, group)

    def _target_maps(self):
        """the owner rewrites part-local target ids while it merges (gn_gather's target_map): it needs every rank's table"""
        if self._maps is None:
            import torch.distributed as dist
            mine = np.ascontiguousarray(self.slice.targets_global, dtype=np.uint32)
            if self.world > 1:
                tables = [None] * self.world
                dist.all_gather_object(tables, mine, group=self.group)
                self._maps = [np.asarray(t, dtype=np.uint32) for t in tables]
            else:
                self._maps = [mine]
        return self._maps

    def classify(self, bases: np.ndarray, off1: np.ndarray, off2: Optional[np.ndarray], k: int, w: int, rel_cutoff: float,
                 fetch: bool = True):
        """-> (read_lo, read_hi, n_hashes[all reads], status[all reads], matches of the owned reads).  With fetch=False the
        last three are None and the owned matches stay in device memory (fetch_owned() copies them out later)."""
        import torch
        n_reads = len(off1) - 1
        out = self.local.classify(bases, off1, off2, k, w, rel_cutoff)
        ranges = read_owner_ranges(n_reads, self.world)
        lo, hi = int(ranges[self.rank]), int(ranges[self.rank + 1])
        maps = self._target_maps()
        rec = self.local.device_records() if self.comm_device != "cpu" else None
        if rec is not None:
            # device-resident: the library's own match buffer and offsets are what RCCL sends; what arrives is merged per
            # read by gn_gather_run_buffers on this rank's GPU
            off = self.local.device_offsets()
            t_ex = time.perf_counter()
            parts_off, parts_rec = exchange_grouped(off, rec, n_reads, self.rank, self.world, self.group)
            torch.cuda.current_stream().synchronize()   # the receive buffers are complete before the library reads them
            t_mg = time.perf_counter()
            self.exchange_ms.append((t_mg - t_ex) * 1e3)
            if self._gather is None:
                from . import HipGather
                self._gather = HipGather(rec.device.index or 0, [m if len(m) else None for m in maps])
            self._keep = (parts_off, parts_rec)
            self._gather.run_buffers([t.data_ptr() for t in parts_off], [t.data_ptr() if t.numel() else 0 for t in parts_rec],
                                     [t.shape[0] for t in parts_rec], hi - lo)
            self.merge_ms.append((time.perf_counter() - t_mg) * 1e3)
            self._last = ("device", out)
            if not fetch:
                return lo, hi, None, None, None
            nh, status = out.fetch_read_info()
            return lo, hi, nh, status, self.fetch_owned()
        nh, status, mo, m = out.fetch() if hasattr(out, "fetch") else out
        off = torch.from_numpy(np.ascontiguousarray(mo).astype(np.int64)).to(self.comm_device)
        flat = np.ascontiguousarray(m).view(np.uint32).reshape(-1, 3).view(np.int32)
        t_ex = time.perf_counter()
        parts_off, parts_rec = exchange_grouped(off, torch.from_numpy(flat).to(self.comm_device), n_reads, self.rank, self.world, self.group)
        t_mg = time.perf_counter()
        self.exchange_ms.append((t_mg - t_ex) * 1e3)
        offs = [t.cpu().numpy() for t in parts_off]
        recs = [np.ascontiguousarray(t.cpu().numpy()).view(np.uint32).reshape(-1, 3).view(MATCH_DTYPE).reshape(-1) for t in parts_rec]
        _, mine = merge_parts_numpy(offs, recs, [mp if len(mp) else None for mp in maps])
        self.merge_ms.append((time.perf_counter() - t_mg) * 1e3)
        self._last = ("host", mine)
        return lo, hi, nh, status, mine

    def fetch_owned(self) -> np.ndarray:
        """the owned reads' matches of the last classify() (global read indices, global target ids)"""
        kind, what = self._last
        if kind == "host":
            return what
        _, m = self._gather.fetch()
        return m.copy()

    def close(self):
        if self._gather is not None:
            self._gather.destroy()
            self._gather = None
        self.local.close()
