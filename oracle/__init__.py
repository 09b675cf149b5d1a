"""ctypes bindings of the CPU oracle (oracle/ganon_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package ``ganon_amd``.  See ganon_oracle.h for the
provenance of every function (reference file:line) and the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libganon_oracle.so")


def build(force: bool = False) -> str:
    """Compile oracle/libganon_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "ganon_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class _IbfS(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bins", C.c_uint64), ("bin_size", C.c_uint64), ("bin_words", C.c_uint64),
                ("hash_shift", C.c_uint64), ("hash_funs", C.c_uint32)]


class _HibfS(C.Structure):
    _fields_ = [("n_ibf", C.c_uint32), ("ibfs", C.POINTER(_IbfS)), ("next_ibf_id", C.POINTER(C.c_void_p)),
                ("bin_to_user", C.POINTER(C.c_void_p)), ("n_user_bins", C.c_uint64)]


class _FilterS(C.Structure):
    _fields_ = [("is_hibf", C.c_int), ("ibf", C.POINTER(_IbfS)), ("hibf", C.POINTER(_HibfS)),
                ("n_targets", C.c_uint32), ("tgt_bin_off", C.c_void_p), ("tgt_bins", C.c_void_p),
                ("tgt_global", C.c_void_p), ("tgt_fpr", C.c_void_p), ("rel_cutoff", C.c_double)]


class _ReadResS(C.Structure):
    _fields_ = [("n_hashes", C.c_uint64), ("max_count_read", C.c_uint64), ("min_count_read", C.c_uint64),
                ("threshold_filter", C.c_uint64), ("n_kept", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.gno_char_to_rank.restype = C.c_uint8
        L.gno_char_to_rank.argtypes = [C.c_ubyte, C.POINTER(C.c_int)]
        L.gno_adjust_seed.restype = C.c_uint64
        L.gno_adjust_seed.argtypes = [C.c_uint32]
        L.gno_minimiser_hash.restype = C.c_size_t
        L.gno_minimiser_hash.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.gno_threshold_rel.restype = C.c_uint64
        L.gno_threshold_rel.argtypes = [C.c_uint64, C.c_double]
        L.gno_threshold_cutoff.restype = C.c_uint64
        L.gno_threshold_cutoff.argtypes = [C.c_uint64, C.c_double]
        L.gno_ibf_hash_shift.restype = C.c_uint64
        L.gno_ibf_hash_shift.argtypes = [C.c_uint64]
        L.gno_ibf_row.restype = C.c_uint64
        L.gno_ibf_row.argtypes = [C.POINTER(_IbfS), C.c_uint64, C.c_uint32]
        L.gno_ibf_emplace.restype = None
        L.gno_ibf_emplace.argtypes = [C.POINTER(_IbfS), C.c_uint64, C.c_uint64]
        L.gno_ibf_emplace_many.restype = None
        L.gno_ibf_emplace_many.argtypes = [C.POINTER(_IbfS), C.c_void_p, C.c_void_p, C.c_size_t]
        L.gno_ibf_bulk_count.restype = None
        L.gno_ibf_bulk_count.argtypes = [C.POINTER(_IbfS), C.c_void_p, C.c_size_t, C.c_void_p]
        L.gno_ibf_bulk_count_gathered.restype = None
        L.gno_ibf_bulk_count_gathered.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        L.gno_hibf_bulk_count.restype = None
        L.gno_hibf_bulk_count.argtypes = [C.POINTER(_HibfS), C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        L.gno_hibf_bulk_count_longreads.restype = None
        L.gno_hibf_bulk_count_longreads.argtypes = [C.POINTER(_HibfS), C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        L.gno_hibf_visited_bytes.restype = C.c_uint64
        L.gno_hibf_visited_bytes.argtypes = [C.POINTER(_HibfS), C.c_void_p, C.c_size_t, C.c_uint64]
        L.gno_binom.restype = C.c_double
        L.gno_binom.argtypes = [C.c_double, C.c_double]
        L.gno_false_positive.restype = C.c_double
        L.gno_false_positive.argtypes = [C.c_uint64, C.c_uint8, C.c_uint64]
        L.gno_target_fpr.restype = C.c_double
        L.gno_target_fpr.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8]
        L.gno_classify_read.restype = C.c_int
        L.gno_classify_read.argtypes = [C.POINTER(_FilterS), C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_ReadResS), C.c_void_p,
                                        C.c_size_t, C.c_void_p]
        L.gno_lca_build.restype = C.c_void_p
        L.gno_lca_build.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.gno_lca_query.restype = C.c_int32
        L.gno_lca_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.gno_lca_free.restype = None
        L.gno_lca_free.argtypes = [C.c_void_p]
        L.gno_baseline_classify.restype = C.c_uint64
        L.gno_baseline_classify.argtypes = [C.POINTER(_FilterS), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                            C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- sequences
_RANK_LUT = None


def rank_lut() -> np.ndarray:
    """256-entry char -> dna4 rank table (values 0..3), built from gno_char_to_rank."""
    global _RANK_LUT
    if _RANK_LUT is None:
        L = lib()
        _RANK_LUT = np.array([L.gno_char_to_rank(c, None) for c in range(256)], dtype=np.uint8)
    return _RANK_LUT


def legal_lut() -> np.ndarray:
    L = lib()
    out = np.zeros(256, dtype=np.uint8)
    for c in range(256):
        ok = C.c_int(0)
        L.gno_char_to_rank(c, C.byref(ok))
        out[c] = ok.value
    return out


def to_ranks(seq) -> np.ndarray:
    """str/bytes (ASCII) or uint8 array of ASCII -> uint8 ranks."""
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        seq = np.frombuffer(bytes(seq), dtype=np.uint8)
    return rank_lut()[np.asarray(seq, dtype=np.uint8)]


def adjust_seed(k: int) -> int:
    return int(lib().gno_adjust_seed(k))


def minimiser_hash(ranks: np.ndarray, k: int, w: int) -> np.ndarray:
    ranks = np.ascontiguousarray(ranks, dtype=np.uint8)
    cap = max(1, len(ranks))
    out = np.empty(cap, dtype=np.uint64)
    n = lib().gno_minimiser_hash(_ptr(ranks), len(ranks), k, w, _ptr(out), cap)
    return out[:n].copy()


def threshold_rel(n: int, p: float) -> int:
    return int(lib().gno_threshold_rel(n, p))


def threshold_cutoff(n: int, p: float) -> int:
    return int(lib().gno_threshold_cutoff(n, p))


# --------------------------------------------------------------------------- IBF
class Ibf:
    """Flat interleaved Bloom filter held in a numpy uint64 array [S, W]."""

    def __init__(self, bins: int, bin_size: int, hash_funs: int, data: Optional[np.ndarray] = None):
        self.bins = int(bins)
        self.bin_size = int(bin_size)
        self.hash_funs = int(hash_funs)
        self.bin_words = (self.bins + 63) >> 6
        self.hash_shift = int(lib().gno_ibf_hash_shift(self.bin_size))
        if data is None:
            data = np.zeros((self.bin_size, self.bin_words), dtype=np.uint64)
        self.data = np.ascontiguousarray(data, dtype=np.uint64).reshape(self.bin_size, self.bin_words)
        self._s = _IbfS(self.data.ctypes.data, self.bins, self.bin_size, self.bin_words, self.hash_shift,
                        self.hash_funs)

    @property
    def technical_bins(self) -> int:
        return self.bin_words * 64

    def cstruct(self) -> _IbfS:
        return self._s

    def row(self, v: int, i: int) -> int:
        return int(lib().gno_ibf_row(C.byref(self._s), int(v), i))

    def emplace(self, v: int, b: int) -> None:
        lib().gno_ibf_emplace(C.byref(self._s), int(v), int(b))

    def emplace_many(self, hashes: np.ndarray, b) -> None:
        """b: one bin for all hashes, or an array of bins (one per hash)."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        bins = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.uint32), hashes.shape))
        lib().gno_ibf_emplace_many(C.byref(self._s), _ptr(hashes), _ptr(bins), len(hashes))

    def bulk_count(self, hashes: np.ndarray) -> np.ndarray:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        counts = np.zeros(max(self.bins, 1), dtype=np.uint16)
        lib().gno_ibf_bulk_count(C.byref(self._s), _ptr(hashes), len(hashes), _ptr(counts))
        return counts[: self.bins]


class SampledIbf:
    """A flat IBF too large for the host: rows are fetched on demand through `fetch_rows(row_idx uint64[m]) ->
    uint64[m, W]` (the device's gn_filter_download_row_list in the full-size tests).  Row selection is gno_ibf_row,
    counting is gno_ibf_bulk_count_gathered -- the same statements as Ibf.bulk_count."""

    def __init__(self, bins: int, bin_size: int, hash_funs: int, fetch_rows):
        self.bins, self.bin_size, self.hash_funs = int(bins), int(bin_size), int(hash_funs)
        self.bin_words = (self.bins + 63) >> 6
        self.hash_shift = int(lib().gno_ibf_hash_shift(self.bin_size))
        self._s = _IbfS(None, self.bins, self.bin_size, self.bin_words, self.hash_shift, self.hash_funs)
        self._fetch = fetch_rows

    def rows_of(self, hashes: np.ndarray) -> np.ndarray:
        L = lib()
        return np.array([L.gno_ibf_row(C.byref(self._s), int(v), i) for v in hashes for i in range(self.hash_funs)],
                        dtype=np.uint64)

    def bulk_count(self, hashes: np.ndarray) -> np.ndarray:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        counts = np.zeros(max(self.bins, 1), dtype=np.uint16)
        if len(hashes):
            g = np.ascontiguousarray(self._fetch(self.rows_of(hashes)), dtype=np.uint64)
            assert g.shape == (len(hashes) * self.hash_funs, self.bin_words)
            lib().gno_ibf_bulk_count_gathered(_ptr(g), len(hashes), self.hash_funs, self.bin_words, self.bins, _ptr(counts))
        return counts[: self.bins]


class Hibf:
    """raptor-style HIBF: list of Ibf + next_ibf_id + bin->user-bin tables."""

    def __init__(self, ibfs: Sequence[Ibf], next_ibf_id: Sequence[Sequence[int]],
                 bin_to_user: Sequence[Sequence[int]], n_user_bins: int):
        self.ibfs = list(ibfs)
        self.next_ibf_id = [np.ascontiguousarray(x, dtype=np.int64) for x in next_ibf_id]
        self.bin_to_user = [np.ascontiguousarray(x, dtype=np.int64) for x in bin_to_user]
        self.n_user_bins = int(n_user_bins)
        n = len(self.ibfs)
        self._ibf_arr = (_IbfS * n)(*[f.cstruct() for f in self.ibfs])
        self._next_arr = (C.c_void_p * n)(*[a.ctypes.data for a in self.next_ibf_id])
        self._b2u_arr = (C.c_void_p * n)(*[a.ctypes.data for a in self.bin_to_user])
        self._s = _HibfS(n, C.cast(self._ibf_arr, C.POINTER(_IbfS)), C.cast(self._next_arr, C.POINTER(C.c_void_p)),
                         C.cast(self._b2u_arr, C.POINTER(C.c_void_p)), self.n_user_bins)

    def cstruct(self) -> _HibfS:
        return self._s

    def bulk_count(self, hashes: np.ndarray, threshold: int) -> np.ndarray:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        res = np.zeros(max(self.n_user_bins, 1), dtype=np.uint16)
        lib().gno_hibf_bulk_count(C.byref(self._s), _ptr(hashes), len(hashes), int(threshold), _ptr(res))
        return res[: self.n_user_bins]

    def bulk_count_longreads(self, hashes: np.ndarray, threshold: int) -> np.ndarray:
        """the agent of the reference's -DLONGREADS build: uint32 counts and sums (no wrap at 2^16)"""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        res = np.zeros(max(self.n_user_bins, 1), dtype=np.uint32)
        lib().gno_hibf_bulk_count_longreads(C.byref(self._s), _ptr(hashes), len(hashes), int(threshold), _ptr(res))
        return res[: self.n_user_bins]

    def visited_bytes(self, hashes: np.ndarray, threshold: int) -> int:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        return int(lib().gno_hibf_visited_bytes(C.byref(self._s), _ptr(hashes), len(hashes), int(threshold)))


@dataclass
class Filter:
    """One filter of a hierarchy level: IBF or HIBF + target map (GanonClassify.cpp:279-287)."""
    ibf: Optional[Ibf] = None
    hibf: Optional[Hibf] = None
    targets: List[str] = field(default_factory=list)          # local target names
    target_bins: List[List[int]] = field(default_factory=list)  # bins of each target
    target_fpr: Optional[List[float]] = None
    rel_cutoff: float = 0.2
    # filled by Level
    _keep: list = field(default_factory=list)

    def build(self, global_ids: Sequence[int]) -> _FilterS:
        off = np.zeros(len(self.targets) + 1, dtype=np.uint32)
        for i, b in enumerate(self.target_bins):
            off[i + 1] = off[i] + len(b)
        bins = np.array([x for b in self.target_bins for x in b], dtype=np.uint32)
        if len(bins) == 0:
            bins = np.zeros(1, dtype=np.uint32)
        glob = np.ascontiguousarray(global_ids, dtype=np.uint32)
        fpr = np.ascontiguousarray(self.target_fpr if self.target_fpr is not None else [0.0] * len(self.targets),
                                   dtype=np.float64)
        self._keep = [off, bins, glob, fpr]
        s = _FilterS()
        s.is_hibf = 1 if self.hibf is not None else 0
        s.ibf = C.pointer(self.ibf.cstruct()) if self.ibf is not None else None
        s.hibf = C.pointer(self.hibf.cstruct()) if self.hibf is not None else None
        s.n_targets = len(self.targets)
        s.tgt_bin_off = off.ctypes.data
        s.tgt_bins = bins.ctypes.data
        s.tgt_global = glob.ctypes.data
        s.tgt_fpr = fpr.ctypes.data
        s.rel_cutoff = float(self.rel_cutoff)
        return s


@dataclass
class ReadResult:
    status: int                 # 0 evaluated, 1 small, 2 big
    n_hashes: int
    max_count: int
    min_count: int
    threshold_filter: int
    matches: dict               # target -> count   (TMatches after select_matches)
    kept: dict                  # target -> count   (after filter_matches)
    discarded_filter: list
    discarded_fpr: list


class Level:
    """All filters of one hierarchy level; classify reads like GanonClassify.cpp:676-768."""

    def __init__(self, filters: Sequence[Filter], k: int, w: int, rel_filter: float = 0.0, fpr_query: float = 1.0):
        self.filters = list(filters)
        self.k, self.w = int(k), int(w)
        self.rel_filter, self.fpr_query = float(rel_filter), float(fpr_query)
        self.names: List[str] = []
        idx = {}
        gids = []
        for f in self.filters:
            g = []
            for t in f.targets:
                if t not in idx:
                    idx[t] = len(self.names)
                    self.names.append(t)
                g.append(idx[t])
            gids.append(g)
        self._structs = (_FilterS * len(self.filters))(*[f.build(g) for f, g in zip(self.filters, gids)])
        maxbins = 1
        for f in self.filters:
            maxbins = max(maxbins, f.ibf.bins if f.ibf is not None else f.hibf.n_user_bins)
        self._counts = np.zeros(maxbins, dtype=np.uint16)

    def classify(self, seq1: np.ndarray, seq2: Optional[np.ndarray] = None) -> ReadResult:
        seq1 = np.ascontiguousarray(seq1, dtype=np.uint8)
        seq2 = np.ascontiguousarray(seq2 if seq2 is not None else np.zeros(0, np.uint8), dtype=np.uint8)
        ng = len(self.names)
        mc = np.zeros(max(ng, 1), dtype=np.uint64)
        mf = np.zeros(max(ng, 1), dtype=np.float64)
        keep = np.zeros(max(ng, 1), dtype=np.uint8)
        cap = len(seq1) + len(seq2) + 1
        hs = np.zeros(cap, dtype=np.uint64)
        res = _ReadResS()
        st = lib().gno_classify_read(self._structs, len(self.filters), ng, _ptr(seq1), len(seq1), _ptr(seq2),
                                     len(seq2), self.k, self.w, self.rel_filter, self.fpr_query, _ptr(mc), _ptr(mf),
                                     _ptr(keep), C.byref(res), _ptr(hs), cap, _ptr(self._counts))
        matches = {self.names[g]: int(mc[g]) for g in range(ng) if mc[g] > 0}
        kept = {self.names[g]: int(mc[g]) for g in range(ng) if keep[g] == 1}
        return ReadResult(st, int(res.n_hashes), int(res.max_count_read), int(res.min_count_read),
                          int(res.threshold_filter), matches, kept,
                          [self.names[g] for g in range(ng) if keep[g] == 2],
                          [self.names[g] for g in range(ng) if keep[g] == 3])


class Lca:
    """LCA over string node ids (src/utils/include/utils/LCA.hpp), via the integer-id C oracle."""

    def __init__(self, edges: Sequence[tuple], root: str):
        # edges: (parent, child)
        self.ids = {}
        self.names = []

        def nid(x):
            if x not in self.ids:
                self.ids[x] = len(self.names)
                self.names.append(x)
            return self.ids[x]

        pairs = [(nid(p), nid(c)) for p, c in edges]
        nid(root)
        parent = np.full(len(self.names), -1, dtype=np.int32)
        for p, c in pairs:
            if c != self.ids[root]:
                parent[c] = p
        self._parent = parent
        self._h = lib().gno_lca_build(_ptr(parent), len(parent), self.ids[root])

    def lca(self, nodes: Sequence[str]) -> str:
        arr = np.array([self.ids[n] for n in nodes], dtype=np.int32)
        return self.names[lib().gno_lca_query(self._h, _ptr(arr), len(arr))]

    def __del__(self):
        try:
            lib().gno_lca_free(self._h)
        except Exception:
            pass


def baseline_classify(flt: Filter, ranks: np.ndarray, off: np.ndarray, k: int, w: int, threads: int):
    """OpenMP CPU baseline over a flat IBF (bench.py cpu_baseline, kind 'port')."""
    st = flt.build(list(range(len(flt.targets))))
    ranks = np.ascontiguousarray(ranks, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    nh = np.zeros(n, dtype=np.uint32)
    nm = np.zeros(n, dtype=np.uint32)
    ck = C.c_uint64(0)
    total = lib().gno_baseline_classify(C.byref(st), _ptr(ranks), _ptr(off), n, k, w, threads, _ptr(nh), _ptr(nm),
                                        C.byref(ck))
    return int(total), nh, nm, int(ck.value)
