"""Helpers shared by the GPU parity tests: batch packing and oracle-side expectations."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

import oracle


class _Switches:
    """The library's switch list (gn_ablate, include/ganon_hip.h) as a set the tests add to and take from; conftest.py clears
    it after every test.  on("chunk=333") replaces an earlier "chunk=..."."""

    def __init__(self):
        self.names: List[str] = []

    def _apply(self):
        import ganon_amd
        ganon_amd.set_ablation(self.names)

    def on(self, name: str):
        key = name.split("=")[0]
        self.names = [n for n in self.names if n.split("=")[0] != key] + [name]
        self._apply()

    def off(self, name: str):
        key = name.split("=")[0]
        self.names = [n for n in self.names if n.split("=")[0] != key]
        self._apply()

    def clear(self):
        if self.names:
            self.names = []
            self._apply()


SW = _Switches()


def pack_reads(seqs1: Sequence[bytes], seqs2: Optional[Sequence[bytes]] = None):
    """-> (bases uint8[], off1 u64[n+1], off2 u64[n+1] | None): mate-1 block then mate-2 block."""
    n = len(seqs1)
    off1 = np.zeros(n + 1, dtype=np.uint64)
    off1[1:] = np.cumsum([len(s) for s in seqs1])
    blob = b"".join(seqs1)
    off2 = None
    if seqs2 is not None:
        assert len(seqs2) == n
        off2 = np.zeros(n + 1, dtype=np.uint64)
        off2[1:] = np.cumsum([len(s) for s in seqs2])
        off2 += np.uint64(len(blob))
        blob += b"".join(seqs2)
    bases = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(0, dtype=np.uint8)
    return bases, off1, off2


def oracle_hashes(seqs1, seqs2, k, w):
    """per read: (status, hashes) following GanonClassify.cpp:690-706"""
    out = []
    for i, s1 in enumerate(seqs1):
        if len(s1) < w:
            out.append((1, np.zeros(0, np.uint64)))
            continue
        h = oracle.minimiser_hash(oracle.to_ranks(s1), k, w)
        if seqs2 is not None and len(seqs2[i]) >= w:
            h = np.concatenate([h, oracle.minimiser_hash(oracle.to_ranks(seqs2[i]), k, w)])
        out.append((2 if len(h) > 65535 else 0, h))
    return out


def oracle_matches(ibf: oracle.Ibf, bin2target: np.ndarray, n_targets: int, hashes: np.ndarray, rel_cutoff: float):
    """select_matches over one flat IBF (GanonClassify.cpp:504-541) -> sorted [(target, count)] and dense counts"""
    n = len(hashes)
    counts = ibf.bulk_count(hashes)
    sums = np.zeros(n_targets, dtype=np.uint64)
    valid = bin2target != 0xFFFFFFFF
    np.add.at(sums, bin2target[valid].astype(np.int64), counts[valid].astype(np.uint64))
    sums = np.minimum(sums, n)
    thr = oracle.threshold_cutoff(n, rel_cutoff)
    hit = np.nonzero(sums >= thr)[0]
    return [(int(t), int(sums[t])) for t in hit], counts


def random_seq(rng, length: int, alphabet: bytes = b"ACGT") -> bytes:
    return bytes(np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=length)])
