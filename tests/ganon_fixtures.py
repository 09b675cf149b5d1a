"""Test-only fixture builder: a minimal restatement of ``ganon-build`` plus
writers for the reference's on-disk formats, so the reference's known-answer tests can run
end-to-end without SeqAn3.

Follows /root/reference/src/ganon-build/GanonBuild.cpp:
  bin_size :290-306, hash_functions_from_ratio :308-314, get_optimal_hash_functions :316-333,
  number_of_bins :336-347, correction_rate :350-362, optimal_bins :365-371, false_positive :373-380,
  true_false_positive :382-412, optimal_hashes :428-616, create_bin_map_hash :619-653,
  build/emplace :655-698, save_filter :251-288 (cereal binary layout),
and the raptor 3.0.1 index layout read at src/ganon-classify/GanonClassify.cpp:884-901 (App. A.4).

Differences (documented, do not change classification semantics): targets keep their input order
(the reference iterates a robin_hood map), and the hashes of a split target are distributed to its
bins in sorted order (the reference uses robin_hood set order).
"""
from __future__ import annotations

import gzip
import math
import os
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import oracle


# ----------------------------------------------------------------------------- size optimiser
def bin_size(max_fp: float, n_hashes: int, hash_functions: Optional[int] = None) -> int:
    if hash_functions is None:  # GanonBuild.cpp:290-296
        return int(math.ceil((n_hashes * math.log(max_fp)) / math.log(1.0 / math.pow(2, math.log(2)))))
    # :298-306
    return int(math.ceil(n_hashes * (-hash_functions / math.log(1 - math.exp(math.log(max_fp) / hash_functions)))))


def hash_functions_from_ratio(bin_size_bits: int, n_hashes: int) -> int:
    return int(math.log(2) * (bin_size_bits / float(n_hashes))) & 0xFF  # :308-314 (uint8_t cast)


def get_optimal_hash_functions(bin_size_bits: int, n_hashes: int, hash_functions: int, max_hash_functions: int) -> int:
    opt = hash_functions  # :316-333
    if opt == 0:
        opt = hash_functions_from_ratio(bin_size_bits, n_hashes)
    if opt > max_hash_functions or opt == 0:
        opt = max_hash_functions
    return opt


def number_of_bins(hashes_count: Dict[str, int], n_hashes: int) -> int:
    return int(sum(math.ceil(c / float(n_hashes)) for c in hashes_count.values()))  # :336-347


def false_positive(bin_size_bits: int, hash_functions: int, n_hashes: int) -> float:
    return math.pow(1 - math.exp(-hash_functions / (bin_size_bits / float(n_hashes))), hash_functions)  # :373-380


def correction_rate(max_split_bins: int, max_fp: float, hash_functions: int, n_hashes: int) -> float:
    target_fpr = 1.0 - math.exp(math.log(1.0 - max_fp) / max_split_bins)  # :350-362
    new_bin_size = bin_size(target_fpr, n_hashes, hash_functions)
    original_bin_size = bin_size(max_fp, n_hashes, hash_functions)
    return float(new_bin_size) / original_bin_size


def optimal_bins(n_bins: int) -> int:
    return int(math.ceil(n_bins / 64.0) * 64)  # :365-371


def true_false_positive(hashes_count, max_hashes_bin, bin_size_bits, hash_functions):
    highest, avg = 0.0, 0.0  # :382-412
    for count in hashes_count.values():
        n_bins_target = int(math.ceil(count / float(max_hashes_bin)))
        n_hashes_bin = int(math.ceil(count / float(n_bins_target)))
        real_fp = 1.0 - math.pow(1.0 - false_positive(bin_size_bits, hash_functions, n_hashes_bin), n_bins_target)
        highest = max(highest, real_fp)
        avg += real_fp
    return highest, avg / float(len(hashes_count))


def optimal_hashes(max_fp: float, filter_size: float, hashes_count: Dict[str, int], hash_functions: int = 0,
                   max_hash_functions: int = 5, mode: str = "avg") -> dict:
    """GanonBuild.cpp:428-616 -> dict(n_bins, max_hashes_bin, hash_functions, bin_size_bits, max_fp)."""
    max_hashes = max(hashes_count.values())
    min_filter_size, min_bins, min_fp = 0, 0, 1.0
    sims = []
    it = 100
    if max_hashes < it:
        it = max_hashes
    n = max_hashes + 1
    while n > it:
        n_hashes = n - 1
        n_bins = number_of_bins(hashes_count, n_hashes)
        bsb = 0
        if filter_size:
            bsb = int((filter_size / float(optimal_bins(n_bins))) * 8388608)
            ohf = get_optimal_hash_functions(bsb, n_hashes, hash_functions, max_hash_functions)
        elif hash_functions == 0:
            bsb = bin_size(max_fp, n_hashes)
            ohf = get_optimal_hash_functions(bsb, n_hashes, hash_functions, max_hash_functions)
        else:
            ohf = get_optimal_hash_functions(bsb, n_hashes, hash_functions, max_hash_functions)
            bsb = bin_size(max_fp, n_hashes, ohf)
        max_split_bins = int(math.ceil(max_hashes / float(n_hashes)))
        fp, fsb = 0.0, 0
        if filter_size:
            fp = 1 - math.pow(1.0 - false_positive(bsb, ohf, n_hashes), max_split_bins)
            min_fp = min(min_fp, fp)
        else:
            avg_n_hashes = int(math.ceil(max_hashes / float(max_split_bins)))
            approx_fp = false_positive(bsb, ohf, avg_n_hashes)
            if approx_fp > max_fp:
                approx_fp = max_fp
            crate = correction_rate(max_split_bins, approx_fp, ohf, n_hashes)
            bsb = int(bsb * crate)
            fsb = bsb * optimal_bins(n_bins)
            if fsb == 0 or math.isinf(crate):
                break
            if fsb < min_filter_size or min_filter_size == 0:
                min_filter_size = fsb
        sims.append((n_hashes, n_bins, fsb, fp))
        if n_bins < min_bins or min_bins == 0:
            min_bins = n_bins
        n -= it

    mode_val = 1.0
    if mode in ("smaller", "faster"):
        mode_val = 0.5
    elif mode in ("smallest", "fastest"):
        mode_val = 0.0
    var_val = bins_val = 1.0
    if mode in ("smaller", "smallest"):
        var_val = mode_val
    elif mode in ("faster", "fastest"):
        bins_val = mode_val
    cfg = dict(n_bins=0, max_hashes_bin=0, hash_functions=0, bin_size_bits=0, max_fp=0.0)
    min_avg = 0.0
    for n_hashes, n_bins, fsb, fp in sims:
        var_ratio = fp / min_fp if filter_size else fsb / float(min_filter_size)
        bins_ratio = n_bins / float(min_bins)
        avg = (1 + mode_val ** 2) * ((var_ratio * bins_ratio) / ((var_val * var_ratio) + (bins_val * bins_ratio)))
        if avg < min_avg or min_avg == 0:
            min_avg = avg
            if filter_size:
                cfg["bin_size_bits"] = int((filter_size / float(optimal_bins(n_bins))) * 8388608)
                cfg["max_fp"] = fp
            else:
                cfg["bin_size_bits"] = fsb // optimal_bins(n_bins)
                cfg["max_fp"] = max_fp
            cfg["max_hashes_bin"] = n_hashes
            cfg["n_bins"] = n_bins
            cfg["hash_functions"] = get_optimal_hash_functions(cfg["bin_size_bits"], n_hashes, hash_functions,
                                                               max_hash_functions)
    return cfg


# ----------------------------------------------------------------------------- mini ganon-build
class BuiltIbf:
    def __init__(self):
        self.ibf: oracle.Ibf = None
        self.config: dict = {}
        self.hashes_count: List[Tuple[str, int]] = []
        self.bin_map: List[Tuple[int, str]] = []
        self.target_hashes: Dict[str, np.ndarray] = {}

    def as_filter(self, rel_cutoff: float = 0.2) -> oracle.Filter:
        """Filter with per-target fpr as computed by the classifier's loader (GanonClassify.cpp:968-982)."""
        targets, bins = [], {}
        for b, t in self.bin_map:
            if t not in bins:
                bins[t] = []
                targets.append(t)
            bins[t].append(b)
        cnt = dict(self.hashes_count)
        fpr = [float(oracle.lib().gno_target_fpr(cnt[t], self.config["max_hashes_bin"], self.config["bin_size_bits"],
                                                 self.config["hash_functions"])) for t in targets]
        return oracle.Filter(ibf=self.ibf, targets=targets, target_bins=[bins[t] for t in targets], target_fpr=fpr,
                             rel_cutoff=rel_cutoff)


def literal_to_ranks(seq: str) -> np.ndarray:
    """The reference's test literals ('-' etc. become rank 0, GanonClassify.test.cpp:813)."""
    return oracle.to_ranks(seq)


def build_ibf(targets: Dict[str, Sequence[str]], k: int, w: int, max_fp: float = 0.05, filter_size: float = 0.0,
              hash_functions: int = 0, mode: str = "avg", min_length: int = 0) -> BuiltIbf:
    """GanonBuild::run (GanonBuild.cpp:752-921) over literal sequences. targets: name -> seq or [seqs]."""
    out = BuiltIbf()
    hashes_count: Dict[str, int] = {}
    for name, seqs in targets.items():
        if isinstance(seqs, str):
            seqs = [seqs]
        hs = set()
        for s in seqs:  # count_hashes :184-249 (distinct minimisers per file/target)
            r = literal_to_ranks(s)
            if len(r) < min_length:
                continue
            hs.update(oracle.minimiser_hash(r, k, w).tolist())
        out.target_hashes[name] = np.array(sorted(hs), dtype=np.uint64)
        hashes_count[name] = len(hs)
    cfg = optimal_hashes(max_fp, filter_size, hashes_count, hash_functions, 5, mode)
    cfg["kmer_size"], cfg["window_size"] = k, w
    cfg["true_max_fp"], cfg["true_avg_fp"] = true_false_positive(hashes_count, cfg["max_hashes_bin"],
                                                                 cfg["bin_size_bits"], cfg["hash_functions"])
    # create_bin_map_hash :619-653
    binno = 0
    bin_map_hash = []
    for target, count in hashes_count.items():
        n_bins_target = int(math.ceil(count / float(cfg["max_hashes_bin"])))
        n_hashes_bin = int(math.ceil(count / float(n_bins_target)))
        if n_hashes_bin > cfg["max_hashes_bin"]:
            n_hashes_bin = cfg["max_hashes_bin"]
        for i in range(n_bins_target):
            st = i * n_hashes_bin
            en = st + n_hashes_bin - 1
            if st >= count:
                break
            if en >= count:
                en = count - 1
            bin_map_hash.append((binno, target, st, en))
            binno += 1
    assert len(bin_map_hash) == cfg["n_bins"]
    ibf = oracle.Ibf(cfg["n_bins"], cfg["bin_size_bits"], cfg["hash_functions"])  # :873-875
    for b, target, st, en in bin_map_hash:  # build :655-698
        ibf.emplace_many(out.target_hashes[target][st:en + 1], b)
    out.ibf = ibf
    out.config = cfg
    out.hashes_count = list(hashes_count.items())
    out.bin_map = [(b, t) for b, t, _, _ in bin_map_hash]
    return out


# ----------------------------------------------------------------------------- cereal binary writers
def _w_str(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def ibf_cereal_bytes(ibf: oracle.Ibf, bv_header: str = "wgb") -> bytes:
    """seqan3::interleaved_bloom_filter<uncompressed> cereal layout:
    6 x u64 members, then sdsl bit_vector = u8 width(1), f32 growth_factor(1.5), u64 size_in_bits, payload.
    `bv_header` spells the variants of the (unpinned) bit_vector header the loader accepts: w = width byte, g = growth
    factor, b / q = size in bits / in 64-bit words."""
    head = struct.pack("<6Q", ibf.bins, ibf.technical_bins, ibf.bin_size, ibf.hash_shift, ibf.bin_words,
                       ibf.hash_funs)
    bits = ibf.technical_bins * ibf.bin_size
    bv = b""
    if "w" in bv_header:
        bv += struct.pack("<B", 1)
    if "g" in bv_header:
        bv += struct.pack("<f", 1.5)
    bv += struct.pack("<Q", bits // 64 if "q" in bv_header else bits)
    return head + bv + np.ascontiguousarray(ibf.data, dtype="<u8").tobytes()


def write_ibf(path: str, built: BuiltIbf, version=(2, 1, 1), bv_header: str = "wgb") -> None:
    """save_filter (GanonBuild.cpp:251-288; reader GanonClassify.cpp:955-965)."""
    c = built.config
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", *version))
        f.write(struct.pack("<QQBBHQddd", c["n_bins"], c["max_hashes_bin"], c["hash_functions"], c["kmer_size"],
                            c["window_size"], c["bin_size_bits"], c["max_fp"], c["true_max_fp"], c["true_avg_fp"]))
        f.write(struct.pack("<Q", len(built.hashes_count)))
        for t, n in built.hashes_count:
            f.write(_w_str(t) + struct.pack("<Q", n))
        f.write(struct.pack("<Q", len(built.bin_map)))
        for b, t in built.bin_map:
            f.write(struct.pack("<Q", b) + _w_str(t))
        f.write(ibf_cereal_bytes(built.ibf, bv_header))


def write_hibf(path: str, hibf: oracle.Hibf, bin_path: Sequence[Sequence[str]], k: int, w: int, fpr: float,
               user_bin_filenames: Optional[Sequence[str]] = None, version: int = 1, bv_header: str = "wgb") -> None:
    """raptor 3.0.1 index as read at GanonClassify.cpp:884-901 + hibf.hpp:163-169,293-298."""
    if user_bin_filenames is None:
        user_bin_filenames = [p[0] for p in bin_path]
    shape_bits = (1 << k) - 1
    with open(path, "wb") as f:
        f.write(struct.pack("<I", version))
        f.write(struct.pack("<Q", w))
        f.write(struct.pack("<QQ", k, shape_bits))  # seqan3::shape: u64 size, u64 bits
        f.write(struct.pack("<B", 1))               # parts
        f.write(struct.pack("<B", 0))               # compressed
        f.write(struct.pack("<Q", len(bin_path)))
        for lst in bin_path:
            f.write(struct.pack("<Q", len(lst)))
            for s in lst:
                f.write(_w_str(s))
        f.write(struct.pack("<d", fpr))
        f.write(struct.pack("<B", 1))               # is_hibf
        f.write(struct.pack("<Q", len(hibf.ibfs)))  # ibf_vector
        for ibf in hibf.ibfs:
            f.write(ibf_cereal_bytes(ibf, bv_header))
        f.write(struct.pack("<Q", len(hibf.next_ibf_id)))
        for a in hibf.next_ibf_id:
            f.write(struct.pack("<Q", len(a)) + np.ascontiguousarray(a, dtype="<i8").tobytes())
        f.write(struct.pack("<Q", len(user_bin_filenames)))
        for s in user_bin_filenames:
            f.write(_w_str(s))
        f.write(struct.pack("<Q", len(hibf.bin_to_user)))
        for a in hibf.bin_to_user:
            f.write(struct.pack("<Q", len(a)) + np.ascontiguousarray(a, dtype="<i8").tobytes())


# ----------------------------------------------------------------------------- sequence files
def write_fasta(path: str, records: Sequence[Tuple[str, str]]) -> None:
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wt") as f:
        for rid, seq in records:
            f.write(f">{rid}\n{seq}\n")


def write_fastq(path: str, records: Sequence[Tuple[str, str]]) -> None:
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wt") as f:
        for rid, seq in records:
            f.write(f"@{rid}\n{seq}\n+\n{'I' * len(seq)}\n")


def write_tax(path: str, tax: Dict[str, str]) -> None:
    """GanonClassify.test.cpp:170-181 (root '1' + auto rank/name)."""
    with open(path, "w") as f:
        f.write("1\t0\troot\troot\n")
        for t, p in tax.items():
            f.write(f"{t}\t{p}\trank-{t}\tname-{t}\n")


# ----------------------------------------------------------------------------- synthetic data
def random_reads(n: int, length: int, seed: int) -> np.ndarray:
    """n x length uint8 ranks, uniform iid."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 4, size=(n, length), dtype=np.uint8)


def ranks_to_ascii(r: np.ndarray) -> np.ndarray:
    return np.frombuffer(b"ACGT", dtype=np.uint8)[r]


def random_ibf(bins: int, bin_size_rows: int, hash_funs: int, density: float, seed: int) -> oracle.Ibf:
    """IBF whose bit matrix is iid Bernoulli(density) (SURVEY 8d 'Synthetic inputs' (b)); padding bins
    (>= bins) are cleared like a real filter."""
    rng = np.random.default_rng(seed)
    W = (bins + 63) >> 6
    if abs(density - 0.5) < 1e-12:
        data = rng.integers(0, 1 << 64, size=(bin_size_rows, W), dtype=np.uint64)
    else:
        bits = rng.random((bin_size_rows, W * 64)) < density
        data = np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(bin_size_rows, W)
    if bins & 63:
        data[:, W - 1] &= np.uint64((1 << (bins & 63)) - 1)
    return oracle.Ibf(bins, bin_size_rows, hash_funs, data)


# ----------------------------------------------------------------------------- synthetic HIBF
def random_hibf(n_user_bins: int, tmax: int, max_depth: int, seed: int, density: float = 0.3, hash_funs: int = 3,
                rows=(300, 900), user_hashes: Optional[Dict[int, np.ndarray]] = None) -> oracle.Hibf:
    """Random raptor-style layout: every IBF has <= tmax technical bins holding single user bins, user bins split
    over 2-3 consecutive technical bins, and merged bins that point to a child IBF (hibf.hpp:124-136,188).  Bit
    matrices are Bernoulli(density); `user_hashes[ub]` (optional) are emplaced into the user bin's leaf bin(s) and
    into every merged bin above it, the way raptor builds the hierarchy."""
    rng = np.random.default_rng(seed)
    ibfs: List[oracle.Ibf] = []
    next_ids: List[List[int]] = []
    b2u: List[List[int]] = []
    plant: List[List[tuple]] = []  # per ibf: (bin, user bins whose hashes go there)

    def build(ubs: List[int], depth: int) -> int:
        idx = len(ibfs)
        ibfs.append(None)
        next_ids.append([])
        b2u.append([])
        plant.append([])
        bins_nxt, bins_usr, pl = [], [], []
        pending = list(ubs)
        slots_left = tmax
        pending_children = []
        while pending:
            remaining_slots = slots_left - len(bins_usr)
            if depth + 1 < max_depth and len(pending) > remaining_slots - 3 and len(pending) > 1:
                # merged bin: put a chunk of user bins below
                take = max(2, int(np.ceil(len(pending) / max(1, remaining_slots - 2))))
                take = min(take + int(rng.integers(0, 3)), len(pending))
                group, pending = pending[:take], pending[take:]
                pl.append((len(bins_usr), group))
                bins_usr.append(-1)
                bins_nxt.append(None)
                pending_children.append((len(bins_usr) - 1, group))
            else:
                ub = pending.pop(0)
                nsplit = int(rng.choice([1, 1, 1, 2, 3])) if remaining_slots - len(pending) > 3 else 1
                for _ in range(nsplit):
                    pl.append((len(bins_usr), [ub]))
                    bins_usr.append(ub)
                    bins_nxt.append(idx)
        for pos, group in pending_children:
            bins_nxt[pos] = build(group, depth + 1)
        nb = len(bins_usr)
        S = int(rng.integers(rows[0], rows[1]))
        ibfs[idx] = random_ibf(nb, S, hash_funs, density, seed=int(rng.integers(0, 1 << 30)))
        next_ids[idx] = bins_nxt
        b2u[idx] = bins_usr
        plant[idx] = pl
        return idx

    build(list(range(n_user_bins)), 0)
    if user_hashes:
        for i, pl in enumerate(plant):
            seen_split = {}
            for b, group in pl:
                for ub in group:
                    hv = user_hashes.get(ub)
                    if hv is None or len(hv) == 0:
                        continue
                    if len(group) == 1 and b2u[i][b] == ub:
                        # split bins share the user bin's hashes round-robin
                        bins_of = [bb for bb, g in pl if g == [ub]]
                        part = hv[bins_of.index(b)::len(bins_of)]
                        ibfs[i].emplace_many(part, b)
                    else:
                        ibfs[i].emplace_many(hv, b)
    return oracle.Hibf(ibfs, next_ids, b2u, n_user_bins)


def hibf_upload_args(h: oracle.Hibf):
    """arguments of ganon_amd.HipFilter.hibf for an oracle.Hibf"""
    return ([(f.data.reshape(-1), f.bins, f.bin_size, f.hash_funs) for f in h.ibfs], h.next_ibf_id, h.bin_to_user,
            h.n_user_bins)
