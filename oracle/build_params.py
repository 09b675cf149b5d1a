"""CPU restatement of ganon-build's filter sizing and bin layout (the part of the build that is plain arithmetic).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ to check the product binary
``ganon_amd/host/ganon-build`` -- never by the product.

Follows /root/reference/src/ganon-build/GanonBuild.cpp, one function per reference function, every double
operation in the reference's order (CPython's ``math`` calls the same libm as std::log/std::pow/std::exp, so values
agree to the last bit on one machine):

  bin_size (2 args)            :290-296      bin_size (3 args)          :298-306
  hash_functions_from_ratio    :308-314      get_optimal_hash_functions :316-333
  number_of_bins               :336-347      correction_rate            :350-362
  optimal_bins                 :365-371      false_positive             :373-380
  true_false_positive          :382-412      optimal_hashes             :427-616
  create_bin_map_hash          :619-653

PARITY UNPINNED: the reference's own tests for this code (tests/ganon-build/GanonBuild.test.cpp) hold no expected
numbers, only properties (true fp <= requested fp, file-size and bin-count inequalities between modes, every inserted
minimiser found again); tests/test_build_oracle.py checks those properties on this restatement, on the reference's own
25-genome data set (tests/golden/build_mode/).  The ORDER of targets is the iteration order of a robin_hood map in the
reference (not reproducible here); this restatement and the product both take targets in the order
given.

Integer/double conversions mirror the C++ types: uint64_t <- std::ceil(double) truncates; ``int64_t bin_size_bits =
bin_size_bits * crate`` multiplies in double and truncates; uint8_t <- double keeps the low 8 bits of the truncated
value (what x86-64 compilers emit; out-of-range is undefined in the standard).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

MAX_HASH_FUNCTIONS = 5  # Config.hpp:27


def _u64(x: float) -> int:
    return int(x)  # non-negative finite doubles only reach this


def bin_size2(max_fp: float, n_hashes: int) -> int:
    return _u64(math.ceil((n_hashes * math.log(max_fp)) / math.log(1.0 / math.pow(2, math.log(2)))))


def bin_size3(max_fp: float, n_hashes: int, hash_functions: int) -> int:
    return _u64(math.ceil(n_hashes * (-hash_functions / math.log(1 - math.exp(math.log(max_fp) / hash_functions)))))


def hash_functions_from_ratio(bin_size_bits: int, n_hashes: int) -> int:
    return int(math.log(2) * (bin_size_bits / float(n_hashes))) & 0xFF


def get_optimal_hash_functions(bin_size_bits: int, n_hashes: int, hash_functions: int, max_hash_functions: int = MAX_HASH_FUNCTIONS) -> int:
    o = hash_functions
    if o == 0:
        o = hash_functions_from_ratio(bin_size_bits, n_hashes)
    if o > max_hash_functions or o == 0:
        o = max_hash_functions
    return o


def number_of_bins(counts: Sequence[int], n_hashes: int) -> int:
    """`n_bins += std::ceil(...)` on a uint64_t converts n_bins to double, adds, converts back; exact below 2^53"""
    n = 0
    for c in counts:
        n = _u64(float(n) + math.ceil(c / float(n_hashes)))
    return n


def correction_rate(max_split_bins: int, max_fp: float, hash_functions: int, n_hashes: int) -> float:
    target_fpr = 1.0 - math.exp(math.log(1.0 - max_fp) / max_split_bins)
    new_bin_size = bin_size3(target_fpr, n_hashes, hash_functions)
    original_bin_size = bin_size3(max_fp, n_hashes, hash_functions)
    if original_bin_size == 0:
        return math.inf if new_bin_size else math.nan
    return float(new_bin_size) / original_bin_size


def optimal_bins(n_bins: int) -> int:
    return _u64(math.ceil(n_bins / 64.0) * 64)


def false_positive(bin_size_bits: int, hash_functions: int, n_hashes: int) -> float:
    return math.pow(1 - math.exp(-hash_functions / (bin_size_bits / float(n_hashes))), hash_functions)


def true_false_positive(counts: Sequence[int], max_hashes_bin: int, bin_size_bits: int, hash_functions: int) -> Tuple[float, float]:
    highest, average = 0.0, 0.0
    for c in counts:
        if c == 0:
            # a target without minimisers: the reference divides 0/0 here and (on x86-64, where NaN converts to 2^63) ends
            # up with 1 - pow(0, 0) = 0; it still counts in the average's denominator
            continue
        n_bins_target = _u64(math.ceil(c / float(max_hashes_bin)))
        n_hashes_bin = _u64(math.ceil(c / float(n_bins_target)))
        real_fp = 1.0 - math.pow(1.0 - false_positive(bin_size_bits, hash_functions, n_hashes_bin), n_bins_target)
        if real_fp > highest:
            highest = real_fp
        average += real_fp
    return highest, average / float(len(counts))


@dataclass
class IbfConfig:  # src/utils/include/utils/IBFConfig.hpp:10-24
    n_bins: int = 0
    max_hashes_bin: int = 0
    hash_functions: int = 0
    kmer_size: int = 0
    window_size: int = 0
    bin_size_bits: int = 0
    max_fp: float = 0.0
    true_max_fp: float = 0.0
    true_avg_fp: float = 0.0


def optimal_hashes(max_fp: float, filter_size: float, counts: Sequence[int], hash_functions: int, mode: str,
                   max_hash_functions: int = MAX_HASH_FUNCTIONS) -> IbfConfig:
    """counts = distinct minimisers per target (zero-count targets included, as in the reference's hashes_count)"""
    cfg = IbfConfig()
    max_hashes = max(counts) if counts else 0
    if max_hashes == 0:
        return cfg  # the reference's loop would not terminate sensibly; n_bins stays 0 -> "No valid sequences to build"
    min_filter_size, min_bins, min_fp = 0, 0, 1.0
    sims: List[Tuple[int, int, int, float]] = []
    it = 100
    if max_hashes < it:
        it = max_hashes
    n = max_hashes + 1
    while n > it:
        n_hashes = n - 1
        n_bins = number_of_bins(counts, n_hashes)
        bin_size_bits, ohf = 0, 0
        if filter_size:
            bin_size_bits = int((filter_size / float(optimal_bins(n_bins))) * 8388608)
            ohf = get_optimal_hash_functions(bin_size_bits, n_hashes, hash_functions, max_hash_functions)
        elif hash_functions == 0:
            bin_size_bits = bin_size2(max_fp, n_hashes)
            ohf = get_optimal_hash_functions(bin_size_bits, n_hashes, hash_functions, max_hash_functions)
        else:
            ohf = get_optimal_hash_functions(bin_size_bits, n_hashes, hash_functions, max_hash_functions)
            bin_size_bits = bin_size3(max_fp, n_hashes, ohf)
        max_split_bins = _u64(math.ceil(max_hashes / float(n_hashes)))
        fp, filter_size_bits = 0.0, 0
        if filter_size:
            fp = 1 - math.pow(1.0 - false_positive(bin_size_bits, ohf, n_hashes), max_split_bins)
            if fp < min_fp:
                min_fp = fp
        else:
            avg_n_hashes = _u64(math.ceil(max_hashes / float(max_split_bins)))
            approx_fp = false_positive(bin_size_bits, ohf, avg_n_hashes)
            if approx_fp > max_fp:
                approx_fp = max_fp
            crate = correction_rate(max_split_bins, approx_fp, ohf, n_hashes)
            if math.isinf(crate) or math.isnan(crate):
                break
            bin_size_bits = int(bin_size_bits * crate)
            filter_size_bits = bin_size_bits * optimal_bins(n_bins)
            if filter_size_bits == 0:
                break
            if filter_size_bits < min_filter_size or min_filter_size == 0:
                min_filter_size = filter_size_bits
        sims.append((n_hashes, n_bins, filter_size_bits, fp))
        if n_bins < min_bins or min_bins == 0:
            min_bins = n_bins
        n -= it

    mode_val = 1.0
    if mode in ("smaller", "faster"):
        mode_val = 0.5
    elif mode in ("smallest", "fastest"):
        mode_val = 0.0
    var_val, bins_val = 1.0, 1.0
    if mode in ("smaller", "smallest"):
        var_val = mode_val
    elif mode in ("faster", "fastest"):
        bins_val = mode_val
    min_avg = 0.0
    for n_hashes, n_bins, filter_size_bits, fp in sims:
        if filter_size:
            var_ratio = fp / min_fp if min_fp else math.nan
        else:
            var_ratio = filter_size_bits / float(min_filter_size)
        bins_ratio = n_bins / float(min_bins)
        den = (var_val * var_ratio) + (bins_val * bins_ratio)
        avg = (1 + math.pow(mode_val, 2)) * ((var_ratio * bins_ratio) / den) if den else math.nan
        if avg < min_avg or min_avg == 0:
            min_avg = avg
            if filter_size:
                cfg.bin_size_bits = int((filter_size / float(optimal_bins(n_bins))) * 8388608)
                cfg.max_fp = fp
            else:
                cfg.bin_size_bits = filter_size_bits // optimal_bins(n_bins)
                cfg.max_fp = max_fp
            cfg.max_hashes_bin = n_hashes
            cfg.n_bins = n_bins
            cfg.hash_functions = get_optimal_hash_functions(cfg.bin_size_bits, n_hashes, hash_functions, max_hash_functions)
    return cfg


def create_bin_map(max_hashes_bin: int, counts: Sequence[int]) -> List[Tuple[int, int, int]]:
    """-> [(target index, first hash index, last hash index)] per technical bin, bins numbered in target order"""
    out = []
    for t, c in enumerate(counts):
        if c == 0:
            continue  # ceil(0/x) = 0 bins (the reference divides 0/0 for n_hashes_bin and then makes no bin)
        n_bins_target = _u64(math.ceil(c / float(max_hashes_bin)))
        n_hashes_bin = _u64(math.ceil(c / float(n_bins_target)))
        if n_hashes_bin > max_hashes_bin:
            n_hashes_bin = max_hashes_bin
        for i in range(n_bins_target):
            st = i * n_hashes_bin
            en = st + n_hashes_bin - 1
            if st >= c:
                break
            if en >= c:
                en = c - 1
            out.append((t, st, en))
    return out
