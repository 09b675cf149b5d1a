"""ctypes mirror of include/ganon_hip.h (the C ABI of libganon_hip.so).  No torch types, no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GANON_HIP_LIB") or os.path.join(_HERE, "csrc", "libganon_hip.so")  # override: A/B builds

READ_OK, READ_SMALL, READ_BIG = 0, 1, 2
FILL_3_OF_8 = 0x38  # gn_filter_fill_random: density 3/8 (GN_FILL_3_OF_8)
FILL_3_OF_16 = 0x3F  # density 3/16: m0 & m1 & (m2 | m3) (GN_FILL_3_OF_16)
MATCH_DTYPE = np.dtype([("read", "<u4"), ("target", "<u4"), ("count", "<u4")])

# every symbol include/ganon_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = ["gn_device_count", "gn_last_error", "gn_filter_upload_ibf", "gn_filter_upload_hibf", "gn_filter_emplace",
               "gn_filter_emplace_ibf", "gn_filter_download_rows", "gn_filter_download_row_list", "gn_pinned_alloc",
               "gn_pinned_free", "gn_filter_write_rows", "gn_filter_write_sync", "gn_filter_finalize",
               "gn_filter_fill_random", "gn_filter_info", "gn_filter_free", "gn_stream_create", "gn_stream_destroy",
               "gn_stream_upload_reads", "gn_stream_minimisers", "gn_stream_classify", "gn_submit_batch", "gn_stream_sync", "gn_fetch_batch", "gn_stream_set_postfilter",
               "gn_fetch_postfilter", "gn_streams_postfilter_joint", "gn_stream_set_long_reads", "gn_stream_device_matches",
               "gn_stream_distinct_hashes", "gn_filter_emplace_split", "gn_stream_fetch_hashes", "gn_stream_dense_counts",
               "gn_stream_timings", "gn_gather_create", "gn_gather_run", "gn_gather_fetch", "gn_gather_device_matches",
               "gn_gather_destroy", "gn_device_memory", "gn_hibf_row_stride_words", "gn_gather_run_buffers", "gn_stream_device_offsets", "gn_stream_hibf_levels", "gn_stream_hibf_level_lines", "gn_stream_classify_shared",
               "gn_stream_upload_fastq", "gn_stream_fastq_index", "gn_stream_fastq_keep", "gn_stream_fastq_records",
               "gn_peer_stats", "gn_stream_upload_text", "gn_stream_upload_text_pair", "gn_stream_text_pair_index", "gn_stream_text_pair_records2",
               "gn_ablate", "gn_filter_probe", "gn_reassign_create", "gn_reassign_run", "gn_reassign_diffs", "gn_reassign_fetch", "gn_reassign_info", "gn_reassign_free",
               "gn_inflate_create", "gn_inflate_destroy", "gn_inflate_feed", "gn_inflate_step", "gn_inflate_text", "gn_inflate_text_device",
               "gn_inflate_get_stats", "gn_inflate_cuts", "gn_inflate_set_carry", "gn_stream_upload_text_device", "gn_stream_fastq_headers",
               "gn_inflate_cuts_lines", "gn_inflate_cut_at_lines", "gn_stream_upload_text_pair_device", "gn_stream_fetch_letters",
               "gn_ibf_hash_constants", "gn_inflate_set_turns", "gn_inflate_handoff",
               "gn_stream_upload_text_pair_devices"]


class PostFilter(C.Structure):  # gn_postfilter
    _fields_ = [("rel_filter", C.c_double), ("fpr_query", C.c_double), ("target_fpr", C.c_void_p), ("joint", C.c_int),
                ("target_gid", C.c_void_p)]


class GanonHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libganon_hip error {code}: {msg}")
        self.code = code


class IbfDesc(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("bin_size", C.c_uint64), ("bin_words", C.c_uint64), ("bins", C.c_uint64),
                ("hash_funs", C.c_uint32), ("hash_shift", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("ms_minimiser", C.c_float), ("ms_count", C.c_float), ("ms_total", C.c_float),
                ("n_hashes", C.c_uint64), ("algo_bytes", C.c_uint64), ("n_matches", C.c_uint64),
                ("n_count_launches", C.c_uint32), ("fetched_bytes", C.c_uint64), ("ms_compact", C.c_float)]


class InflateStats(C.Structure):  # gn_inflate_stats
    _fields_ = [("steps", C.c_uint64), ("chunks", C.c_uint64), ("fixups", C.c_uint64), ("markers", C.c_uint64), ("members", C.c_uint64),
                ("text_bytes", C.c_uint64), ("ms_decode", C.c_double), ("ms_chain", C.c_double), ("ms_resolve", C.c_double),
                ("ms_step_wall", C.c_double), ("prof_ms", C.c_double * 8)]


_lib = None


_ablated: Tuple[str, ...] = tuple(x.strip() for x in os.environ.get("GANON_HIP_ABLATE", "").split(",") if x.strip())


def set_ablation(names) -> None:
    """gn_ablate(): replace the library's switch list (include/ganon_hip.h); "" = the product path"""
    global _ablated
    if isinstance(names, str):
        names = [x.strip() for x in names.split(",") if x.strip()]
    names = tuple(names)
    L = load_library()
    rc = L.gn_ablate(",".join(names).encode())
    if rc:
        raise GanonHipError(rc, L.gn_last_error().decode())
    _ablated = names


def ablation() -> Tuple[str, ...]:
    return _ablated


class ablate:
    """context manager: the named switches on top of the current list, the previous list back on exit
    (`with ganon_amd.ablate("early_exit"): ...`; an empty name list changes nothing)"""

    def __init__(self, *names: str):
        self.names = [x.strip() for n in names for x in n.split(",") if x.strip()]

    def __enter__(self):
        self.prev = _ablated
        key = lambda s: s.split("=")[0]
        set_ablation([p for p in self.prev if key(p) not in {key(n) for n in self.names}] + self.names)
        return self

    def __exit__(self, *exc):
        set_ablation(self.prev)
        return False


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the in-tree libganon_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise GanonHipError(-19, f"{_LIB_PATH} is missing: run `python -m ganon_amd.build` (hipcc, gfx950). "
                                 "There is no CPU fallback.")
    L = C.CDLL(_LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.gn_last_error.restype = C.c_char_p
    L.gn_device_count.argtypes = [C.POINTER(i32)]
    L.gn_filter_upload_ibf.argtypes = [i32, C.POINTER(IbfDesc), vp, u32, C.POINTER(vp)]
    L.gn_filter_upload_hibf.argtypes = [i32, u32, C.POINTER(IbfDesc), C.POINTER(vp), C.POINTER(vp), u64, C.POINTER(vp)]
    L.gn_filter_emplace.argtypes = [vp, vp, vp, u64]
    L.gn_filter_emplace_ibf.argtypes = [vp, u32, vp, vp, u64]
    L.gn_filter_download_rows.argtypes = [vp, u32, u64, u64, vp]
    L.gn_filter_download_row_list.argtypes = [vp, u32, vp, u64, vp]
    L.gn_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.gn_pinned_free.argtypes = [vp]
    L.gn_filter_write_rows.argtypes = [vp, u32, u64, u64, vp, u64, u64]
    L.gn_filter_write_sync.argtypes = [vp]
    L.gn_filter_finalize.argtypes = [vp]
    L.gn_filter_fill_random.argtypes = [vp, u32, u64, u32, u64, u64]
    L.gn_filter_info.argtypes = [vp, C.POINTER(i32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    L.gn_filter_free.argtypes = [vp]
    L.gn_stream_create.argtypes = [vp, u32, u64, u64, C.POINTER(vp)]
    L.gn_stream_destroy.argtypes = [vp]
    L.gn_stream_upload_reads.argtypes = [vp, vp, u64, vp, vp, u32]
    L.gn_stream_minimisers.argtypes = [vp, u32, u32]
    L.gn_stream_classify.argtypes = [vp, u32, u32, C.c_double]
    L.gn_submit_batch.argtypes = [vp, vp, u64, vp, vp, u32, u32, u32, C.c_double]
    L.gn_stream_sync.argtypes = [vp]
    L.gn_stream_classify_shared.argtypes = [vp, vp, C.c_double]
    L.gn_stream_upload_fastq.argtypes = [vp, vp, u64]
    L.gn_stream_upload_text.argtypes = [vp, vp, u64, i32]
    L.gn_stream_upload_text_pair.argtypes = [vp, vp, u64, vp, u64, i32]
    L.gn_stream_text_pair_index.argtypes = [vp, C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    L.gn_stream_text_pair_records2.argtypes = [vp, vp, vp, vp]
    L.gn_stream_fastq_index.argtypes = [vp, C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    L.gn_stream_fastq_keep.argtypes = [vp, u32]
    L.gn_stream_fastq_records.argtypes = [vp, vp, vp, vp]
    L.gn_fetch_batch.argtypes = [vp, vp, vp, vp, vp, u64, C.POINTER(u64)]
    L.gn_stream_device_matches.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.gn_stream_set_postfilter.argtypes = [vp, vp]
    L.gn_fetch_postfilter.argtypes = [vp, vp, C.POINTER(u64), C.POINTER(u64)]
    L.gn_stream_set_long_reads.argtypes = [vp, i32]
    L.gn_streams_postfilter_joint.argtypes = [vp, u32]
    L.gn_stream_fetch_hashes.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.gn_stream_distinct_hashes.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.gn_filter_emplace_split.argtypes = [vp, vp, u64, u32, u64]
    L.gn_stream_dense_counts.argtypes = [vp, u32, u32, vp]
    L.gn_stream_timings.argtypes = [vp, C.POINTER(Timings)]
    L.gn_gather_create.argtypes = [i32, u32, C.POINTER(vp), vp, C.POINTER(vp)]
    L.gn_gather_run.argtypes = [vp, C.POINTER(vp), u32]
    L.gn_gather_fetch.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.gn_gather_device_matches.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.gn_gather_destroy.argtypes = [vp]
    L.gn_gather_run_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), vp, u32, u32]
    L.gn_stream_device_offsets.argtypes = [vp, C.POINTER(vp)]
    L.gn_ablate.argtypes = [C.c_char_p]
    L.gn_filter_probe.argtypes = [vp, vp, u64, vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.gn_stream_hibf_levels.argtypes = [vp, C.POINTER(u32), vp, vp, vp, vp, u32]
    L.gn_stream_hibf_level_lines.argtypes = [vp, vp, u32]
    L.gn_device_memory.argtypes = [i32, C.POINTER(u64), C.POINTER(u64)]
    L.gn_ibf_hash_constants.argtypes = [vp, C.POINTER(u64)]
    L.gn_inflate_set_turns.argtypes = [vp, u32, u32]
    L.gn_inflate_handoff.argtypes = [vp, vp]
    L.gn_stream_upload_text_pair_devices.argtypes = [vp, vp, u64, i32, vp, u64, i32, i32]
    L.gn_hibf_row_stride_words.argtypes = [u64]
    L.gn_hibf_row_stride_words.restype = u64
    L.gn_peer_stats.argtypes = [i32, i32, C.POINTER(i32), C.POINTER(u64)]
    L.gn_reassign_create.argtypes = [i32, u64, u64, u32, vp, vp, C.POINTER(vp)]
    L.gn_reassign_run.argtypes = [vp, u32, C.c_double, C.POINTER(u32)]
    L.gn_reassign_diffs.argtypes = [vp, vp, u32]
    L.gn_reassign_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.gn_reassign_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_float), C.POINTER(u64)]
    L.gn_reassign_free.argtypes = [vp]
    L.gn_inflate_create.argtypes = [i32, u64, u32, u64, C.POINTER(vp)]
    L.gn_inflate_destroy.argtypes = [vp]
    L.gn_inflate_feed.argtypes = [vp, vp, u64]
    L.gn_inflate_step.argtypes = [vp, C.POINTER(u64), C.POINTER(i32)]
    L.gn_inflate_text.argtypes = [vp, vp, u64, u64]
    L.gn_inflate_text_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.gn_inflate_get_stats.argtypes = [vp, C.POINTER(InflateStats)]
    L.gn_inflate_cuts.argtypes = [vp, u32, u64, vp, u32, C.POINTER(u32)]
    L.gn_inflate_set_carry.argtypes = [vp, u64]
    L.gn_inflate_cuts_lines.argtypes = [vp, u32, u64, vp, vp, u32, C.POINTER(u32)]
    L.gn_inflate_cut_at_lines.argtypes = [vp, vp, u32, vp, C.POINTER(u64)]
    L.gn_stream_upload_text_pair_device.argtypes = [vp, vp, u64, vp, u64, i32, i32]
    L.gn_stream_fetch_letters.argtypes = [vp, vp, u64, vp, vp, C.POINTER(u64)]
    L.gn_stream_upload_text_device.argtypes = [vp, vp, u64, i32, i32]
    L.gn_stream_fastq_headers.argtypes = [vp, vp, u64, vp, C.POINTER(u64)]
    for name in ABI_SYMBOLS:
        if name not in ("gn_last_error", "gn_hibf_row_stride_words"):   # (const char* and uint64_t returns, set above)
            getattr(L, name).restype = i32
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise GanonHipError(rc, load_library().gn_last_error().decode(errors="replace"))


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int(0)
    _check(load_library().gn_device_count(C.byref(n)))
    return n.value


def _desc(rows: Optional[np.ndarray], bins: int, bin_size: int, hash_funs: int) -> IbfDesc:
    W = (bins + 63) >> 6
    shift = 64 - int(bin_size).bit_length()
    if rows is not None:
        assert rows.dtype == np.uint64 and rows.size == bin_size * W and rows.flags["C_CONTIGUOUS"]
    return IbfDesc(rows.ctypes.data if rows is not None else None, bin_size, W, bins, hash_funs, shift)


def _mix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill_random_words(seed: int, rows: np.ndarray, n_words: int, and_words: int = 1, word_lo: int = 0,
                      row_words_total: int = 0, bins: int = 0) -> np.ndarray:
    """numpy statement of gn_filter_fill_random (include/ganon_hip.h): words [word_lo, word_lo+n_words) of the given
    rows -> uint64 [len(rows), n_words].  `bins` (of this column slice) clears the padding bins of its last word."""
    rows = np.asarray(rows, dtype=np.uint64).reshape(-1, 1)
    total = np.uint64(row_words_total or n_words)
    with np.errstate(over="ignore"):
        g = (rows * total + np.uint64(word_lo) + np.arange(n_words, dtype=np.uint64)[None, :]) * np.uint64(0x9E3779B97F4A7C15)
        v = np.full(g.shape, np.uint64(0xFFFFFFFFFFFFFFFF))
        key = [_mix64(np.array([(seed + a) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0] for a in range(3 if and_words == FILL_3_OF_8 else 4 if and_words == FILL_3_OF_16 else and_words)]
        if and_words == FILL_3_OF_8:
            v = _mix64(key[0] + g) & (_mix64(key[1] + g) | _mix64(key[2] + g))
        elif and_words == FILL_3_OF_16:
            v = _mix64(key[0] + g) & _mix64(key[1] + g) & (_mix64(key[2] + g) | _mix64(key[3] + g))
        else:
            for a in range(and_words):
                v &= _mix64(key[a] + g)
    if bins & 63:
        v[:, -1] &= np.uint64((1 << (bins & 63)) - 1)
    return v


def ibf_hash_constants() -> Tuple[list, int]:
    """(five seeds, multiplier) of hash_and_fit as libganon_hip.so was built (gn_ibf_hash_constants; no device needed)"""
    seeds = np.zeros(5, dtype=np.uint64)
    mul = C.c_uint64(0)
    _check(load_library().gn_ibf_hash_constants(_p(seeds), C.byref(mul)))
    return [int(x) for x in seeds], int(mul.value)


def device_memory(device: int = 0) -> Tuple[int, int]:
    """(free, total) bytes of a device (gn_device_memory)"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(load_library().gn_device_memory(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


class HipInflate:
    """gn_inflate_*: a gzip file inflated on the device (include/ganon_hip.h).  `inflate(data)` = the whole file at once."""

    def __init__(self, compressed_bytes: int, device: int = 0, chunk_bytes: int = 0, step_bytes: int = 0):
        self._L = load_library()
        self._h = C.c_void_p()
        _check(self._L.gn_inflate_create(device, compressed_bytes, chunk_bytes, step_bytes, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            self._L.gn_inflate_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        _check(self._L.gn_inflate_feed(self._h, _p(data), data.size))

    def step(self) -> Tuple[int, bool]:
        n, done = C.c_uint64(0), C.c_int(0)
        _check(self._L.gn_inflate_step(self._h, C.byref(n), C.byref(done)))
        return int(n.value), bool(done.value)

    def text(self, n: int, off: int = 0) -> np.ndarray:
        out = np.empty(n, dtype=np.uint8)
        _check(self._L.gn_inflate_text(self._h, _p(out), off, n))
        return out

    def text_device(self) -> Tuple[int, int]:
        """(device pointer, bytes) of the last step's text"""
        ptr, n = C.c_void_p(), C.c_uint64()
        _check(self._L.gn_inflate_text_device(self._h, C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def cuts(self, lines_per_record: int, piece_bytes: int) -> np.ndarray:
        """gn_inflate_cuts: record boundaries of the last step's text, one at or behind every multiple of piece_bytes + the last one"""
        _, n = self.text_device()
        out = np.empty(n // max(1, piece_bytes) + 2, dtype=np.uint64)
        k = C.c_uint32()
        _check(self._L.gn_inflate_cuts(self._h, lines_per_record, piece_bytes, _p(out), out.size, C.byref(k)))
        return out[:k.value].copy()

    def set_carry(self, n_tail: int) -> None:
        _check(self._L.gn_inflate_set_carry(self._h, n_tail))

    def set_turns(self, n_turns: int, my_turn: int) -> None:
        """gn_inflate_set_turns: this inflater runs steps my_turn, my_turn + n_turns, ... of a file that n_turns inflaters share"""
        _check(self._L.gn_inflate_set_turns(self._h, n_turns, my_turn))

    def handoff(self, to: "HipInflate") -> None:
        """gn_inflate_handoff: what the next step needs of the one this inflater just ran goes to `to` (device to device)"""
        _check(self._L.gn_inflate_handoff(self._h, to._h))

    def stats(self) -> dict:
        st = InflateStats()
        _check(self._L.gn_inflate_get_stats(self._h, C.byref(st)))
        return {k: (list(getattr(st, k)) if k == "prof_ms" else getattr(st, k)) for k, _ in InflateStats._fields_}

    def inflate_all(self, data: np.ndarray, feed_bytes: int = 0, fetch: bool = True):
        """feeds `data` (the whole file) in pieces of feed_bytes (0: at once), steps until the stream ends; returns the text
        (fetch=False: only its length -- timing runs)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        parts, total, fed = [], 0, 0
        fb = feed_bytes or data.size
        done = False
        while not done:
            if fed < data.size:
                self.feed(data[fed:fed + fb])
                fed = min(data.size, fed + fb)
                if fed < data.size and fed < 2 * fb + (4 << 20):
                    continue
            n, done = self.step()
            if fetch and n:
                parts.append(self.text(n))
            total += n
        if not fetch:
            return total
        return np.concatenate(parts) if parts else np.empty(0, dtype=np.uint8)


class HipReassign:
    """gn_reassign_*: the EM of `ganon reassign` over a CSR table of (read -> entries naming targets), on the device"""

    def __init__(self, off: np.ndarray, target: np.ndarray, n_targets: int, device: int = 0):
        self._off = np.ascontiguousarray(off, dtype=np.uint64)
        self._target = np.ascontiguousarray(target, dtype=np.uint32)
        self.n_reads, self.n_targets = self._off.size - 1, int(n_targets)
        self._h = C.c_void_p()
        _check(load_library().gn_reassign_create(device, self.n_reads, self._target.size, self.n_targets, _p(self._off), _p(self._target),
                                                 C.byref(self._h)))

    def run(self, max_iter: int = 10, threshold: float = 0.0):
        """-> (diffs float64[iterations], counts uint64[n_targets], unique uint64[n_targets], prob float64[n_targets],
        choice uint64[n_reads] = entry index per read)"""
        L = load_library()
        it = C.c_uint32(0)
        _check(L.gn_reassign_run(self._h, max_iter, float(threshold), C.byref(it)))
        diffs = np.zeros(it.value, dtype=np.float64)
        _check(L.gn_reassign_diffs(self._h, _p(diffs), it.value))
        counts = np.zeros(self.n_targets, dtype=np.uint64)
        unique = np.zeros(self.n_targets, dtype=np.uint64)
        prob = np.zeros(self.n_targets, dtype=np.float64)
        choice = np.zeros(self.n_reads, dtype=np.uint64)
        _check(L.gn_reassign_fetch(self._h, _p(counts), _p(unique), _p(prob), _p(choice)))
        return diffs, counts, unique, prob, choice

    def info(self) -> dict:
        nu, nm, nw, by = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ms = C.c_float(0)
        _check(load_library().gn_reassign_info(self._h, C.byref(nu), C.byref(nm), C.byref(nw), C.byref(ms), C.byref(by)))
        return dict(unique_reads=nu.value, multi_reads=nm.value, wave_reads=nw.value, ms=ms.value, bytes_per_iteration=by.value)

    def free(self):
        if self._h:
            load_library().gn_reassign_free(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()
        return False

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class HipGather:
    """gn_gather: the column parts of a bin-range partitioned flat IBF put back together on the batch's owner device."""

    def __init__(self, device: int, target_maps: Sequence[Optional[np.ndarray]]):
        self._maps = [None if m is None else np.ascontiguousarray(m, dtype=np.uint32) for m in target_maps]
        n = len(self._maps)
        ptrs = (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in self._maps])
        sizes = np.array([0 if m is None else len(m) for m in self._maps], dtype=np.uint32)
        self._h = C.c_void_p()
        self.n_reads = 0
        _check(load_library().gn_gather_create(device, n, ptrs, _p(sizes), C.byref(self._h)))

    def run(self, streams: Sequence["HipStream"]) -> None:
        arr = (C.c_void_p * len(streams))(*[st._h for st in streams])
        self.n_reads = streams[0].n_reads
        _check(load_library().gn_gather_run(self._h, arr, len(streams)))

    def run_buffers(self, off_ptrs: Sequence[int], match_ptrs: Sequence[int], n_matches: Sequence[int], n_reads: int) -> None:
        """gn_gather_run_buffers: parts given as raw device pointers (offsets, records) on the gather's device"""
        k = len(off_ptrs)
        a = (C.c_void_p * k)(*[int(x) for x in off_ptrs])
        b = (C.c_void_p * k)(*[int(x) if x else None for x in match_ptrs])
        nm = np.asarray(n_matches, dtype=np.uint64)
        self.n_reads = int(n_reads)
        _check(load_library().gn_gather_run_buffers(self._h, a, b, _p(nm), k, int(n_reads)))

    def fetch(self):
        """-> (match_off u64[n+1], matches MATCH_DTYPE[m]) grouped by read, ascending target"""
        L = load_library()
        mo = np.zeros(self.n_reads + 1, dtype=np.uint64)
        need = C.c_uint64(0)
        _check(L.gn_gather_fetch(self._h, _p(mo), None, 0, C.byref(need)))
        m = np.zeros(max(int(need.value), 1), dtype=MATCH_DTYPE)
        _check(L.gn_gather_fetch(self._h, None, _p(m), len(m), C.byref(need)))
        return mo, m[: int(need.value)]

    def peer_bytes(self) -> int:
        """bytes the last run() copied between devices"""
        b = C.c_uint64(0)
        _check(load_library().gn_gather_device_matches(self._h, None, None, None, C.byref(b)))
        return int(b.value)

    def destroy(self) -> None:
        if self._h:
            load_library().gn_gather_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipFilter:
    """Device-resident, immutable IBF / HIBF (gn_filter)."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    @classmethod
    def ibf(cls, rows: Optional[np.ndarray], bins: int, bin_size: int, hash_funs: int,
            bin2target: Optional[np.ndarray] = None, n_targets: Optional[int] = None, device: int = 0) -> "HipFilter":
        if bin2target is None:
            bin2target = np.arange(bins, dtype=np.uint32)
            n_targets = bins
        bin2target = np.ascontiguousarray(bin2target, dtype=np.uint32)
        if n_targets is None:
            n_targets = int(bin2target[bin2target != 0xFFFFFFFF].max()) + 1
        d = _desc(rows, bins, bin_size, hash_funs)
        h = C.c_void_p()
        _check(load_library().gn_filter_upload_ibf(device, C.byref(d), _p(bin2target), n_targets, C.byref(h)))
        return cls(h)

    @classmethod
    def hibf(cls, ibfs: Sequence[Tuple[np.ndarray, int, int, int]], next_ibf_id: Sequence[np.ndarray],
             bin2user: Sequence[np.ndarray], n_user_bins: int, device: int = 0) -> "HipFilter":
        """ibfs: sequence of (rows, bins, bin_size, hash_funs)."""
        n = len(ibfs)
        descs = (IbfDesc * n)(*[_desc(r, b, s, hf) for r, b, s, hf in ibfs])
        nx = [np.ascontiguousarray(a, dtype=np.int64) for a in next_ibf_id]
        bu = [np.ascontiguousarray(a, dtype=np.int64) for a in bin2user]
        nxp = (C.c_void_p * n)(*[a.ctypes.data for a in nx])
        bup = (C.c_void_p * n)(*[a.ctypes.data for a in bu])
        h = C.c_void_p()
        _check(load_library().gn_filter_upload_hibf(device, n, descs, nxp, bup, n_user_bins, C.byref(h)))
        return cls(h, keep=(nx, bu))

    def emplace(self, hashes: np.ndarray, bins: np.ndarray, ibf_idx: int = 0) -> None:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        bins = np.ascontiguousarray(bins, dtype=np.uint32)
        assert len(hashes) == len(bins)
        _check(load_library().gn_filter_emplace_ibf(self._h, ibf_idx, _p(hashes), _p(bins), len(hashes)))

    def probe(self, hashes: np.ndarray, bins: np.ndarray) -> Tuple[int, int, int]:
        """gn_filter_probe -> (hits summed over the bins, hashes in none of the bins, index of the first such hash or -1)"""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        bins = np.ascontiguousarray(bins, dtype=np.uint32)
        hits, missing, first = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(load_library().gn_filter_probe(self._h, _p(hashes), len(hashes), _p(bins), len(bins), C.byref(hits), C.byref(missing), C.byref(first)))
        return hits.value, missing.value, -1 if first.value == 0xFFFFFFFFFFFFFFFF else first.value

    def fill_random(self, seed: int, and_words: int = 1, word_lo: int = 0, row_words_total: int = 0, ibf_idx: int = 0) -> None:
        """device-side seeded Bernoulli(2**-and_words) fill (gn_filter_fill_random); fill_random_words() is its numpy twin"""
        _check(load_library().gn_filter_fill_random(self._h, ibf_idx, seed & 0xFFFFFFFFFFFFFFFF, and_words, word_lo,
                                                    row_words_total))

    def write_rows(self, row_begin: int, src: np.ndarray, word_lo: int = 0, ibf_idx: int = 0) -> None:
        """streaming upload of rows [row_begin, row_begin+len(src)) from a host chunk [n_rows, src_row_words]"""
        src = np.ascontiguousarray(src, dtype=np.uint64)
        assert src.ndim == 2
        L = load_library()
        _check(L.gn_filter_write_rows(self._h, ibf_idx, row_begin, src.shape[0], _p(src), src.shape[1], word_lo))
        _check(L.gn_filter_write_sync(self._h))

    def finalize(self) -> None:
        _check(load_library().gn_filter_finalize(self._h))

    def download_row_list(self, row_idx: np.ndarray, words: int, ibf_idx: int = 0) -> np.ndarray:
        row_idx = np.ascontiguousarray(row_idx, dtype=np.uint64)
        out = np.empty((len(row_idx), words), dtype=np.uint64)
        _check(load_library().gn_filter_download_row_list(self._h, ibf_idx, _p(row_idx), len(row_idx), _p(out)))
        return out

    def download_rows(self, row_begin: int, n_rows: int, words: int, ibf_idx: int = 0) -> np.ndarray:
        out = np.empty((n_rows, words), dtype=np.uint64)
        _check(load_library().gn_filter_download_rows(self._h, ibf_idx, row_begin, n_rows, _p(out)))
        return out

    def info(self) -> dict:
        a, b, c, d = C.c_int(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(load_library().gn_filter_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(is_hibf=bool(a.value), n_ibf=b.value, n_targets=c.value, device_bytes=d.value)

    def free(self) -> None:
        if self._h:
            load_library().gn_filter_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipStream:
    """Batch context (gn_stream): upload reads, classify, fetch sparse matches."""

    def __init__(self, flt: HipFilter, max_reads: int, max_bases: int, max_matches: int = 0):
        self._f = flt
        self._h = C.c_void_p()
        _check(load_library().gn_stream_create(flt._h, max_reads, max_bases, max_matches, C.byref(self._h)))
        self.n_reads = 0

    def upload(self, bases: np.ndarray, off1: np.ndarray, off2: Optional[np.ndarray] = None) -> None:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off1 = np.ascontiguousarray(off1, dtype=np.uint64)
        if off2 is not None:
            off2 = np.ascontiguousarray(off2, dtype=np.uint64)
        self.n_reads = len(off1) - 1
        self._keep = (bases, off1, off2)
        _check(load_library().gn_stream_upload_reads(self._h, _p(bases), bases.size, _p(off1), _p(off2), self.n_reads))

    def minimisers(self, k: int, w: int) -> None:
        """hash only (no filter lookup); results via fetch_hashes()"""
        _check(load_library().gn_stream_minimisers(self._h, k, w))

    def classify(self, k: int, w: int, rel_cutoff: float) -> None:
        _check(load_library().gn_stream_classify(self._h, k, w, float(rel_cutoff)))

    def classify_shared(self, source: "HipStream", rel_cutoff: float) -> None:
        """count the batch resident (and hashed) in `source` against this stream's filter too (gn_stream_classify_shared)"""
        self.n_reads = source.n_reads
        _check(load_library().gn_stream_classify_shared(self._h, source._h, float(rel_cutoff)))

    def submit(self, bases, off1, off2, k: int, w: int, rel_cutoff: float) -> None:
        self.upload(bases, off1, off2)
        self.classify(k, w, rel_cutoff)

    def upload_fastq(self, text, fasta=False) -> Tuple[int, int, int]:
        """four-line FASTQ text (or two-line FASTA text), tokenised on the device (gn_stream_upload_text + gn_stream_fastq_index):
        -> (reads, bases, parsed_bytes); the stream then holds the reads like after upload()"""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text, dtype=np.uint8)
        self._keep = (text,)
        L = load_library()
        _check(L.gn_stream_upload_text(self._h, _p(text), text.size, int(fasta)))
        n, nb, pb = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(L.gn_stream_fastq_index(self._h, C.byref(n), C.byref(nb), C.byref(pb)))
        self.n_reads = n.value
        return n.value, nb.value, pb.value

    def upload_text_device(self, d_text: int, n_bytes: int, fasta=False, src_device: int = 0) -> Tuple[int, int, int]:
        """the same for a text that lies in device memory (gn_stream_upload_text_device): -> (reads, bases, parsed_bytes)"""
        L = load_library()
        _check(L.gn_stream_upload_text_device(self._h, C.c_void_p(d_text), n_bytes, int(fasta), src_device))
        n, nb, pb = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(L.gn_stream_fastq_index(self._h, C.byref(n), C.byref(nb), C.byref(pb)))
        self.n_reads = n.value
        return n.value, nb.value, pb.value

    def fastq_headers(self) -> Tuple[bytes, np.ndarray]:
        """gn_stream_fastq_headers: the batch's header lines back to back, and record i's offset (n_reads + 1 entries)"""
        L = load_library()
        off = np.empty(self.n_reads + 1, dtype=np.uint32)
        cap = max(4096, 64 * self.n_reads)
        for _ in range(2):
            dst = np.empty(cap, dtype=np.uint8)
            nb = C.c_uint64()
            rc = L.gn_stream_fastq_headers(self._h, _p(dst), cap, _p(off), C.byref(nb))
            if rc == -75:
                cap = int(nb.value) + 64
                continue
            _check(rc)
            return dst[:nb.value].tobytes(), off
        raise GanonHipError(-75, "gn_stream_fastq_headers: the header lines did not fit twice")

    def upload_text_pair(self, text1, text2, fasta=False) -> Tuple[int, int, int]:
        """the two mate files' pieces as text -> (pairs, parsed_bytes1, parsed_bytes2); the stream then holds the pairs like after upload()"""
        t1, t2 = (np.ascontiguousarray(np.frombuffer(t, dtype=np.uint8) if isinstance(t, (bytes, bytearray)) else t, dtype=np.uint8) for t in (text1, text2))
        self._keep = (t1, t2)
        L = load_library()
        _check(L.gn_stream_upload_text_pair(self._h, _p(t1), t1.size, _p(t2), t2.size, int(fasta)))
        n, p1, p2 = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(L.gn_stream_text_pair_index(self._h, C.byref(n), C.byref(p1), C.byref(p2)))
        self.n_reads = n.value
        return n.value, p1.value, p2.value

    def text_pair_records2(self):
        out = [np.empty(self.n_reads, dtype=np.uint32) for _ in range(3)]
        _check(load_library().gn_stream_text_pair_records2(self._h, _p(out[0]), _p(out[1]), _p(out[2])))
        return tuple(out)

    def fastq_keep(self, n_reads: int) -> None:
        _check(load_library().gn_stream_fastq_keep(self._h, n_reads))
        self.n_reads = n_reads

    def fastq_records(self):
        """(rec_at, seq_at, seq_len): uint32 per read, offsets into the uploaded text"""
        out = [np.empty(self.n_reads, dtype=np.uint32) for _ in range(3)]
        _check(load_library().gn_stream_fastq_records(self._h, _p(out[0]), _p(out[1]), _p(out[2])))
        return tuple(out)

    def sync(self) -> None:
        _check(load_library().gn_stream_sync(self._h))

    def fetch(self):
        """-> (n_hashes u32[n], status u8[n], match_off u64[n+1], matches MATCH_DTYPE[m])"""
        L = load_library()
        n = self.n_reads
        nh = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        mo = np.zeros(n + 1, dtype=np.uint64)
        need = C.c_uint64(0)
        _check(L.gn_fetch_batch(self._h, _p(nh), _p(st), _p(mo), None, 0, C.byref(need)))
        m = np.zeros(max(int(need.value), 1), dtype=MATCH_DTYPE)
        _check(L.gn_fetch_batch(self._h, None, None, None, _p(m), len(m), C.byref(need)))
        return nh, st, mo, m[: int(need.value)]

    def set_postfilter(self, rel_filter: Optional[float] = None, fpr_query: float = 1.0, target_fpr=None, joint=False, target_gid=None) -> None:
        """device-side pre-pass of filter_matches on the following batches (gn_stream_set_postfilter); rel_filter=None: off"""
        L = load_library()
        if rel_filter is None:
            _check(L.gn_stream_set_postfilter(self._h, None))
            return
        tf = None if target_fpr is None else np.ascontiguousarray(target_fpr, dtype=np.float64)
        tg = None if target_gid is None else np.ascontiguousarray(target_gid, dtype=np.uint32)
        pf = PostFilter(float(rel_filter), float(fpr_query), None if tf is None else tf.ctypes.data_as(C.c_void_p),
                        2 if tg is not None else (1 if joint else 0), None if tg is None else tg.ctypes.data_as(C.c_void_p))
        _check(L.gn_stream_set_postfilter(self._h, C.byref(pf)))

    def set_long_reads(self, on: bool = True) -> None:
        """classify reads with more than 65535 minimisers too (the reference's -DLONGREADS build); flat IBF only"""
        _check(load_library().gn_stream_set_long_reads(self._h, 1 if on else 0))

    @staticmethod
    def postfilter_joint(streams) -> None:
        """gn_streams_postfilter_joint: the pre-pass over the streams of one hierarchy level (same batch, disjoint targets)"""
        arr = (C.c_void_p * len(streams))(*[st._h for st in streams])
        _check(load_library().gn_streams_postfilter_joint(arr, len(streams)))

    def fetch_postfilter(self):
        """-> (max_count u32[n] before filtering, dropped by --rel-filter, dropped by --fpr-query) of the last batch"""
        mx = np.zeros(self.n_reads, dtype=np.uint32)
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(load_library().gn_fetch_postfilter(self._h, _p(mx), C.byref(a), C.byref(b)))
        return mx, int(a.value), int(b.value)

    def distinct_hashes(self) -> np.ndarray:
        """sorted distinct minimiser hashes of the resident sequences (gn_stream_distinct_hashes)"""
        L = load_library()
        n = C.c_uint64(0)
        _check(L.gn_stream_distinct_hashes(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.uint64)
        _check(L.gn_stream_distinct_hashes(self._h, _p(out), len(out), C.byref(n)))
        return out[: int(n.value)]

    def fetch_read_info(self):
        """-> (n_hashes u32[n], status u8[n]) only"""
        n = self.n_reads
        nh = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        need = C.c_uint64(0)
        _check(load_library().gn_fetch_batch(self._h, _p(nh), _p(st), None, None, 0, C.byref(need)))
        return nh, st

    def device_records(self, device_index: int = 0):
        """the batch's matches as an int32 torch tensor [m, 3] (read, target, count) that ALIASES the library's device
        buffer (gn_stream_device_matches); valid until the next submit on this stream"""
        import torch
        ptr, n = C.c_void_p(), C.c_uint64(0)
        _check(load_library().gn_stream_device_matches(self._h, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return torch.empty((0, 3), dtype=torch.int32, device=f"cuda:{device_index}")

        class _Dev:  # __cuda_array_interface__ v2: torch wraps the pointer without copying
            __cuda_array_interface__ = {"data": (int(ptr.value), False), "shape": (int(n.value), 3), "typestr": "<i4",
                                        "version": 2}
        return torch.as_tensor(_Dev(), device=f"cuda:{device_index}")

    def device_offsets(self, device_index: int = 0):
        """the n_reads+1 per-read offsets that go with device_records(), as an int64 torch tensor ALIASING the library's buffer"""
        import torch
        ptr = C.c_void_p()
        _check(load_library().gn_stream_device_offsets(self._h, C.byref(ptr)))

        class _Dev:
            __cuda_array_interface__ = {"data": (int(ptr.value), False), "shape": (self.n_reads + 1,), "typestr": "<i8", "version": 2}
        return torch.as_tensor(_Dev(), device=f"cuda:{device_index}")

    def fetch_hashes(self):
        """-> (hash_off u64[n+1], hashes u64[total]) in emission order (parity tap)."""
        L = load_library()
        n = self.n_reads
        ho = np.zeros(n + 1, dtype=np.uint64)
        tot = C.c_uint64(0)
        _check(L.gn_stream_fetch_hashes(self._h, _p(ho), None, 0, C.byref(tot)))
        hs = np.zeros(max(int(tot.value), 1), dtype=np.uint64)
        _check(L.gn_stream_fetch_hashes(self._h, _p(ho), _p(hs), len(hs), C.byref(tot)))
        return ho, hs[: int(tot.value)]

    def dense_counts(self, read_begin: int, read_end: int, width: int) -> np.ndarray:
        out = np.zeros((read_end - read_begin, width), dtype=np.uint16)
        _check(load_library().gn_stream_dense_counts(self._h, read_begin, read_end, _p(out)))
        return out

    def timings(self) -> dict:
        t = Timings()
        _check(load_library().gn_stream_timings(self._h, C.byref(t)))
        return dict(ms_minimiser=t.ms_minimiser, ms_count=t.ms_count, ms_total=t.ms_total, n_hashes=t.n_hashes,
                    algo_bytes=t.algo_bytes, n_matches=t.n_matches, n_count_launches=t.n_count_launches,
                    fetched_bytes=t.fetched_bytes, ms_compact=t.ms_compact)

    def hibf_levels(self):
        """per tree level of the last HIBF batch: [dict(ms, algo_bytes, table_bytes, row_bytes, line_bytes)] (gn_stream_hibf_levels / _level_lines)"""
        cap = 8
        n = C.c_uint32(0)
        ms = np.zeros(cap, dtype=np.float32)
        ab = np.zeros(cap, dtype=np.uint64)
        tb = np.zeros(cap, dtype=np.uint64)
        rb = np.zeros(cap, dtype=np.uint32)
        _check(load_library().gn_stream_hibf_levels(self._h, C.byref(n), _p(ms), _p(ab), _p(tb), _p(rb), cap))
        lb = np.zeros(cap, dtype=np.uint64)
        _check(load_library().gn_stream_hibf_level_lines(self._h, _p(lb), cap))
        return [dict(ms=float(ms[i]), algo_bytes=int(ab[i]), table_bytes=int(tb[i]), row_bytes=int(rb[i]), line_bytes=int(lb[i])) for i in range(min(cap, n.value))]

    def destroy(self) -> None:
        if self._h:
            load_library().gn_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
