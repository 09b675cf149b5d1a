"""`.ibf` files from the Python side of the C ABI: streaming load (whole filter or one column slice), save, and a
minimal device-side build.

  load_ibf(path, device, word_lo, word_hi)  parse the ganon-build file (reader /root/reference/src/ganon-classify/GanonClassify.cpp:949-986,
        writer src/ganon-build/GanonBuild.cpp:251-288) and stream its bit matrix into HBM in pinned,
        double-buffered chunks (gn_filter_write_rows); with word_lo/word_hi only those 64-bin words of every row are
        kept -- one rank's column slice of a bin-range partitioned filter (ganon_amd.partition) -- so that no rank ever
        materialises the whole matrix, on the host or on the device
  save_ibf(path, ...)                       the same layout written from a device filter (rows are downloaded in chunks)
  build_ibf(targets, k, w, ...)             minimisers of the target sequences on the device (gn_stream_minimisers), distinct
        hashes per target, bins assigned like create_bin_map_hash (GanonBuild.cpp:619-653), bits set by gn_filter_emplace
        (GanonBuild.cpp:694).  The size optimiser of ganon-build (:428-616) is NOT restated: bin size and number of hash
        functions come from the textbook Bloom formula for the requested false-positive rate unless given.

The C++ host (ganon_amd/host/filter_io.cpp) has its own loader; this module serves the multi-process Python driver of
the partitioned path, the benchmarks and self-hosted fixtures.  No oracle, no CPU fallback.
"""
from __future__ import annotations

import concurrent.futures as cf
import ctypes as C
import math
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hip


@dataclass
class IbfFileMeta:
    version: Tuple[int, int, int] = (0, 0, 0)
    config: dict = field(default_factory=dict)           # IBFConfig.hpp:28-40
    hashes_count: List[Tuple[str, int]] = field(default_factory=list)
    bin_map: List[Tuple[int, str]] = field(default_factory=list)
    bins: int = 0
    technical_bins: int = 0
    bin_size: int = 0
    hash_shift: int = 0
    bin_words: int = 0
    hash_funs: int = 0
    payload_offset: int = 0

    @property
    def payload_bytes(self) -> int:
        return self.bin_size * self.bin_words * 8

    def targets(self) -> Tuple[List[str], np.ndarray]:
        """target names in first-appearance order of ascending bins, and bin -> target index (0xFFFFFFFF: none)"""
        names: List[str] = []
        idx: Dict[str, int] = {}
        b2t = np.full(self.bins, 0xFFFFFFFF, dtype=np.uint32)
        for b, t in sorted(self.bin_map):
            if t not in idx:
                idx[t] = len(names)
                names.append(t)
            b2t[b] = idx[t]
        return names, b2t


class IbfFormatError(RuntimeError):
    pass


def _rd(f, fmt):
    n = struct.calcsize(fmt)
    b = f.read(n)
    if len(b) != n:
        raise IbfFormatError("truncated .ibf header")
    return struct.unpack(fmt, b)


def _rd_str(f) -> str:
    (n,) = _rd(f, "<Q")
    if n > (1 << 20):
        raise IbfFormatError(f"implausible string length {n}")
    return f.read(n).decode()


def read_ibf_meta(path: str) -> IbfFileMeta:
    """everything but the bits; the same self-checks as the C++ loader, including the bit_vector header variants"""
    m = IbfFileMeta()
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        m.version = _rd(f, "<3i")
        keys = ("n_bins", "max_hashes_bin", "hash_functions", "kmer_size", "window_size", "bin_size_bits", "max_fp", "true_max_fp",
                "true_avg_fp")
        m.config = dict(zip(keys, _rd(f, "<QQBBHQddd")))
        (n,) = _rd(f, "<Q")
        for _ in range(n):
            t = _rd_str(f)
            m.hashes_count.append((t, _rd(f, "<Q")[0]))
        (n,) = _rd(f, "<Q")
        for _ in range(n):
            (b,) = _rd(f, "<Q")
            m.bin_map.append((b, _rd_str(f)))
        m.bins, m.technical_bins, m.bin_size, m.hash_shift, m.bin_words, m.hash_funs = _rd(f, "<6Q")
        if (m.bins == 0 or m.bin_size == 0 or m.bin_words != (m.bins + 63) >> 6 or m.technical_bins != 64 * m.bin_words
                or m.hash_shift != 64 - m.bin_size.bit_length() or not 1 <= m.hash_funs <= 5):
            raise IbfFormatError(f"{path}: not a SeqAn3 IBF (inconsistent shape fields)")
        head = size - f.tell() - m.payload_bytes      # the payload is last: what precedes it is the bit_vector header
        bits = m.technical_bins * m.bin_size
        layouts = {13: "<BfQ", 8: "<Q", 9: "<BQ", 12: "<fQ"}
        if head not in layouts:
            raise IbfFormatError(f"{path}: {head} bytes between the IBF fields and a payload of {m.payload_bytes} bytes")
        vals = _rd(f, layouts[head])
        width = vals[0] if layouts[head][1] == "B" else 1
        if width != 1 or vals[-1] not in (bits, bits // 64):
            raise IbfFormatError(f"{path}: unexpected sdsl bit_vector header {vals}")
        m.payload_offset = f.tell()
    if (m.bins, m.bin_size, m.hash_funs) != (m.config["n_bins"], m.config["bin_size_bits"], m.config["hash_functions"]):
        raise IbfFormatError(f"{path}: IBFConfig disagrees with the stored IBF")
    return m


class _Pinned:
    """numpy view of a gn_pinned_alloc buffer"""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        hip._check(hip.load_library().gn_pinned_alloc(nbytes, C.byref(self.ptr)))
        self.arr = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def free(self):
        if self.ptr:
            hip.load_library().gn_pinned_free(self.ptr)
            self.ptr = C.c_void_p()


def _parallel_io(fn, fd: int, buf: np.ndarray, offset: int, threads: int) -> None:
    """pread / pwrite `buf` at `offset` in parallel slices (both release the GIL)"""
    n = buf.size
    parts = max(1, min(threads, n >> 23))
    step = ((n + parts - 1) // parts + 4095) & ~4095
    spans = [(a, min(n, a + step)) for a in range(0, n, step)]

    def work(span):
        a, b = span
        mv = memoryview(buf[a:b])
        done = 0
        while done < b - a:
            got = fn(fd, mv[done:], offset + a + done)
            if got <= 0:
                raise IOError("short read/write in the filter payload")
            done += got

    if len(spans) == 1:
        work(spans[0])
    else:
        with cf.ThreadPoolExecutor(max_workers=len(spans)) as ex:
            list(ex.map(work, spans))


def load_ibf(path: str, device: int = 0, word_lo: int = 0, word_hi: Optional[int] = None, bin2target: Optional[np.ndarray] = None,
             n_targets: Optional[int] = None, chunk_bytes: int = 256 << 20, threads: int = 16):
    """-> (HipFilter, IbfFileMeta).  Words [word_lo, word_hi) of every row go to the device; the default is the whole
    filter with the file's own bin -> target map.  For a slice the caller passes the slice's local map (ganon_amd.partition)."""
    m = read_ibf_meta(path)
    word_hi = m.bin_words if word_hi is None else word_hi
    if not 0 <= word_lo < word_hi <= m.bin_words:
        raise ValueError("column slice outside the filter")
    W = word_hi - word_lo
    bins_local = min(m.bins, word_hi * 64) - word_lo * 64
    if bin2target is None:
        names, b2t = m.targets()
        bin2target, n_targets = b2t[word_lo * 64: word_lo * 64 + bins_local], len(names)
    flt = hip.HipFilter.ibf(None, bins_local, m.bin_size, m.hash_funs, bin2target, n_targets, device=device)
    L = hip.load_library()
    row_bytes = m.bin_words * 8
    per = max(1, min(chunk_bytes, m.payload_bytes) // row_bytes)
    stage = [_Pinned(per * row_bytes), _Pinned(per * row_bytes)]
    fd = os.open(path, os.O_RDONLY)
    try:
        for c, row in enumerate(range(0, m.bin_size, per)):
            n = min(per, m.bin_size - row)
            buf = stage[c & 1].arr[: n * row_bytes]
            _parallel_io(lambda fd_, mv, off: os.preadv(fd_, [mv], off), fd, buf, m.payload_offset + row * row_bytes, threads)
            # the copy of the previous chunk (other buffer) ran while this one was read; it must be done before the next round
            hip._check(L.gn_filter_write_sync(flt._h))
            hip._check(L.gn_filter_write_rows(flt._h, 0, row, n, buf.ctypes.data_as(C.c_void_p), m.bin_words, word_lo))
        hip._check(L.gn_filter_finalize(flt._h))
    finally:
        os.close(fd)
        for s in stage:
            s.free()
    return flt, m


def save_ibf(path: str, flt, config: dict, hashes_count: Sequence[Tuple[str, int]], bin_map: Sequence[Tuple[int, str]], bins: int,
             bin_size: int, hash_funs: int, version=(2, 1, 1), chunk_bytes: int = 256 << 20, threads: int = 16) -> None:
    """save_filter (GanonBuild.cpp:251-288) for a flat device filter: header, then the rows downloaded chunk by chunk"""
    W = (bins + 63) >> 6
    shift = 64 - int(bin_size).bit_length()
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", *version))
        f.write(struct.pack("<QQBBHQddd", config["n_bins"], config["max_hashes_bin"], config["hash_functions"], config["kmer_size"],
                            config["window_size"], config["bin_size_bits"], config["max_fp"], config["true_max_fp"],
                            config["true_avg_fp"]))
        f.write(struct.pack("<Q", len(hashes_count)))
        for t, n in hashes_count:
            tb = t.encode()
            f.write(struct.pack("<Q", len(tb)) + tb + struct.pack("<Q", n))
        f.write(struct.pack("<Q", len(bin_map)))
        for b, t in bin_map:
            tb = t.encode()
            f.write(struct.pack("<Q", b) + struct.pack("<Q", len(tb)) + tb)
        f.write(struct.pack("<6Q", bins, W * 64, bin_size, shift, W, hash_funs))
        f.write(struct.pack("<BfQ", 1, 1.5, W * 64 * bin_size))  # sdsl bit_vector header
        payload_at = f.tell()
        f.truncate(payload_at + bin_size * W * 8)
    row_bytes = W * 8
    per = max(1, min(chunk_bytes, bin_size * row_bytes) // row_bytes)
    stage = _Pinned(per * row_bytes)
    fd = os.open(path, os.O_WRONLY)
    try:
        L = hip.load_library()
        for row in range(0, bin_size, per):
            n = min(per, bin_size - row)
            buf = stage.arr[: n * row_bytes]
            hip._check(L.gn_filter_download_rows(flt._h, 0, row, n, buf.ctypes.data_as(C.c_void_p)))
            _parallel_io(lambda fd_, mv, off: os.pwritev(fd_, [mv], off), fd, buf, payload_at + row * row_bytes, threads)
    finally:
        os.close(fd)
        stage.free()


def save_hibf(path: str, flt, ibfs: Sequence[Tuple[int, int, int]], next_ibf_id, bin_to_user, user_bin_names: Sequence[str], k: int, w: int,
              fpr: float, chunk_bytes: int = 256 << 20) -> None:
    """A raptor 3.0.1 index as the reference reads it (GanonClassify.cpp:884-901, hibf.hpp:163-169,293-298) from a
    device-resident HIBF: ibfs[i] = (bins, rows, hash_funs) of IBF i, rows downloaded chunk by chunk.  user_bin_names[u] becomes the
    one file of user bin u (`<name>.minimiser`, with '.' spelt '|||' as raptor's callers do)."""
    L = hip.load_library()
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 1))
        f.write(struct.pack("<Q", w))
        f.write(struct.pack("<QQ", k, (1 << k) - 1))     # seqan3::shape: size, bits
        f.write(struct.pack("<BB", 1, 0))                # parts, compressed
        f.write(struct.pack("<Q", len(user_bin_names)))
        files = []
        for n in user_bin_names:
            fn = ("/db/" + n.replace(".", "|||").replace(" ", "---") + ".minimiser").encode()
            files.append(fn)
            f.write(struct.pack("<QQ", 1, len(fn)) + fn)
        f.write(struct.pack("<d", fpr))
        f.write(struct.pack("<B", 1))                    # is_hibf
        f.write(struct.pack("<Q", len(ibfs)))
        stage = None
        try:
            for i, (bins, rows, h) in enumerate(ibfs):
                W = (bins + 63) >> 6
                f.write(struct.pack("<6Q", bins, W * 64, rows, 64 - int(rows).bit_length(), W, h))
                f.write(struct.pack("<BfQ", 1, 1.5, W * 64 * rows))
                row_bytes = W * 8
                per = max(1, min(chunk_bytes, rows * row_bytes) // row_bytes)
                if stage is None or stage.arr.size < per * row_bytes:
                    if stage is not None:
                        stage.free()
                    stage = _Pinned(max(per * row_bytes, min(chunk_bytes, 16 << 20)))
                for row in range(0, rows, per):
                    n = min(per, rows - row)
                    buf = stage.arr[: n * row_bytes]
                    hip._check(L.gn_filter_download_rows(flt._h, i, row, n, buf.ctypes.data_as(C.c_void_p)))
                    f.write(memoryview(buf))
        finally:
            if stage is not None:
                stage.free()
        f.write(struct.pack("<Q", len(next_ibf_id)))
        for a in next_ibf_id:
            f.write(struct.pack("<Q", len(a)) + np.ascontiguousarray(a, dtype="<i8").tobytes())
        f.write(struct.pack("<Q", len(files)))
        for fn in files:
            f.write(struct.pack("<Q", len(fn)) + fn)
        f.write(struct.pack("<Q", len(bin_to_user)))
        for a in bin_to_user:
            f.write(struct.pack("<Q", len(a)) + np.ascontiguousarray(a, dtype="<i8").tobytes())


def bloom_bin_size(n_hashes: int, max_fp: float, hash_funs: int) -> int:
    """bits of one Bloom filter holding n_hashes at false-positive rate max_fp with hash_funs functions (textbook formula)"""
    return int(math.ceil(n_hashes * (-hash_funs / math.log(1.0 - math.exp(math.log(max_fp) / hash_funs)))))


def false_positive(bin_size_bits: int, hash_funs: int, n_hashes: int) -> float:
    return math.pow(1.0 - math.exp(-hash_funs / (bin_size_bits / float(n_hashes))), hash_funs)


def build_ibf(targets: Dict[str, Sequence[bytes]], k: int, w: int, max_fp: float = 0.05, hash_funs: int = 4,
              max_hashes_bin: Optional[int] = None, bin_size_bits: Optional[int] = None, device: int = 0):
    """-> (HipFilter, save_ibf keyword arguments).  targets: name -> sequences (ASCII bytes)."""
    names = list(targets)
    seqs, owner = [], []
    for ti, t in enumerate(names):
        for s in targets[t]:
            seqs.append(np.frombuffer(bytes(s), dtype=np.uint8))
            owner.append(ti)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.concatenate(seqs) if seqs else np.zeros(1, np.uint8)
    tmp = hip.HipFilter.ibf(None, 64, 64, 1, device=device)  # (a stream needs a filter; hashing does not look at it)
    st = hip.HipStream(tmp, max(len(seqs), 1), max(int(bases.size), 1))
    st.upload(bases, off, None)
    st.minimisers(k, w)
    ho, hs = st.fetch_hashes()
    st.destroy()
    tmp.free()
    per_target: List[np.ndarray] = []
    owner = np.asarray(owner)
    for ti in range(len(names)):  # distinct minimisers of every target (count_hashes, GanonBuild.cpp:184-249)
        parts = [hs[int(ho[i]):int(ho[i + 1])] for i in np.nonzero(owner == ti)[0]]
        per_target.append(np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.uint64))
    counts = [len(x) for x in per_target]
    if max_hashes_bin is None:
        max_hashes_bin = max(max(counts), 1)
    if bin_size_bits is None:
        bin_size_bits = bloom_bin_size(max_hashes_bin, max_fp, hash_funs)
    bin_map, hb, bb = [], [], []
    binno = 0
    for t, hv in zip(names, per_target):  # create_bin_map_hash (GanonBuild.cpp:619-653): equal shares over ceil(count/max) bins
        if len(hv) == 0:
            continue
        nb = int(math.ceil(len(hv) / float(max_hashes_bin)))
        share = min(int(math.ceil(len(hv) / float(nb))), max_hashes_bin)
        for i in range(nb):
            part = hv[i * share:(i + 1) * share]
            if len(part) == 0:
                break
            bin_map.append((binno, t))
            hb.append(part)
            bb.append(np.full(len(part), binno, dtype=np.uint32))
            binno += 1
    n_bins = binno
    names_used, b2t = IbfFileMeta(bin_map=bin_map, bins=n_bins).targets()
    flt = hip.HipFilter.ibf(None, n_bins, bin_size_bits, hash_funs, b2t, len(names_used), device=device)
    if hb:
        flt.emplace(np.concatenate(hb), np.concatenate(bb))
    fps = []
    for c in counts:
        if c:
            nb = int(math.ceil(c / float(max_hashes_bin)))
            fps.append(1.0 - math.pow(1.0 - false_positive(bin_size_bits, hash_funs, int(math.ceil(c / float(nb)))), nb))
    config = dict(n_bins=n_bins, max_hashes_bin=max_hashes_bin, hash_functions=hash_funs, kmer_size=k, window_size=w,
                  bin_size_bits=bin_size_bits, max_fp=max_fp, true_max_fp=max(fps) if fps else 0.0,
                  true_avg_fp=sum(fps) / len(fps) if fps else 0.0)
    return flt, dict(config=config, hashes_count=[(t, c) for t, c in zip(names, counts) if c], bin_map=bin_map, bins=n_bins,
                     bin_size=bin_size_bits, hash_funs=hash_funs)
