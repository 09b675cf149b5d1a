#!/usr/bin/env python3
"""Device inflate (gn_inflate_*, csrc/gn_inflate.hip) on synthetic FASTQ: bytes against zlib's, then rates.

  python scripts/inflate_probe.py [--reads N] [--level L] [--chunk BYTES] [--step BYTES] [--reps R] [--out FILE.jsonl]
"""
import argparse
import gzip
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganon_amd import hip  # noqa: E402


def synth_fastq(n_reads: int, read_len: int = 100, seed: int = 1) -> bytes:
    """Illumina-like records: structured ids, bases from a small genome with errors, qualities from a few-level distribution"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=2_000_000, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    pos = rng.integers(0, genome.size - read_len, size=n_reads)
    idx = pos[:, None] + np.arange(read_len)[None, :]
    bases = lut[genome[idx]]
    err = rng.random((n_reads, read_len)) < 0.01
    bases[err] = lut[rng.integers(0, 4, size=int(err.sum()))]
    q = rng.choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8), size=(n_reads, read_len))
    out = []
    tile = rng.integers(1101, 2678, size=n_reads)
    x = rng.integers(1000, 32000, size=n_reads)
    y = rng.integers(1000, 32000, size=n_reads)
    for i in range(n_reads):
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n" % (tile[i], x[i], y[i]))
        out.append(bases[i].tobytes())
        out.append(b"\n+\n")
        out.append(q[i].tobytes())
        out.append(b"\n")
    return b"".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--step", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tile", type=int, default=1, help="repeat the compressed member this many times (multi-member file)")
    ap.add_argument("--bgzf", action="store_true", help="blocked gzip (members of 64 KiB text with the BC field, as bgzip writes) instead of one member")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    text = synth_fastq(a.reads)
    t0 = time.time()
    if a.bgzf:
        out = []
        for at in list(range(0, len(text), 65280)) + [None]:
            piece = b"" if at is None else text[at:at + 65280]
            co = zlib.compressobj(a.level, zlib.DEFLATED, -15)
            body = co.compress(piece) + co.flush()
            out.append(bytes([0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 66, 67, 2, 0]) + (12 + 6 + len(body) + 8 - 1).to_bytes(2, "little") + body
                       + zlib.crc32(piece).to_bytes(4, "little") + (len(piece) & 0xFFFFFFFF).to_bytes(4, "little"))
        gz = b"".join(out)
    else:
        co = zlib.compressobj(a.level, zlib.DEFLATED, 31)
        gz = co.compress(text) + co.flush()
    t_c = time.time() - t0
    t0 = time.time()
    if a.bgzf:
        d, back, rest = zlib.decompressobj(31), [], gz
        while rest:
            back.append(d.decompress(rest))
            rest = d.unused_data
            d = zlib.decompressobj(31)
        back = b"".join(back)
    else:
        back = zlib.decompress(gz, 31)
    t_z = time.time() - t0
    assert back == text
    if a.tile > 1:
        gz = gz * a.tile
        text = text * a.tile
    data = np.frombuffer(gz, dtype=np.uint8)
    rec = {"reads": a.reads * a.tile, "level": a.level, "text_bytes": len(text), "gz_bytes": len(gz), "ratio": len(text) / len(gz),
           "zlib_inflate_MBps_1thread": len(text) / a.tile / t_z / 1e6, "chunk": a.chunk, "step": a.step}
    with hip.HipInflate(data.size, chunk_bytes=a.chunk, step_bytes=a.step) as z:
        got = z.inflate_all(data)
        st = z.stats()
    ok = got.size == len(text) and got.tobytes() == text
    rec["bytes_equal"] = bool(ok)
    if not ok:
        ref = np.frombuffer(text, dtype=np.uint8)
        n = min(got.size, ref.size)
        bad = np.nonzero(got[:n] != ref[:n])[0]
        rec["got_bytes"] = int(got.size)
        rec["first_diff"] = int(bad[0]) if bad.size else n
    rec["first_run"] = st
    best = None
    for _ in range(a.reps):
        with hip.HipInflate(data.size, chunk_bytes=a.chunk, step_bytes=a.step) as z:
            t0 = time.time()
            n = z.inflate_all(data, fetch=False)
            dt = time.time() - t0
            st = z.stats()
        st["wall_s"] = dt
        st["text_GBps_wall"] = n / dt / 1e9
        dev_ms = st["ms_decode"] + st["ms_chain"] + st["ms_resolve"]
        st["text_GBps_device"] = n / (dev_ms / 1e3) / 1e9 if dev_ms else 0
        if best is None or st["text_GBps_device"] > best["text_GBps_device"]:
            best = st
    rec["best"] = best
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "a") as f:
            f.write(line + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
