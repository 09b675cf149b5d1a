"""host/pgzip.cpp (parallel inflate of ordinary gzip files: speculative block starts, 16-bit symbols with window markers,
windows settled in order) against zlib's own gzread on the same file, through tests/host_oracle/pgzip_check."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def check():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host_oracle"), "-s", "pgzip_check"])
    exe = os.path.join(HERE, "host_oracle", "pgzip_check")

    def run(path, threads, chunk, unit=1 << 20):
        return subprocess.run([exe, path, str(threads), str(chunk), str(unit)], capture_output=True, text=True, check=True).stdout.strip()
    return run


@pytest.fixture(scope="module")
def fastq():
    rng = np.random.default_rng(3)
    n, L = 40_000, 150
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    qs = np.frombuffer(b"FFFFF:FFF,FFFFFFFFFF#", dtype=np.uint8)
    seqs = acgt[rng.integers(0, 4, size=(n, L))]
    quals = qs[rng.integers(0, len(qs), size=(n, L))]
    return b"".join(b"@SRR123456.%d %d/1\n%s\n+\n%s\n" % (i, i, seqs[i].tobytes(), quals[i].tobytes()) for i in range(n))


def _stats(line):
    assert line.startswith("OK "), line
    f = line.split()
    return int(f[1]), {k: int(v) for k, v in (x.split("=") for x in f[2:])}


@pytest.mark.parametrize("level", [1, 6, 9])
def test_every_level_and_chunk_size(check, fastq, tmp_path, level):
    p = str(tmp_path / "x.fq.gz")
    open(p, "wb").write(gzip.compress(fastq, level))
    for threads, chunk, unit in ((1, 1 << 21, 1 << 20), (4, 4096, 1 << 20), (3, 100_000, 777), (8, 30_000, 1 << 22)):
        n, st = _stats(check(p, threads, chunk, unit))
        assert n == len(fastq)
        if chunk < 1 << 20:
            assert st["chunks"] > 5 and st["markers"] > 0   # chunks did start in the middle of the stream


def test_members_flushes_fixed_and_stored_blocks(check, fastq, tmp_path):
    cut = [0, len(fastq) // 3 + 11, len(fastq) // 3 + 500, 2 * len(fastq) // 3, len(fastq)]
    files = {"multi": b"".join(gzip.compress(fastq[a:b], 6) for a, b in zip(cut, cut[1:]))}
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, len(fastq), 300_000):
        parts += [co.compress(fastq[i:i + 300_000]), co.flush(zlib.Z_FULL_FLUSH if (i // 300_000) % 2 else zlib.Z_SYNC_FLUSH)]
    files["flush"] = b"".join(parts) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    files["fixed"] = co.compress(fastq[:2_000_000]) + co.flush()
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    files["stored"] = co.compress(fastq[:2_000_000]) + co.flush()
    # a member of stored blocks between two ordinary ones; header with a file name and a comment
    hdr = b"\x1f\x8b\x08\x18\0\0\0\0\0\x03" + b"name.fq\0" + b"a comment\0"
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(fastq[:500_000]) + raw.flush()
    named = hdr + body + zlib.crc32(fastq[:500_000]).to_bytes(4, "little") + (500_000).to_bytes(4, "little")
    files["named_then_stored_then_plain"] = named + files["stored"] + gzip.compress(fastq[2_000_000:3_000_000], 9)
    files["garbage_behind"] = gzip.compress(fastq[:1_000_000]) + b"\0" * 64 + b"not gzip"
    for name, blob in files.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        for chunk in (8192, 150_000, 1 << 21):
            n, st = _stats(check(p, 4, chunk))
            if name == "multi":
                assert n == len(fastq) and st["members"] == 4
            if name == "named_then_stored_then_plain":
                assert n == 500_000 + 2_000_000 + 1_000_000 and st["members"] == 3


def test_binary_content_is_still_correct(check, tmp_path):
    # the block finder only accepts blocks that decode to text: with binary data no chunk but the first finds a start, and the
    # whole stream is decoded from the front -- slower, never wrong
    rng = np.random.default_rng(5)
    blob = bytes(rng.integers(0, 256, size=300_000, dtype=np.uint8)) + bytes(rng.integers(0, 4, size=3_000_000, dtype=np.uint8))
    p = str(tmp_path / "bin.gz")
    open(p, "wb").write(gzip.compress(blob, 6))
    n, st = _stats(check(p, 4, 50_000))
    assert n == len(blob)


def test_damaged_streams_are_reported(check, fastq, tmp_path):
    z = gzip.compress(fastq, 6)
    cases = {"truncated": z[: len(z) // 2], "crc": z[:-6] + bytes([z[-6] ^ 0xFF]) + z[-5:], "length": z[:-2] + bytes([z[-2] ^ 1]) + z[-1:]}
    flipped = bytearray(z)
    flipped[len(z) // 2] ^= 0x10
    cases["flipped_bit"] = bytes(flipped)
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        for chunk in (20_000, 1 << 21):
            out = check(p, 4, chunk)
            assert out.startswith("ERROR ") or out.startswith("DIFF "), (name, out)   # never "OK": the stream is not what its trailer says
    assert check(str(tmp_path / "truncated.gz"), 2, 20_000).startswith("ERROR ")
    p = str(tmp_path / "plain.txt")
    open(p, "wb").write(fastq[:100_000])
    assert check(p, 2, 20_000) == "NOTGZIP"


def test_data_that_inflates_a_hundredfold(check, tmp_path):
    # one chunk of compressed bytes may hold hundreds of MB of text: a decode keeps at most 96 M symbols and hands over at a
    # block boundary; the stitcher continues from there (also behind the file's last chunk)
    rec = b"@r\n" + b"ACGT" * 37 + b"AC\n+\n" + b"I" * 150 + b"\n"
    data = rec * 420_000
    p = str(tmp_path / "const.fq.gz")
    open(p, "wb").write(gzip.compress(data, 6))
    assert os.path.getsize(p) < 2_000_000
    for chunk in (1 << 21, 200_000):
        n, st = _stats(check(p, 3, chunk, 1 << 22))
        assert n == len(data)
    n, st = _stats(check(p, 3, 1 << 21, 1 << 22))
    assert st["redone"] >= 1     # the continuation behind the only chunk
