"""gn_inflate_* (csrc/gn_inflate.hip): a gzip file inflated on the device must be zlib's bytes.

The checker is zlib itself (what the reference reads `.gz` input through: seqan3 over zlib, GanonClassify.cpp:1220-1287)."""
import gzip
import io
import zlib

import numpy as np
import pytest

from ganon_amd import hip

pytestmark = pytest.mark.gpu


def _fastq(n, seed=3, read_len=100):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=300_000, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n):
        p = int(rng.integers(0, genome.size - read_len))
        b = lut[genome[p:p + read_len]].copy()
        e = rng.random(read_len) < 0.01
        b[e] = lut[rng.integers(0, 4, size=int(e.sum()))]
        q = rng.choice(np.frombuffer(b"FFFFFFF:,#", dtype=np.uint8), size=read_len)
        out.append(b"@M01:%d:000000000-ABCDE:1:%d:%d:%d 1:N:0:7\n" % (seed, 1101 + i % 900, int(rng.integers(1000, 30000)), int(rng.integers(1000, 30000))))
        out.append(b.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


def _gz(text, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, memlevel, strategy)
    return co.compress(text) + co.flush()


def _inflate(gz, chunk=0, step=0, feed=0):
    data = np.frombuffer(gz, dtype=np.uint8)
    with hip.HipInflate(data.size, chunk_bytes=chunk, step_bytes=step) as z:
        out = z.inflate_all(data, feed_bytes=feed)
        return out.tobytes(), z.stats()


TEXT = _fastq(20000)


@pytest.mark.parametrize("level", [1, 4, 6, 9])
@pytest.mark.parametrize("chunk", [0, 4096, 1024])
def test_levels_and_chunk_sizes(level, chunk):
    gz = _gz(TEXT, level)
    got, st = _inflate(gz, chunk=chunk)
    assert got == TEXT
    if chunk:
        assert st["chunks"] > 4  # really decoded in parallel pieces


def test_default_sizes_use_markers_and_chunks():
    gz = _gz(TEXT * 8, 6)
    got, st = _inflate(gz)
    assert got == TEXT * 8
    assert st["chunks"] >= len(gz) // 32768 // 2 and st["markers"] > 0


@pytest.mark.parametrize("text", [b"", b"A", b"@r\nACGT\n+\nIIII\n", b"N" * 100000, bytes(range(256)) * 40])
def test_small_and_odd_texts(text):
    got, _ = _inflate(_gz(text))
    assert got == text


def test_binary_data_goes_through_the_fixups_or_is_refused():
    rng = np.random.default_rng(5)
    text = rng.integers(0, 256, size=300_000, dtype=np.uint8).tobytes() + TEXT[:200_000]
    gz = _gz(text)
    try:
        got, _ = _inflate(gz, chunk=4096)
    except hip.GanonHipError as e:
        assert e.code == -34
    else:
        assert got == text


@pytest.mark.parametrize("strategy", [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
def test_strategies(strategy):
    text = TEXT[:400_000]
    gz = _gz(text, 6, strategy)
    try:
        got, _ = _inflate(gz, chunk=8192)
    except hip.GanonHipError as e:  # (a file of fixed blocks only has nothing for the search: refused, not wrong)
        assert strategy == zlib.Z_FIXED and e.code == -34
    else:
        assert got == text


def test_stored_blocks():
    text = TEXT[:300_000]
    gz = _gz(text, 0)
    try:
        got, _ = _inflate(gz, chunk=65536)
    except hip.GanonHipError as e:
        assert e.code == -34
    else:
        assert got == text


def test_flush_points_and_mixed_blocks():
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts, text = [], []
    for i in range(40):
        t = TEXT[i * 50_000:(i + 1) * 50_000]
        text.append(t)
        parts.append(co.compress(t))
        parts.append(co.flush(zlib.Z_FULL_FLUSH if i % 3 == 0 else zlib.Z_SYNC_FLUSH if i % 3 == 1 else zlib.Z_NO_FLUSH))
    parts.append(co.flush())
    got, _ = _inflate(b"".join(parts), chunk=4096)
    assert got == b"".join(text)


def test_members_with_header_fields_and_trailing_garbage():
    a, b, c = TEXT[:500_000], TEXT[500_000:1_200_000], b""
    bio = io.BytesIO()
    with gzip.GzipFile(filename="reads_with_a_name.fq", mode="wb", fileobj=bio, compresslevel=6, mtime=0) as f:
        f.write(a)
    m1 = bio.getvalue()
    # FEXTRA + FCOMMENT + FHCRC by hand around a raw deflate body
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    body = co.compress(b) + co.flush()
    hdr = bytes([0x1F, 0x8B, 8, 4 | 16 | 2, 0, 0, 0, 0, 0, 255]) + (5).to_bytes(2, "little") + b"extra" + b"a comment\0"
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    m2 = hdr + body + zlib.crc32(b).to_bytes(4, "little") + (len(b) & 0xFFFFFFFF).to_bytes(4, "little")
    m3 = _gz(c)
    gz = m1 + m2 + m3 + b"\0\0\0garbage that is no member"
    assert zlib.decompressobj(31).decompress(m1) == a
    got, st = _inflate(gz, chunk=8192)
    assert got == a + b + c
    assert st["members"] == 3


def test_many_steps_and_partial_feeds():
    gz = _gz(TEXT * 4, 6)
    got, st = _inflate(gz, chunk=2048, step=65536, feed=50_000)
    assert got == TEXT * 4
    assert st["steps"] > 4


def test_truncated_and_damaged():
    gz = _gz(TEXT, 6)
    with pytest.raises(hip.GanonHipError):
        _inflate(gz[:len(gz) // 2], chunk=4096)
    bad = bytearray(gz)
    for i in range(len(gz) // 3, len(gz) // 3 + 64):
        bad[i] ^= 0x5A
    try:
        got, _ = _inflate(bytes(bad), chunk=4096)
    except hip.GanonHipError:
        pass
    else:  # (whatever decodes must not be passed off as the file's text when zlib refuses it)
        with pytest.raises(zlib.error):
            zlib.decompress(bytes(bad), 31)
        pytest.fail("damaged stream was accepted")


def test_wrong_crc_or_length_in_a_trailer_is_refused():
    gz = bytearray(_gz(TEXT, 6))
    for at in (len(gz) - 8, len(gz) - 4):  # CRC-32, then ISIZE
        bad = bytearray(gz)
        bad[at] ^= 1
        with pytest.raises(hip.GanonHipError) as e:
            _inflate(bytes(bad), chunk=4096)
        assert e.value.code == -34
    # ... in the first of two members, with the steps cutting the members anywhere
    two = bytearray(_gz(TEXT[:700_001], 6) + _gz(TEXT[700_001:], 9))
    assert _inflate(bytes(two), chunk=2048, step=65536, feed=30_000)[0] == TEXT
    first_len = len(_gz(TEXT[:700_001], 6))
    two[first_len - 7] ^= 0x80
    with pytest.raises(hip.GanonHipError):
        _inflate(bytes(two), chunk=2048, step=65536, feed=30_000)


def test_a_flipped_bit_that_still_decodes_is_caught_by_the_crc():
    # stored blocks: a flipped data bit changes no code, no length -- only the CRC notices
    text = TEXT[:200_000]
    gz = bytearray(_gz(text, 0))
    gz[len(gz) // 2] ^= 0x04
    assert zlib.decompressobj(31).decompress(bytes(gz[:-8]))  # (the deflate data itself is fine)
    with pytest.raises(hip.GanonHipError) as e:
        _inflate(bytes(gz), chunk=65536)
    assert e.value.code == -34


def test_cuts_carry_and_batches_from_device_text():
    """the C ABI of the compressed-input path: gn_inflate_step -> gn_inflate_cuts -> gn_stream_upload_text_device ->
    gn_stream_fastq_headers must yield exactly the records of the text, step after step, with the record a step's end cuts carried over"""
    import ganon_fixtures as gf
    recs = [TEXT[i:j] for i, j in zip(*(lambda nl: ([0] + [nl[k] + 1 for k in range(3, len(nl) - 1, 4)], [nl[k] + 1 for k in range(3, len(nl), 4)]))(
        [i for i, b in enumerate(TEXT) if b == 10]))]
    assert b"".join(recs) == TEXT
    gz = np.frombuffer(_gz(TEXT, 6), dtype=np.uint8)
    ibf = gf.random_ibf(64, 257, 3, 0.3, 1)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    with hip.HipInflate(gz.size, chunk_bytes=4096, step_bytes=262144) as z:
        z.feed(gz)
        seen, done, steps = [], False, 0
        stream = None
        while not done:
            n, done = z.step()
            steps += 1
            if n == 0:
                continue
            cuts = z.cuts(4, 100_000)
            text = z.text(n)
            assert cuts.size and all(text[int(c) - 1] == 10 for c in cuts) and list(cuts) == sorted(set(int(c) for c in cuts))
            # every cut is a record boundary: the number of newlines before it is a multiple of four
            nl_before = np.cumsum(text == 10)
            assert all(int(nl_before[int(c) - 1]) % 4 == 0 for c in cuts)
            last = int(cuts[-1])
            assert last == (int(np.nonzero(text == 10)[0][(int(nl_before[-1]) // 4) * 4 - 1]) + 1 if nl_before[-1] >= 4 else 0)
            ptr, nb = z.text_device()
            assert nb == n
            if flt is not None:
                if stream is None:
                    stream = hip.HipStream(flt, 200_000, 4 << 20)
                a = 0
                for c in cuts:
                    c = int(c)
                    k, _, parsed = stream.upload_text_device(ptr + a, c - a)
                    assert parsed == c - a
                    hdr, off = stream.fastq_headers()
                    for i in range(k):
                        seen.append(hdr[off[i]:off[i + 1]])
                    a = c
            if not done:
                z.set_carry(n - last)
            else:
                assert last == n
        assert steps > 2
        if flt is not None:
            assert seen == [r[:r.index(b"\n") + 1] for r in recs]


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_compressor_settings_against_zlib(seed):
    """random compressor settings (level, strategy, window, memLevel: block sizes from a few hundred symbols to 64 Ki), random flush
    points, random mixtures of FASTQ text, long runs and bytes -- the device's text is zlib's, or the file is refused (GN_ERANGE), never wrong"""
    rng = np.random.default_rng(1000 + seed)
    parts = []
    for _ in range(int(rng.integers(3, 9))):
        kind = int(rng.integers(0, 4))
        n = int(rng.integers(1_000, 400_000))
        if kind == 0:
            a = int(rng.integers(0, len(TEXT) - n))
            parts.append(TEXT[a:a + n])
        elif kind == 1:
            parts.append(bytes([int(rng.integers(65, 91))]) * n)
        elif kind == 2:
            parts.append(rng.integers(32, 127, size=n, dtype=np.uint8).tobytes())
        else:
            parts.append((b"@id%d\nACGTNNNN\n+\nIIIIIIII\n" % seed) * (n // 28))
    level = int(rng.integers(1, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_DEFAULT_STRATEGY][int(rng.integers(0, 4))]
    wbits = 16 + int(rng.integers(9, 16))
    mem = int(rng.integers(1, 10))
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    gz = b""
    for t in parts:
        gz += co.compress(t)
        f = int(rng.integers(0, 4))
        if f == 1:
            gz += co.flush(zlib.Z_SYNC_FLUSH)
        elif f == 2:
            gz += co.flush(zlib.Z_FULL_FLUSH)
    gz += co.flush()
    text = b"".join(parts)
    assert zlib.decompress(gz, 31) == text
    chunk = [0, 1024, 4096, 16384][int(rng.integers(0, 4))]
    step = [0, 65536, 1 << 20][int(rng.integers(0, 3))]
    try:
        got, _ = _inflate(gz, chunk=chunk, step=step, feed=[0, 70_000][int(rng.integers(0, 2))])
    except hip.GanonHipError as e:
        assert e.code == -34, e
    else:
        assert got == text


def _bgzf(text, block=65280, level=6):
    """blocked gzip as bgzip writes it: members of at most 64 KiB with the BC extra field, an empty member at the end"""
    out = []
    for a in list(range(0, len(text), block)) + [None]:
        piece = b"" if a is None else text[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(bytes([0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 66, 67, 2, 0]) + bsize.to_bytes(2, "little") + body
                   + zlib.crc32(piece).to_bytes(4, "little") + (len(piece) & 0xFFFFFFFF).to_bytes(4, "little"))
    return b"".join(out)


@pytest.mark.parametrize("chunk,step", [(0, 0), (4096, 262144), (65536, 0)])
def test_blocked_gzip_members_are_chunk_starts(chunk, step):
    text = TEXT * 3
    gz = _bgzf(text)
    assert zlib.decompressobj(31).decompress(gz[:gz.index(b"\x1f\x8b\x08\x04", 10)]) == text[:65280]
    got, st = _inflate(gz, chunk=chunk, step=step)
    assert got == text
    assert st["members"] == len(text) // 65280 + 2 and st["markers"] == 0  # every chunk began at a member: nothing refers across
    assert st["chunks"] > 8


def test_many_small_members_and_a_wrong_crc_among_them():
    parts = [TEXT[i:i + 7000] for i in range(0, 700_000, 7000)]
    gz = b"".join(gzip.compress(t, 6) for t in parts)
    got, st = _inflate(gz, chunk=4096)
    assert got == b"".join(parts) and st["members"] == len(parts)
    bad = bytearray(gz)
    at = len(b"".join(gzip.compress(t, 6) for t in parts[:37])) - 8  # the CRC-32 of member 37
    bad[at] ^= 0x10
    with pytest.raises(hip.GanonHipError):
        _inflate(bytes(bad), chunk=4096)


def _inflate_in_turns(gz, n_turns, chunk, step, lines_per_record=0, piece=100_000):
    """N inflaters of the same file take its steps in turn (gn_inflate_set_turns / gn_inflate_handoff) -- here all on the one GPU of the
    box; on a node each sits on its own device and the text of step k lies where worker k mod N classifies it.  lines_per_record != 0:
    the caller cuts at record boundaries and carries the record a step's end cuts, as devgzip.cpp does."""
    data = np.frombuffer(gz, dtype=np.uint8)
    zs = [hip.HipInflate(data.size, chunk_bytes=chunk, step_bytes=step) for _ in range(n_turns)]
    try:
        for i, z in enumerate(zs):
            z.set_turns(n_turns, i)
            z.feed(data)
        out, k, done, steps_of = [], 0, False, [0] * n_turns
        while not done:
            z = zs[k % n_turns]
            n, done = z.step()
            steps_of[k % n_turns] += 1
            text = z.text(n) if n else np.zeros(0, np.uint8)
            if lines_per_record and not done:
                cuts = z.cuts(lines_per_record, piece) if n else np.zeros(0, np.uint64)
                last = int(cuts[-1]) if cuts.size else 0
                out.append(text[:last].tobytes())
                z.set_carry(n - last)
            else:
                out.append(text.tobytes())
            if not done:
                z.handoff(zs[(k + 1) % n_turns])
            k += 1
        return b"".join(out), steps_of, [z.stats() for z in zs]
    finally:
        for z in zs:
            z.close()


@pytest.mark.parametrize("n_turns", [2, 3])
@pytest.mark.parametrize("level,chunk,step,lpr", [(6, 4096, 262144, 0), (6, 4096, 262144, 4), (1, 1024, 65536, 4), (9, 0, 1 << 20, 4)])
def test_steps_taken_in_turns_by_several_inflaters_give_zlibs_bytes(n_turns, level, chunk, step, lpr):
    text = TEXT * 6
    gz = _gz(text, level)
    got, steps_of, stats = _inflate_in_turns(gz, n_turns, chunk, step, lpr)
    assert got == text
    assert min(steps_of) >= 2 and max(steps_of) - min(steps_of) <= 1   # really in turns
    assert sum(s["text_bytes"] for s in stats) == len(text) and all(s["text_bytes"] > 0 for s in stats)
    assert sum(s["markers"] for s in stats) > 0                         # back-references crossed chunks and steps (one member)


def test_turns_with_many_members_blocked_gzip_and_a_crc_that_crosses_steps():
    parts = [TEXT[i:i + 50_000] for i in range(0, len(TEXT), 50_000)]
    gz = b"".join(gzip.compress(t, 6) for t in parts) * 2          # members end inside steps: the open member's CRC is handed on
    got, steps_of, stats = _inflate_in_turns(gz, 2, 4096, 131072, 4)
    assert got == TEXT * 2 and sum(s["members"] for s in stats) == 2 * len(parts)
    got, _, stats = _inflate_in_turns(_bgzf(TEXT * 3), 3, 4096, 262144)
    assert got == TEXT * 3 and sum(s["markers"] for s in stats) == 0
    bad = bytearray(gz)
    bad[len(gz) - 8] ^= 0x01                                           # the last member's CRC-32
    with pytest.raises(hip.GanonHipError):
        _inflate_in_turns(bytes(bad), 2, 4096, 131072, 4)


def test_handoff_refuses_inflaters_of_different_files_and_late_turns():
    gz = np.frombuffer(_gz(TEXT, 6), dtype=np.uint8)
    with hip.HipInflate(gz.size, chunk_bytes=4096, step_bytes=262144) as a, hip.HipInflate(gz.size + 8, chunk_bytes=4096, step_bytes=262144) as b:
        with pytest.raises(hip.GanonHipError):
            a.handoff(b)
        with pytest.raises(hip.GanonHipError):
            a.set_turns(2, 2)
        a.feed(gz)
        a.step()
        with pytest.raises(hip.GanonHipError):
            a.set_turns(2, 0)
