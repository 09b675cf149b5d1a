"""FASTQ text tokenised on the device (gn_stream_upload_fastq): the records it takes are exactly those the host's slab parser
takes (ParallelFastq::Impl::parse -- restated here line by line as `expected_records`), it stops where that parser stops, and
what is classified afterwards equals the same reads uploaded as packed bases."""
import numpy as np
import pytest

import ganon_fixtures as gf
import gpu_util as gu

pytestmark = pytest.mark.gpu

LEGAL = set(b"ACGTURYSWKMBDHVNacgturyswkmbdhvn")


@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return ganon_amd


def expected_records(text: bytes, max_reads: int = 1 << 30):
    """the slab parser's rule (host/seq_io.cpp ParallelFastq::Impl::parse + RangeLines): -> (rec_at, seq_at, seq_len, parsed_bytes)"""
    rec, seq, ln = [], [], []
    pos, n = 0, len(text)

    def line(p):  # -> (begin, end without '\r', next) or None when no '\n' follows
        e = text.find(b"\n", p)
        if e < 0:
            return None
        end = e - 1 if e > p and text[e - 1] == 0x0D else e
        return p, end, e + 1

    while pos < n and len(rec) < max_reads:
        l0 = line(pos)
        if not l0 or l0[1] == l0[0] or text[l0[0]] != ord("@"):
            break
        l1 = line(l0[2])
        if not l1:
            break
        letters = text[l1[0]:l1[1]]
        if any(c not in LEGAL for c in letters):
            break
        l2 = line(l1[2])
        if not l2 or l2[1] == l2[0] or text[l2[0]] != ord("+"):
            break
        q = l2[2]
        if q + len(letters) >= n:  # RangeLines::skip_exact_line: the quality characters and their '\n' must be there
            break
        if text[q + len(letters)] != 0x0A or b"\n" in text[q:q + len(letters)]:
            break
        rec.append(pos)
        seq.append(l1[0])
        ln.append(len(letters))
        pos = q + len(letters) + 1
    return np.array(rec, np.uint32), np.array(seq, np.uint32), np.array(ln, np.uint32), pos


def fastq(records, eol=b"\n"):
    return b"".join(b"@" + i + eol + s + eol + b"+" + p + eol + q + eol for i, s, p, q in records)


def random_records(rng, n, lo=0, hi=300, alphabet=b"ACGT"):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = gu.random_seq(rng, L, alphabet) if L else b""
        q = bytes(rng.integers(33, 75, size=L, dtype=np.uint8))  # includes '@' (64) and '+' (43) as quality characters
        ident = b"read%d/%d some text" % (i, int(rng.integers(0, 10**6)))
        out.append((ident, s, ident if i % 7 == 0 else b"", q))
    return out


def check(hip, flt, text: bytes, max_reads=None, note=""):
    exp = expected_records(text, max_reads if max_reads else 1 << 30)
    st = hip.HipStream(flt, max_reads if max_reads else max(len(exp[0]) + 8, 16), max(len(text) + 64, 256))
    n, nb, parsed = st.upload_fastq(text)
    assert n == len(exp[0]), (note, n, len(exp[0]))
    assert parsed == exp[3], (note, parsed, exp[3])
    assert nb == int(exp[2].sum()), note
    rec_at, seq_at, seq_len = st.fastq_records()
    assert np.array_equal(rec_at, exp[0]) and np.array_equal(seq_at, exp[1]) and np.array_equal(seq_len, exp[2]), note
    return st, exp


def test_regular_text_and_what_follows_equals_the_packed_upload(hip):
    rng = np.random.default_rng(5)
    ibf = gf.random_ibf(64, 4099, 3, 0.4, 2)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    recs = random_records(rng, 3000, 0, 400, b"ACGTNacgtnRYKM")
    text = fastq(recs)
    st, exp = check(hip, flt, text)
    assert len(exp[0]) == 3000 and exp[3] == len(text)
    st.classify(19, 31, 0.1)
    nh, status, mo, m = st.fetch()
    ho, hs = st.fetch_hashes()
    seqs = [r[1] for r in recs]
    bases, off1, _ = gu.pack_reads(seqs, None)
    s2 = hip.HipStream(flt, len(seqs), bases.size)
    s2.submit(bases, off1, None, 19, 31, 0.1)
    nh2, status2, mo2, m2 = s2.fetch()
    ho2, hs2 = s2.fetch_hashes()
    assert np.array_equal(nh, nh2) and np.array_equal(status, status2) and np.array_equal(mo, mo2) and np.array_equal(m, m2)
    assert np.array_equal(ho, ho2) and np.array_equal(hs, hs2)
    assert len(m) > 0
    # the stream takes packed batches and text batches in turn
    s2.upload_fastq(text[:exp[0][1000]])
    s2.classify(19, 31, 0.1)
    nh3, _, mo3, m3 = s2.fetch()
    assert np.array_equal(nh3, nh[:1000]) and np.array_equal(mo3, mo[:1001]) and np.array_equal(m3, m[:int(mo[1000])])
    s2.submit(bases, off1, None, 19, 31, 0.1)
    assert np.array_equal(s2.fetch()[3], m)
    # fewer records than were found
    st.upload_fastq(text)
    st.fastq_keep(17)
    st.classify(19, 31, 0.1)
    nh4, _, mo4, m4 = st.fetch()
    assert len(nh4) == 17 and np.array_equal(mo4, mo[:18]) and np.array_equal(m4, m[:int(mo[17])])
    flt.free()


@pytest.mark.parametrize("case", ["crlf", "no_final_newline", "truncated", "illegal_letter", "wrapped", "blank_line", "plus_missing", "quality_short",
                                  "quality_long", "at_missing", "empty_text", "only_newlines", "first_record_bad", "empty_sequences",
                                  "cr_in_letters_only", "text_is_one_line", "lone_at", "trailing_newlines"])
def test_stops_where_the_slab_parser_stops(hip, case):
    rng = np.random.default_rng(11)
    flt, _ = (hip.HipFilter.ibf(*(lambda i: (i.data, i.bins, i.bin_size, i.hash_funs))(gf.random_ibf(64, 257, 3, 0.3, 1))), None)
    recs = random_records(rng, 200, 1, 200)
    good = fastq(recs)
    cut = len(fastq(recs[:120]))
    if case == "crlf":
        text = fastq(recs, b"\r\n")
    elif case == "no_final_newline":
        text = good[:-1]
    elif case == "truncated":
        text = good[:cut + 37]
    elif case == "illegal_letter":
        i, s, p, q = recs[120]
        text = fastq(recs[:120]) + fastq([(i, s[:3] + b"X" + s[3:], p, q + b"I")]) + fastq(recs[121:])
    elif case == "wrapped":
        i, s, p, q = recs[120]
        s = s + b"ACGTACGT"
        text = fastq(recs[:120]) + b"@" + i + b"\n" + s[:4] + b"\n" + s[4:] + b"\n+\n" + b"I" * len(s) + b"\n" + fastq(recs[121:])
    elif case == "blank_line":
        text = fastq(recs[:120]) + b"\n" + fastq(recs[120:])
    elif case == "plus_missing":
        i, s, p, q = recs[120]
        text = fastq(recs[:120]) + b"@" + i + b"\n" + s + b"\n-\n" + q + b"\n" + fastq(recs[121:])
    elif case == "quality_short":
        i, s, p, q = recs[120]
        text = fastq(recs[:120]) + fastq([(i, s + b"AC", p, q + b"I")]) + fastq(recs[121:])
    elif case == "quality_long":
        i, s, p, q = recs[120]
        text = fastq(recs[:120]) + fastq([(i, s, p, q + b"II")]) + fastq(recs[121:])
    elif case == "at_missing":
        text = fastq(recs[:120]) + fastq(recs[120:])[1:]
    elif case == "empty_text":
        text = b""
    elif case == "only_newlines":
        text = b"\n" * 4097
    elif case == "first_record_bad":
        text = b">fasta\nACGT\n" + good
    elif case == "empty_sequences":
        text = fastq([(b"a", b"", b"", b""), (b"b", b"ACGT", b"", b"IIII"), (b"c", b"", b"c", b"")] * 50)
    elif case == "cr_in_letters_only":
        # '\r' ends the letters line only: the quality line has one character fewer than the raw line, as many as there are letters
        text = b"".join(b"@" + i + b"\n" + s + b"\r\n+\n" + q + b"\n" for i, s, p, q in recs)
    elif case == "text_is_one_line":
        text = b"@" + b"A" * 10000
    elif case == "trailing_newlines":
        # more groups of four lines than records of six bytes fit the text: the per-record arrays beyond that bound are never
        # written, and nothing may be read from them (the stream is large enough for every group to count as a record)
        text = fastq(recs[:50]) + b"\n" * 300_000
    else:
        text = fastq(recs[:5]) + b"@\n"
    st, exp = check(hip, flt, text, max_reads=200_000 if case == "trailing_newlines" else None, note=case)
    if case == "trailing_newlines":
        assert len(exp[0]) == 50 and exp[3] == len(fastq(recs[:50]))
    if case in ("truncated", "illegal_letter", "wrapped", "blank_line", "plus_missing", "quality_short", "quality_long", "at_missing"):
        assert len(exp[0]) == 120 and exp[3] == cut
    if case in ("crlf", "first_record_bad", "empty_text", "only_newlines", "text_is_one_line"):
        assert len(exp[0]) == 0 and exp[3] == 0
    if case in ("empty_sequences", "cr_in_letters_only"):
        assert exp[3] == len(text)
    if case == "no_final_newline":
        assert len(exp[0]) == 199
    st.destroy()
    flt.free()


def test_large_text_tile_borders_and_the_stream_capacity(hip):
    rng = np.random.default_rng(3)
    ibf = gf.random_ibf(64, 257, 3, 0.3, 1)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    # 150-letter records (315 bytes: every alignment against the 4 KiB tiles and the 16-byte lanes occurs) and a mix of lengths
    recs = [(b"r%09d" % i, gu.random_seq(rng, 150), b"", b"I" * 150) for i in range(40000)]
    text = fastq(recs)
    st, exp = check(hip, flt, text)
    assert len(exp[0]) == 40000
    st.destroy()
    recs = random_records(rng, 30000, 0, 700)
    text = fastq(recs)
    st, exp = check(hip, flt, text)
    assert exp[3] == len(text)
    st.destroy()
    # more records than the stream holds: the batch ends at the capacity, the rest is the caller's
    st, exp = check(hip, flt, text, max_reads=1000)
    assert len(exp[0]) == 1000 and exp[3] == len(fastq(recs[:1000]))
    st.destroy()
    # text larger than the stream: refused
    s = hip.HipStream(flt, 16, 1024)
    with pytest.raises(Exception):
        s.upload_fastq(text)
    s.destroy()
    flt.free()


# ---------------------------------------------------------------------------------------------------------------- FASTA
def expected_fasta_records(text: bytes, max_reads: int = 1 << 30):
    """the two-line rule of GN_TEXT_FASTA: >id / letters on one line / then a '>' or the end of the text"""
    rec, seq, ln = [], [], []
    pos, n = 0, len(text)
    while pos < n and len(rec) < max_reads:
        a = text.find(b"\n", pos)
        if a < 0 or a == pos or text[pos] != ord(">"):
            break
        b = text.find(b"\n", a + 1)
        if b < 0:
            break
        end = b - 1 if b > a + 1 and text[b - 1] == 0x0D else b
        letters = text[a + 1:end]
        if b > a + 1 and text[a + 1] in b">;":
            break
        if any(c not in LEGAL for c in letters):
            break
        if b + 1 < n and text[b + 1] != ord(">"):
            break
        rec.append(pos)
        seq.append(a + 1)
        ln.append(len(letters))
        pos = b + 1
    return np.array(rec, dtype=np.uint32), np.array(seq, dtype=np.uint32), np.array(ln, dtype=np.uint32), pos


def fasta(records, eol=b"\n"):
    return b"".join(b">" + i + eol + s + eol for i, s in records)


@pytest.mark.parametrize("case", ["plain", "crlf_ids_only", "crlf", "wrapped_in_the_middle", "blank_line", "semicolon", "illegal_letter", "no_final_newline",
                                  "empty_sequences", "header_only_at_end", "first_line_no_header", "digits_and_spaces", "empty_text", "capacity"])
def test_fasta_text_two_line_records(hip, case):
    rng = np.random.default_rng(17)
    ibf = gf.random_ibf(64, 4099, 3, 0.4, 2)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    recs = [(b"read%d some words" % i, gu.random_seq(rng, int(rng.integers(1, 300)), b"ACGTNacgtRYKM")) for i in range(2500)]
    good = fasta(recs)
    cut = len(fasta(recs[:1200]))
    max_reads = None
    if case == "plain":
        text = good
    elif case == "crlf":
        text = fasta(recs, b"\r\n")
    elif case == "crlf_ids_only":
        text = b"".join(b">" + i + b"\r\n" + s + b"\n" for i, s in recs)
    elif case == "wrapped_in_the_middle":
        i, s = recs[1200]
        s = s + b"ACGTACGTAC"
        text = fasta(recs[:1200]) + b">" + i + b"\n" + s[:5] + b"\n" + s[5:] + b"\n" + fasta(recs[1201:])
    elif case == "blank_line":
        text = fasta(recs[:1201]) + b"\n" + fasta(recs[1201:])   # the record BEFORE the blank line is not followed by '>': not taken
    elif case == "semicolon":
        text = fasta(recs[:1201]) + b";comment\nACGT\n" + fasta(recs[1201:])
    elif case == "illegal_letter":
        i, s = recs[1200]
        text = fasta(recs[:1200]) + fasta([(i, s[:1] + b"!" + s[1:])]) + fasta(recs[1201:])
    elif case == "no_final_newline":
        text = good[:-1]
    elif case == "empty_sequences":
        text = fasta([(b"a", b""), (b"b", b"ACGT"), (b"c", b"")] * 100)
    elif case == "header_only_at_end":
        text = fasta(recs[:10]) + b">lonely\n"
    elif case == "first_line_no_header":
        text = b"ACGT\n" + good
    elif case == "digits_and_spaces":
        i, s = recs[1200]
        text = fasta(recs[:1200]) + b">" + i + b"\n" + s[:4] + b" 10 " + s[4:] + b"\n" + fasta(recs[1201:])
    elif case == "empty_text":
        text = b""
    else:
        text, max_reads = good, 700
    exp = expected_fasta_records(text, max_reads if max_reads else 1 << 30)
    st = hip.HipStream(flt, max_reads if max_reads else max(len(exp[0]) + 8, 16), max(len(text) + 64, 256))
    n, nb, parsed = st.upload_fastq(text, fasta=True)
    assert (n, parsed, nb) == (len(exp[0]), exp[3], int(exp[2].sum())), case
    rec_at, seq_at, seq_len = st.fastq_records()
    assert np.array_equal(rec_at, exp[0]) and np.array_equal(seq_at, exp[1]) and np.array_equal(seq_len, exp[2]), case
    if case in ("wrapped_in_the_middle", "semicolon", "illegal_letter", "digits_and_spaces"):
        assert n == 1200 and parsed == cut
    if case == "blank_line":
        assert n == 1200   # (record 1200 is followed by a blank line, not by a header)
    if case in ("plain", "crlf", "crlf_ids_only", "empty_sequences"):
        assert parsed == len(text)
    if case == "no_final_newline":
        assert n == 2499
    if case in ("first_line_no_header", "empty_text"):
        assert n == 0 and parsed == 0
    if case == "header_only_at_end":
        assert n == 10
    if case == "plain":   # what is classified afterwards equals the packed upload of the same reads
        st.classify(19, 31, 0.1)
        nh, status, mo, m = st.fetch()
        bases, off1, _ = gu.pack_reads([s for _, s in recs], None)
        s2 = hip.HipStream(flt, len(recs), bases.size)
        s2.submit(bases, off1, None, 19, 31, 0.1)
        nh2, status2, mo2, m2 = s2.fetch()
        assert np.array_equal(nh, nh2) and np.array_equal(status, status2) and np.array_equal(mo, mo2) and np.array_equal(m, m2) and len(m) > 0
        s2.destroy()
    st.destroy()
    flt.free()


# ---------------------------------------------------------------------------------------------------------------- pairs
@pytest.mark.parametrize("case", ["plain", "bad_mate", "bad_first", "file2_shorter", "file2_longer", "empty", "fasta"])
def test_text_pair_is_the_packed_paired_upload(hip, case):
    rng = np.random.default_rng(23)
    ibf = gf.random_ibf(64, 4099, 3, 0.4, 2)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    n = 3000
    r1 = random_records(rng, n, 0, 300, b"ACGTNacgtn")
    r2 = [(b"mate%d/2 with an id of another length" % i, gu.random_seq(rng, int(rng.integers(0, 300)), b"ACGT"), b"", None) for i in range(n)]
    r2 = [(i, s, p, b"J" * len(s)) for i, s, p, _ in r2]
    cut = n
    if case == "bad_mate":      # mate 1700 has a quality line one character short: the batch is the 1700 pairs before it
        i, s, p, q = r2[1700]
        r2[1700] = (i, s + b"A", p, q)
        cut = 1700
    if case == "bad_first":
        i, s, p, q = r1[900]
        r1[900] = (i, s[:2] + b"!" + s[2:], p, q + b"I")
        cut = 900
    if case == "fasta":
        t1, t2 = fasta([(i, s) for i, s, _, _ in r1]), fasta([(i, s) for i, s, _, _ in r2])
        sizes1 = np.cumsum([0] + [len(fasta([(i, s)])) for i, s, _, _ in r1])
        sizes2 = np.cumsum([0] + [len(fasta([(i, s)])) for i, s, _, _ in r2])
    else:
        t1, t2 = fastq(r1), fastq(r2)
        sizes1 = np.cumsum([0] + [len(fastq([r])) for r in r1])
        sizes2 = np.cumsum([0] + [len(fastq([r])) for r in r2])
    if case == "file2_shorter":
        t2, cut = t2[: int(sizes2[2000])], 2000
    if case == "file2_longer":
        t1, cut = t1[: int(sizes1[2500])], 2500
    if case == "empty":
        t1, t2, cut = b"", b"", 0
    st = hip.HipStream(flt, n + 8, len(t1) + len(t2) + 256)
    got, p1, p2 = st.upload_text_pair(t1, t2, fasta=case == "fasta")
    assert (got, p1, p2) == (cut, int(sizes1[cut]), int(sizes2[cut])), case
    if cut:
        rec1, seq1, len1 = st.fastq_records()
        rec2, seq2, len2 = st.text_pair_records2()
        assert np.array_equal(rec1, sizes1[:cut]) and np.array_equal(rec2, sizes2[:cut])
        assert np.array_equal(len1, [len(r[1]) for r in r1[:cut]]) and np.array_equal(len2, [len(r[1]) for r in r2[:cut]])
        assert all(t1[a:a + l] == r[1] for a, l, r in zip(seq1[:50], len1[:50], r1)) and all(t2[a:a + l] == r[1] for a, l, r in zip(seq2[:50], len2[:50], r2))
        st.classify(19, 31, 0.1)
        nh, status, mo, m = st.fetch()
        ho, hs = st.fetch_hashes()
        bases, off1, off2 = gu.pack_reads([r[1] for r in r1[:cut]], [r[1] for r in r2[:cut]])
        s2 = hip.HipStream(flt, cut, bases.size)
        s2.submit(bases, off1, off2, 19, 31, 0.1)
        nh2, status2, mo2, m2 = s2.fetch()
        ho2, hs2 = s2.fetch_hashes()
        assert np.array_equal(nh, nh2) and np.array_equal(status, status2) and np.array_equal(mo, mo2) and np.array_equal(m, m2)
        assert np.array_equal(ho, ho2) and np.array_equal(hs, hs2) and len(m) > 0
        # a single text on the same stream afterwards, and a pair again
        st.upload_fastq(t1[: int(sizes1[min(cut, 500)])], fasta=case == "fasta")
        st.classify(19, 31, 0.1)
        assert len(st.fetch()[0]) == min(cut, 500)
        assert st.upload_text_pair(t1, t2, fasta=case == "fasta")[0] == cut
        s2.destroy()
    st.destroy()
    flt.free()
