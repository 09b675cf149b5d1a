"""Input-format semantics of the host reader (seqan3::sequence_file_input as used at GanonClassify.cpp:1220-1287):
the same reads written as FASTQ / FASTA, wrapped, CRLF, gzipped, without a final newline ... must give byte-identical
outputs; a parse error keeps the records before it and moves on.  CPU: test-only oracle backend."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import cli_util as cu
import ganon_fixtures as gf
from test_cli_kat import _sim_reads, oracle_bin, sim_db  # noqa: F401  (fixtures)


def _wrap(s, n):
    return "\n".join(s[i:i + n] for i in range(0, len(s), n)) if s else ""


def _write(path, text, nl="\n", gz=False):
    data = text.replace("\n", nl).encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def _write_bgzf(path, text, block=20000):
    """blocked gzip as bgzip / Illumina converters write it: independent members with a 'BC' extra field, empty
    end-of-file member"""
    import struct, zlib
    data = text.encode()
    with open(path, "wb") as f:
        for lo in list(range(0, len(data), block)) + [None]:
            chunk = b"" if lo is None else data[lo:lo + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            bsize = 12 + 6 + len(comp) + 8 - 1
            f.write(struct.pack("<4BI2BH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6) + struct.pack("<2BHH", ord("B"), ord("C"), 2, bsize))
            f.write(comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _fastq(recs, wrap=None):
    out = []
    for rid, s in recs:
        q = "I" * len(s)
        out.append(f"@{rid}\n{_wrap(s, wrap) if wrap else s}\n+\n{_wrap(q, wrap) if wrap else q}\n")
    return "".join(out)


def _fasta(recs, wrap=70, lower=False):
    return "".join(f">{rid}\n{_wrap(s.lower() if lower else s, wrap)}\n" for rid, s in recs)


def _run(binary, sim_db, reads_arg, prefix, check=True, cutoff="0.25"):
    args = ["--ibf", sim_db["ibf"], "--tax", sim_db["tax"], "-o", prefix, "--output-all", "--output-lca",
            "--output-unclassified", "--quiet", "--rel-cutoff", cutoff, "--rel-filter", "0.1"] + reads_arg
    return cu.run(binary, args, check=check)


def _outputs(prefix):
    return {e: open(prefix + e, "rb").read() for e in (".rep", ".all", ".one", ".unc")}


def _variants(d, r1):
    v = {}
    _write(os.path.join(d, "a.fq"), _fastq(r1)); v["plain"] = "a.fq"
    _write(os.path.join(d, "b.fastq"), _fastq(r1).rstrip("\n"), nl="\r\n"); v["crlf_noeol"] = "b.fastq"
    _write(os.path.join(d, "c.fq"), "\n\n" + _fastq(r1, wrap=60)); v["wrapped_fastq"] = "c.fq"
    _write(os.path.join(d, "d.fa"), _fasta(r1, 70, lower=True)); v["fasta_lower"] = "d.fa"
    _write(os.path.join(d, "e.fasta.gz"), _fasta(r1, 10_000), gz=True); v["fasta_gz"] = "e.fasta.gz"
    _write_bgzf(os.path.join(d, "g.fq.gz"), _fastq(r1), block=1500); v["fastq_bgzf"] = "g.fq.gz"   # members inflated in parallel
    _write_bgzf(os.path.join(d, "h.fastq.bgzf"), _fastq(r1, wrap=60), block=64000); v["fastq_bgzf_wrapped"] = "h.fastq.bgzf"
    spaced = "".join(f">{rid}\n{_wrap(s, 33).replace(chr(10), ' ' + chr(10))} \n" for rid, s in r1)  # blanks inside sequences
    _write(os.path.join(d, "f.fna"), spaced, nl="\r\n"); v["fasta_crlf_spaces"] = "f.fna"
    import bz2  # bzip2: one stream, and several streams behind each other (what pbzip2 writes)
    fq = _fastq(r1).encode()
    open(os.path.join(d, "i.fq.bz2"), "wb").write(bz2.compress(fq)); v["fastq_bz2"] = "i.fq.bz2"
    cut = [0, len(fq) // 3, len(fq) // 3 + 17, len(fq)]
    open(os.path.join(d, "j.fastq.bz2"), "wb").write(b"".join(bz2.compress(fq[a:b], 1) for a, b in zip(cut, cut[1:])))
    v["fastq_bz2_three_streams"] = "j.fastq.bz2"
    open(os.path.join(d, "k.fa.bz2"), "wb").write(bz2.compress(_fasta(r1, 70).encode())); v["fasta_bz2"] = "k.fa.bz2"
    # padding / garbage behind the last stream: the bzip2 tool warns and ignores it, and so does the reader
    open(os.path.join(d, "l.fq.bz2"), "wb").write(bz2.compress(fq) + b"\0" * 513 + b"not a stream"); v["fastq_bz2_trailing_garbage"] = "l.fq.bz2"
    return v


def _check_variants(binary, sim_db, tmp):
    r1, r2 = _sim_reads()
    r1 = r1[:40]
    v = _variants(tmp, r1)
    ref = None
    for name, fn in v.items():
        p = os.path.join(tmp, "o_" + name)
        _run(binary, sim_db, ["--single-reads", os.path.join(tmp, fn)], p)
        out = _outputs(p)
        assert len(out[".all"]) > 100
        if ref is None:
            ref = out
        assert out == ref, name


def test_formats_equivalent_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_variants(oracle_bin, sim_db, str(tmp_path))


def _check_errors(binary, sim_db, tmp):
    r1, r2 = _sim_reads()
    r1, r2 = r1[:30], r2[:30]
    good1, good2 = os.path.join(tmp, "g1.fq"), os.path.join(tmp, "g2.fq")
    # reference run: first 10 pairs only
    _write(good1, _fastq(r1[:10])); _write(good2, _fastq(r2[:10]))
    p_ref = os.path.join(tmp, "ref")
    _run(binary, sim_db, ["--paired-reads", good1 + "," + good2], p_ref)
    _write(os.path.join(tmp, "full1.fq"), _fastq(r1))
    # (1) illegal letter in record 11 of file 1 -> error reported, 10 records kept
    bad = list(r1)
    bad[10] = (bad[10][0], bad[10][1][:50] + "!" + bad[10][1][51:])
    b1, b2 = os.path.join(tmp, "b1.fq"), os.path.join(tmp, "b2.fq")
    _write(b1, _fastq(bad)); _write(b2, _fastq(r2))
    p = os.path.join(tmp, "bad")
    res = _run(binary, sim_db, ["--paired-reads", b1 + "," + b2], p)
    assert "Error parsing file" in res.stderr and "'!'" in res.stderr
    assert _outputs(p) == _outputs(p_ref)
    # (2) truncated record (no quality line) at the end
    t1 = os.path.join(tmp, "t1.fq")
    _write(t1, _fastq(r1[:10]) + f"@{r1[10][0]}\n{r1[10][1]}\n+\n")
    p = os.path.join(tmp, "trunc")
    res = _run(binary, sim_db, ["--paired-reads", t1 + "," + good2], p)
    assert "Error parsing file" in res.stderr
    assert _outputs(p) == _outputs(p_ref)
    # (3) the error only stops THAT file: a second, good pair of files is still processed
    p = os.path.join(tmp, "two")
    res = _run(binary, sim_db, ["--paired-reads", ",".join([b1, b2, good1, good2])], p)
    assert "Error parsing file" in res.stderr
    two = cu.Res(p)
    one = cu.Res(p_ref)
    assert two.total_classified + two.total_unclassified == 2 * (one.total_classified + one.total_unclassified)
    # (4) mates file shorter than file 1: the missing mates count as empty
    s2 = os.path.join(tmp, "s2.fq")
    _write(s2, _fastq(r2[:5]))
    p = os.path.join(tmp, "short")
    _run(binary, sim_db, ["--paired-reads", good1 + "," + s2], p)
    r = cu.Res(p)
    assert r.total_classified + r.total_unclassified == 10
    # (4a) illegal letter in record 11 of the MATES file (parsed on its own thread): the error arrives in place --
    # ten complete pairs, the eleventh read keeps an empty mate, nothing after it
    badm = list(r2)
    badm[10] = (badm[10][0], badm[10][1][:20] + "?" + badm[10][1][21:])
    bm = os.path.join(tmp, "bm2.fq")
    _write(bm, _fastq(badm))
    p = os.path.join(tmp, "badmate")
    res = _run(binary, sim_db, ["--paired-reads", os.path.join(tmp, "full1.fq") + "," + bm], p)
    assert "Error parsing file" in res.stderr and "'?'" in res.stderr
    r = cu.Res(p)
    assert r.total_classified + r.total_unclassified == 11
    # (4b) blocked gzip with a damaged member: error reported, nothing after the damage is used
    bz = os.path.join(tmp, "dmg.fq.gz")
    _write_bgzf(bz, _fastq(r1), block=700)
    raw = bytearray(open(bz, "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    open(bz, "wb").write(bytes(raw))
    p = os.path.join(tmp, "dmg")
    res = _run(binary, sim_db, ["--single-reads", bz], p)
    assert "Error parsing file" in res.stderr
    # (4c) bzip2 with a damaged block: error reported
    import bz2
    bzf = os.path.join(tmp, "dmg.fq.bz2")
    rawb = bytearray(bz2.compress(_fastq(r1).encode()))
    rawb[len(rawb) // 2] ^= 0xFF
    open(bzf, "wb").write(bytes(rawb))
    p = os.path.join(tmp, "dmgbz")
    res = _run(binary, sim_db, ["--single-reads", bzf], p, check=False)
    assert "Error parsing file" in res.stderr
    # (4d) two bzip2 streams, the second one damaged: the first stream's records are all delivered, then the error
    two = bytearray(bz2.compress(_fastq(r1[:10]).encode()) + bz2.compress(_fastq(r1[10:]).encode()))
    two[len(two) - 40] ^= 0xFF
    bz2f = os.path.join(tmp, "dmg2.fq.bz2")
    open(bz2f, "wb").write(bytes(two))
    p = os.path.join(tmp, "dmgbz2")
    res = _run(binary, sim_db, ["--single-reads", bz2f], p, check=False)
    assert "Error parsing file" in res.stderr
    _write(os.path.join(tmp, "first10.fq"), _fastq(r1[:10]))
    p10 = os.path.join(tmp, "first10")
    _run(binary, sim_db, ["--single-reads", os.path.join(tmp, "first10.fq")], p10)
    got, want = _outputs(p), _outputs(p10)
    assert got[".all"].startswith(want[".all"]) and len(got[".all"]) >= len(want[".all"])
    # (5) unknown extension
    u = os.path.join(tmp, "reads.txt")
    _write(u, _fastq(r1))
    p = os.path.join(tmp, "unk")
    res = _run(binary, sim_db, ["--single-reads", u], p, check=False)
    assert "Error parsing file" in res.stderr


def test_parse_errors_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_errors(oracle_bin, sim_db, str(tmp_path))


def _check_long_line(binary, sim_db, tmp):
    """one 6 Mbp sequence on a single line (longer than the reader's buffer) == the same sequence wrapped"""
    rng = np.random.default_rng(4)
    tgt = list(sim_db["targets"].values())[3]
    seq = "".join("ACGT"[x] for x in rng.integers(0, 4, size=6_000_000))
    seq = seq[:3_000_000] + tgt + seq[3_000_000:]
    a, b = os.path.join(tmp, "long1.fa"), os.path.join(tmp, "long2.fa")
    mid = seq[:200_000] + tgt + seq[200_000:400_000]
    _write(a, f">chr\n{seq}\n>mid\n{mid}\n>tail\n{tgt[:400]}")
    _write(b, f">chr\n{_wrap(seq, 80)}\n>mid\n{_wrap(mid, 80)}\n>tail\n{_wrap(tgt[:400], 80)}\n")
    outs = []
    for fn in (a, b):
        p = os.path.join(tmp, "o_" + os.path.basename(fn))
        _run(binary, sim_db, ["--single-reads", fn], p, cutoff="0")
        outs.append(_outputs(p))
    assert outs[0] == outs[1]
    # chr has more than 65535 minimisers -> skipped like the reference does (:674); the records after it are intact
    assert outs[0][".unc"] == b"chr\n" and b"mid\tT3.1\t" in outs[0][".all"] and b"tail\tT3.1\t" in outs[0][".all"]


def test_long_line_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_long_line(oracle_bin, sim_db, str(tmp_path))


@pytest.mark.gpu
def test_formats_errors_long_line_hip(sim_db, tmp_path):
    _check_variants(cu.BIN_HIP, sim_db, str(tmp_path))
    _check_errors(cu.BIN_HIP, sim_db, str(tmp_path))
    _check_long_line(cu.BIN_HIP, sim_db, str(tmp_path))


# ---------------------------------------------------------------------------------------------------------------
# parallel FASTQ slabs (ganon_amd/host/seq_io.cpp ParallelFastq): same batches' content, same outputs as the sequential reader
# ---------------------------------------------------------------------------------------------------------------
def _fastq_text(recs, wrap=0, crlf=False, blank_after=None):
    nl = "\r\n" if crlf else "\n"
    out = []
    for i, (rid, s) in enumerate(recs):
        q = "".join("I@+#"[(i + j) % 4] for j in range(len(s)))  # quality lines that start with '@' and '+'
        if wrap and len(s) > wrap and i % 5 == 0:
            body = nl.join(s[a:a + wrap] for a in range(0, len(s), wrap))
            qual = nl.join(q[a:a + wrap] for a in range(0, len(q), wrap))
        else:
            body, qual = s, q
        out.append(f"@{rid}{nl}{body}{nl}+{nl}{qual}{nl}")
        if blank_after is not None and i == blank_after:
            out.append(nl)
    return "".join(out)


def _run_reader_case(binary, ibf, files, out, paired, env, extra_args=()):
    args = ["--ibf", ibf, "-o", out, "--output-all", "--output-unclassified", "--rel-cutoff", "0.3", "--quiet"] + list(extra_args)
    args += ["--paired-reads", ",".join(files)] if paired else ["--single-reads", files[0]]
    p = subprocess.run([binary] + args, capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)  # (a hang fails the test)
    assert p.returncode == 0, p.stderr
    return p.stderr, {e: open(out + e, "rb").read() for e in (".all", ".unc", ".rep")}


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("variant", ["plain", "crlf", "wrapped", "blank_line", "bad_letter", "mate_short", "mate_bad"])
def test_parallel_fastq_equals_sequential_reader(oracle_bin, sim_db, tmp_path, paired, variant):
    import numpy as np
    rng = np.random.default_rng(17)
    n = 1500
    g = list(sim_db["targets"].values())
    db = sim_db
    recs1, recs2 = [], []
    for i in range(n):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        L = int(rng.integers(40, 152))
        recs1.append((f"read{i} extra words", src[p:p + L]))
        recs2.append((f"read{i}/2", src[p + 150:p + 150 + L][::-1].translate(str.maketrans("ACGT", "TGCA"))))
    if variant == "bad_letter":
        recs1[n // 2] = (recs1[n // 2][0], recs1[n // 2][1][:20] + "!" + recs1[n // 2][1][21:])
    if variant == "mate_short":
        recs2 = recs2[: n // 3]
    if variant == "mate_bad":
        recs2[2 * n // 3] = (recs2[2 * n // 3][0], "ACGT*ACGT" * 8)
    kw = dict(crlf=variant == "crlf", wrap=40 if variant == "wrapped" else 0, blank_after=n // 4 if variant == "blank_line" else None)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    open(f1, "w", newline="").write(_fastq_text(recs1, **kw))
    open(f2, "w", newline="").write(_fastq_text(recs2, **(kw if variant in ("crlf",) else {})))
    files = [f1, f2] if paired else [f1]
    if not paired and variant.startswith("mate"):
        pytest.skip("paired-only case")
    seq_err, seq_out = _run_reader_case(oracle_bin, db["ibf"], files, str(tmp_path / "seq"), paired, {"GANON_HOST_PARSE_THREADS": "0"})
    for slab in ("65536", "200000"):
        par_err, par_out = _run_reader_case(oracle_bin, db["ibf"], files, str(tmp_path / ("par" + slab)), paired,
                                            {"GANON_HOST_PARSE_THREADS": "4", "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_PARALLEL_MIN": "0",
                                             "GANON_HOST_BATCH_READS": "333"})
        assert par_out == seq_out, (variant, slab)
        assert ("Error parsing" in par_err) == ("Error parsing" in seq_err) == (variant in ("bad_letter", "mate_bad"))
    assert seq_out[".all"].count(b"\n") > 100


RAW_VARIANTS = ["plain", "crlf", "wrapped", "blank_line", "bad_letter", "no_final_newline", "truncated", "cr_letters", "empty_reads", "tiny_records"]


def _raw_case_file(variant, sim_db, tmp_path):
    import numpy as np
    rng = np.random.default_rng(23)
    n = 4000
    g = list(sim_db["targets"].values())
    recs = []
    for i in range(n):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        L = int(rng.integers(40, 152)) if variant != "tiny_records" else int(rng.integers(0, 6))
        if variant == "empty_reads" and i % 11 == 0:
            L = 0
        recs.append((f"read{i} extra words" if variant != "tiny_records" else f"{i}", src[p:p + L]))
    if variant == "bad_letter":
        recs[n // 2] = (recs[n // 2][0], recs[n // 2][1][:20] + "!" + recs[n // 2][1][21:])
    kw = dict(crlf=variant == "crlf", wrap=40 if variant == "wrapped" else 0, blank_after=n // 4 if variant == "blank_line" else None)
    text = _fastq_text(recs, **kw)
    if variant == "no_final_newline":
        text = text[:-1]
    if variant == "truncated":
        text = text[: len(text) * 2 // 3 + 17]
    if variant == "cr_letters":
        text = "".join(f"@{rid}\n{s}\r\n+\n{'I' * len(s)}\n" for rid, s in recs)
    f1 = str(tmp_path / "r1.fq")
    open(f1, "w", newline="").write(text)
    return f1


@pytest.mark.gpu
@pytest.mark.parametrize("variant", RAW_VARIANTS)
def test_device_tokenised_fastq_equals_sequential_reader(oracle_bin, sim_db, tmp_path, variant):
    """Uncompressed single-end FASTQ reaches the HIP backend as pieces of the file; the records are found on the device
    (csrc/gn_fastq.hip) and the file's pieces are accepted in file order.  Output and messages are those of the sequential reader
    (checker backend, one thread), of the host's slab parser, and do not depend on piece size or worker count."""
    f1 = _raw_case_file(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / "seq"), False, {"GANON_HOST_PARSE_THREADS": "0"})
    common = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": "1"}
    for slab, extra, tag in (("65536", (), "a"), ("200000", ("--device", "0"), "b"), ("1048576", ("--device", "0,0"), "c"), ("65536", ("--device", "0,0"), "d")):
        err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1], str(tmp_path / ("dev" + slab + tag)), False,
                                    dict(common, GANON_HOST_SLAB_BYTES=slab), extra)
        assert out == seq_out, (variant, slab)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant not in ("crlf", "wrapped"):
            assert "pieces of FASTQ text tokenised on the device" in err, err[-600:]
    # ... and with the host's slab parser instead
    err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1], str(tmp_path / "host"), False,
                                dict(common, GANON_HOST_SLAB_BYTES="65536", GANON_HOST_DEVICE_FASTQ="0"))
    assert out == seq_out and "tokenised on the device" not in err
    if variant in ("plain", "cr_letters", "empty_reads"):
        assert seq_out[".all"].count(b"\n") > 300


@pytest.mark.parametrize("variant", RAW_VARIANTS)
def test_raw_pieces_and_worker_lanes_with_the_checker_backend(oracle_bin, sim_db, tmp_path, variant):
    """The host side of the same machinery without a GPU: the checker backend finds the records of raw pieces by the slab parser's
    rule and offers twin contexts, so the reader's raw mode, the in-file-order acceptance of pieces (a piece that is not records
    from end to end stops its file, later pieces are dropped, the sequential reader goes on there) and the worker threads' lanes
    run here with many small batches and several workers -- same bytes as the sequential reader, and no run may hang."""
    f1 = _raw_case_file(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / "seq"), False, {"GANON_HOST_PARSE_THREADS": "0"})
    for raw, lanes, slab, dev in (("1", "2", "65536", "0,0,0"), ("1", "3", "70000", "0,0"), ("1", "1", "200000", "0"),
                                  ("1", "2", "65536", "0,0"), ("0", "2", "65536", "0,0,0,0"), ("0", "3", "65536", "0")):
        env = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": raw,
               "GANON_HOST_LANES": lanes, "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_BATCH_READS": "97", "GANON_HOST_POST_THREADS": "2"}
        err, out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / f"o{raw}{lanes}{slab}{dev.count(',')}"), False, env, ("--device", dev))
        assert out == seq_out, (variant, raw, lanes, slab, dev)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant not in ("crlf", "wrapped"):  # (a file whose FIRST record is not taken never delivers a piece)
            assert ("tokenised on the device" in err) == (raw == "1"), err[-400:]


@pytest.mark.parametrize("raw", ["0", "1"])
def test_a_failing_batch_ends_the_run_and_nothing_hangs(oracle_bin, sim_db, tmp_path, raw):
    """A backend error in the middle of a file: every stage lets go (reader waiting for its pieces, workers waiting for their turn or
    for earlier pieces, post pool, writer) and the binary exits with the message -- with raw pieces and with parsed batches, several
    workers, two contexts each."""
    f1 = _raw_case_file("plain", sim_db, tmp_path)
    for fail_at, dev in (("1", "0"), ("5", "0,0,0"), ("11", "0,0")):  # (the file is 16 raw pieces)
        env = dict(os.environ, GANON_HOST_PARSE_THREADS="3", GANON_HOST_PARALLEL_MIN="0", GANON_HOST_DEVICE_FASTQ=raw, GANON_HOST_LANES="2",
                   GANON_HOST_SLAB_BYTES="65536", GANON_HOST_BATCH_READS="97", GANON_TEST_FAIL_AT_BATCH=fail_at)
        p = subprocess.run([oracle_bin, "--ibf", sim_db["ibf"], "--single-reads", f1, "-o", str(tmp_path / "x"), "--output-all", "--quiet", "--device", dev],
                           capture_output=True, text=True, env=env, timeout=120)
        assert p.returncode != 0 and "injected failure" in p.stderr, (fail_at, dev, p.returncode, p.stderr[-300:])


def _fasta_text(recs, variant):
    nl = "\r\n" if variant == "crlf" else "\n"
    out = []
    for i, (rid, seq) in enumerate(recs):
        if variant == "lower_iupac" and i % 4 == 0:
            seq = seq.lower().replace("a", "r", 1).replace("c", "y", 1)
        if variant == "spaces_digits" and i % 3 == 0:
            seq = " ".join(seq[a:a + 10] for a in range(0, len(seq), 10)) + " 60"
        width = 60 if variant in ("wrapped", "crlf", "blank_lines") else 10 ** 9
        lines = [seq[a:a + width] for a in range(0, len(seq), width)] or [""]
        if variant == "blank_lines" and i % 7 == 0:
            lines.insert(1, "")
        out.append(f">{rid}{nl}" + nl.join(lines) + nl)
        if variant == "blank_lines" and i % 11 == 0:
            out.append(nl)
        if variant == "semicolon" and i == len(recs) // 2:
            out.append(f";old-style header{nl}ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT{nl}")
    return "".join(out)


FASTA_RAW_VARIANTS = ["plain", "wrapped", "crlf", "blank_lines", "lower_iupac", "spaces_digits", "semicolon", "bad_letter", "empty_records",
                      "leading_blank", "one_wrapped_record", "no_final_newline"]


def _fasta_raw_case_file(variant, sim_db, tmp_path):
    import numpy as np
    rng = np.random.default_rng(37)
    g = list(sim_db["targets"].values())
    recs = []
    for i in range(4000):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        L = int(rng.integers(40, 152))
        if variant == "empty_records" and i % 9 == 0:
            L = 0
        recs.append((f"read{i} extra words", src[p:p + L]))
    if variant == "bad_letter":
        recs[2000] = (recs[2000][0], recs[2000][1][:20] + "!" + recs[2000][1][21:])
    if variant == "one_wrapped_record":   # two-line records all over, one record in the middle with its letters on three lines
        text = _fasta_text(recs[:2100], "plain") + ">wrapped one\n" + recs[2100][1][:30] + "\n" + recs[2100][1][30:60] + "\n" + recs[2100][1][60:] + "\n" \
            + _fasta_text(recs[2101:], "plain")
    else:
        text = _fasta_text(recs, variant if variant not in ("empty_records", "bad_letter", "leading_blank", "no_final_newline") else "plain")
    if variant == "leading_blank":
        text = "\n" + text
    if variant == "no_final_newline":
        text = text[:-1]
    f1 = str(tmp_path / "r1.fasta")
    open(f1, "w", newline="").write(text)
    return f1


@pytest.mark.parametrize("variant", FASTA_RAW_VARIANTS)
def test_raw_fasta_pieces_with_the_checker_backend(oracle_bin, sim_db, tmp_path, variant):
    """FASTA text as raw pieces (records of two lines found by the backend; whatever the two-line rule does not cover -- wrapped
    letters, blank lines, ';' headers, white space and digits among the letters -- stops the file's pieces there and continues in
    the sequential reader): the host side with the checker backend, small pieces, several workers and lanes."""
    f1 = _fasta_raw_case_file(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / "seq"), False, {"GANON_HOST_PARSE_THREADS": "0"})
    for raw, lanes, slab, dev in (("1", "2", "65536", "0,0,0"), ("1", "1", "200000", "0"), ("1", "3", "70000", "0,0"), ("0", "2", "65536", "0,0")):
        env = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": raw,
               "GANON_HOST_LANES": lanes, "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_BATCH_READS": "97", "GANON_HOST_POST_THREADS": "2"}
        err, out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / f"o{raw}{lanes}{slab}"), False, env, ("--device", dev))
        assert out == seq_out, (variant, raw, lanes, slab, dev)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant != "leading_blank":  # (a first line that is no header: how far the pieces get is the reader's business)
            assert ("tokenised on the device" in err) == (raw == "1"), err[-400:]
    if variant in ("plain", "lower_iupac", "empty_records", "one_wrapped_record"):
        assert seq_out[".all"].count(b"\n") > 300


@pytest.mark.gpu
@pytest.mark.parametrize("variant", FASTA_RAW_VARIANTS)
def test_device_tokenised_fasta_equals_sequential_reader(oracle_bin, sim_db, tmp_path, variant):
    """the same through the HIP binary: the two-line records are found on the device (csrc/gn_fastq.hip, GN_TEXT_FASTA)"""
    f1 = _fasta_raw_case_file(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1], str(tmp_path / "seq"), False, {"GANON_HOST_PARSE_THREADS": "0"})
    common = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": "1"}
    for slab, extra in (("65536", ()), ("200000", ("--device", "0")), ("1048576", ("--device", "0,0")), ("65536", ("--device", "0,0"))):
        err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1], str(tmp_path / ("dev" + slab + str(len(extra)))), False,
                                    dict(common, GANON_HOST_SLAB_BYTES=slab), extra)
        assert out == seq_out, (variant, slab)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant != "leading_blank":
            assert "tokenised on the device" in err, err[-600:]
    err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1], str(tmp_path / "host"), False,
                                dict(common, GANON_HOST_SLAB_BYTES="65536", GANON_HOST_DEVICE_FASTQ="0"))
    assert out == seq_out and "tokenised on the device" not in err


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("variant", ["plain", "wrapped", "crlf", "blank_lines", "lower_iupac", "spaces_digits", "semicolon", "bad_letter",
                                     "empty_records", "leading_blank"])
def test_parallel_fasta_equals_sequential_reader(oracle_bin, sim_db, tmp_path, paired, variant):
    # uncompressed FASTA goes through the same slab parser as FASTQ (records start at '>' lines); whatever it does not take
    # (a ';' header, a file that does not begin with a header) continues in the sequential reader at that byte
    import numpy as np
    rng = np.random.default_rng(23)
    n = 1200
    g = list(sim_db["targets"].values())
    recs1, recs2 = [], []
    for i in range(n):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=900))
        L = int(rng.choice([40, 100, 150, 320]))
        p = int(rng.integers(0, len(src) - 2 * L))
        recs1.append((f"read{i} some words", src[p:p + L]))
        recs2.append((f"read{i}/2", src[p + L:p + 2 * L]))
    if variant == "bad_letter":
        recs1[n // 2] = (recs1[n // 2][0], recs1[n // 2][1][:30] + "!" + recs1[n // 2][1][31:])
    if variant == "empty_records":
        for i in range(0, n, 97):
            recs1[i] = (recs1[i][0], "")
    f1, f2 = str(tmp_path / "r1.fasta"), str(tmp_path / "r2.fa")
    text1 = _fasta_text(recs1, variant)
    if variant == "leading_blank":
        text1 = "\n\n" + text1
    open(f1, "w", newline="").write(text1)
    open(f2, "w", newline="").write(_fasta_text(recs2, "crlf" if variant == "crlf" else "plain"))
    files = [f1, f2] if paired else [f1]
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / "seq"), paired, {"GANON_HOST_PARSE_THREADS": "0"})
    for slab in ("40000", "250000"):
        par_err, par_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / ("par" + slab)), paired,
                                            {"GANON_HOST_PARSE_THREADS": "4", "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_PARALLEL_MIN": "0",
                                             "GANON_HOST_BATCH_READS": "277"})
        assert par_out == seq_out, (variant, slab)
        assert ("Error parsing" in par_err) == ("Error parsing" in seq_err) == (variant == "bad_letter")
    assert seq_out[".all"].count(b"\n") > 100


# ---------------------------------------------------------------------------------------------------------------
# ordinary gzip input, inflated by several threads (host/pgzip.cpp): the sequential zlib reader's records, byte for byte
# ---------------------------------------------------------------------------------------------------------------
def _gz_variants(raw: bytes, variant: str) -> bytes:
    import gzip
    import zlib
    if variant in ("level1", "level6", "level9"):
        return gzip.compress(raw, int(variant[-1]))
    if variant == "multi_member":          # what `cat a.gz b.gz c.gz` gives; members end in the middle of records
        cut = [0, len(raw) // 3 + 7, len(raw) // 3 + 1000, 2 * len(raw) // 3, len(raw)]
        return b"".join(gzip.compress(raw[a:b], 6) for a, b in zip(cut, cut[1:]))
    if variant == "flush_points":          # sync / full flushes: empty stored blocks between the others
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        parts = []
        for i in range(0, len(raw), 50_000):
            parts += [co.compress(raw[i:i + 50_000]), co.flush(zlib.Z_FULL_FLUSH if (i // 50_000) % 2 else zlib.Z_SYNC_FLUSH)]
        return b"".join(parts) + co.flush()
    if variant == "fixed_codes":
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
        return co.compress(raw) + co.flush()
    if variant == "stored":
        co = zlib.compressobj(0, zlib.DEFLATED, 31)
        return co.compress(raw) + co.flush()
    if variant == "trailing_garbage":
        return gzip.compress(raw, 6) + b"\0" * 100 + b"garbage"
    if variant == "truncated":
        z = gzip.compress(raw, 6)
        return z[: 2 * len(z) // 3]
    if variant == "corrupt":
        z = bytearray(gzip.compress(raw, 6))
        z[len(z) // 2] ^= 0x55
        return bytes(z)
    if variant == "bad_crc":
        z = bytearray(gzip.compress(raw, 6))
        z[-6] ^= 0xFF
        return bytes(z)
    raise ValueError(variant)


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("variant", ["level1", "level6", "level9", "multi_member", "flush_points", "fixed_codes", "stored", "trailing_garbage",
                                     "truncated", "corrupt", "bad_crc", "wrapped_records", "fasta"])
def test_parallel_gzip_equals_sequential_reader(oracle_bin, sim_db, tmp_path, paired, variant):
    import numpy as np
    rng = np.random.default_rng(29)
    n = 6000
    g = list(sim_db["targets"].values())
    recs1, recs2 = [], []
    for i in range(n):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        L = int(rng.integers(60, 152))
        recs1.append((f"SRR000001.{i} {i} length={L}", src[p:p + L]))
        recs2.append((f"SRR000001.{i}/2", src[p + 150:p + 150 + L][::-1].translate(str.maketrans("ACGT", "TGCA"))))
    ext = ".fa.gz" if variant == "fasta" else ".fq.gz"
    if variant == "fasta":
        t1, t2 = _fasta_text(recs1, "wrapped"), _fasta_text(recs2, "plain")
    else:
        t1, t2 = _fastq_text(recs1, wrap=50 if variant == "wrapped_records" else 0), _fastq_text(recs2)
    gzv = variant if variant not in ("wrapped_records", "fasta") else "level6"
    f1, f2 = str(tmp_path / ("r1" + ext)), str(tmp_path / ("r2" + ext))
    open(f1, "wb").write(_gz_variants(t1.encode(), gzv))
    open(f2, "wb").write(_gz_variants(t2.encode(), gzv if variant in ("level1", "level9", "multi_member") else "level6"))
    files = [f1, f2] if paired else [f1]
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / "seq"), paired, {"GANON_HOST_PARSE_THREADS": "0"})
    broken = variant in ("truncated", "corrupt", "bad_crc")
    assert ("Error parsing" in seq_err) == broken
    for chunk, slab in (("4096", "65536"), ("70000", "300000")):
        env = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_BATCH_READS": "777",
               "GANON_HOST_INFLATE_CHUNK": chunk, "GANON_HOST_INFLATE_THREADS": "3", "GANON_HOST_TIMING": "1"}
        par_err, par_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / ("par" + chunk)), paired, env)
        assert par_out == seq_out, (variant, chunk)
        assert ("Error parsing" in par_err) == broken
        off_err, off_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / ("off" + chunk)), paired,
                                            dict(env, GANON_HOST_NO_PGZIP="1"))
        assert off_out == seq_out
    assert broken or seq_out[".all"].count(b"\n") > 300   # (zlib hands out nothing of the buffer an error turns up in)


@pytest.mark.parametrize("paired", [False, True])
def test_gzip_damaged_far_into_the_stream_equals_sequential_reader(oracle_bin, sim_db, tmp_path, paired):
    # The damage lies more than 8 MiB into the decompressed stream, the slabs are small and two parsers take adjacent ones: a
    # slab's search for its first record can run into the damage while the slab before it came out whole (the search reads
    # ahead in 4 MiB pieces).  The sequential reader must then take over where that slab ended -- not at the file's start,
    # which delivered every read before the damage twice (and misaligned the mates of a pair).
    import gzip
    import numpy as np
    rng = np.random.default_rng(31)
    g = list(sim_db["targets"].values())
    n = 52_000
    recs1, recs2 = [], []
    for i in range(n):
        src = g[i % len(g)] if i % 4 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        recs1.append((f"SRR000009.{i} {i} length=150", src[p:p + 150]))
        recs2.append((f"SRR000009.{i}/2", src[p + 150:p + 300][::-1].translate(str.maketrans("ACGT", "TGCA"))))
    t1, t2 = _fastq_text(recs1).encode(), _fastq_text(recs2).encode()
    assert len(t1) > 14 << 20
    z1, z2 = gzip.compress(t1, 1), gzip.compress(t2, 1)
    f1, f2 = str(tmp_path / "r1.fq.gz"), str(tmp_path / "r2.fq.gz")
    open(f1, "wb").write(z1[: int(len(z1) * 0.9)])                     # truncated ~13 MiB into the text
    open(f2, "wb").write(z2 if not paired else z2[: int(len(z2) * 0.95)])
    files = [f1, f2] if paired else [f1]
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / "seq"), paired, {"GANON_HOST_PARSE_THREADS": "0"})
    assert "Error parsing" in seq_err and seq_out[".all"].count(b"\n") > 10_000
    for slab, threads in (("65536", "2"), ("3000000", "2"), ("1000000", "5")):
        env = {"GANON_HOST_PARSE_THREADS": threads, "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_INFLATE_THREADS": "3"}
        par_err, par_out = _run_reader_case(oracle_bin, sim_db["ibf"], files, str(tmp_path / f"par{slab}_{threads}"), paired, env)
        assert par_out == seq_out, (slab, threads, par_out[".all"].count(b"\n"), seq_out[".all"].count(b"\n"))
        assert "Error parsing" in par_err


# ---------------------------------------------------------------------------------------------------------------
# pairs as text: pieces of both mate files, the second cut at the first's record count by a line index of the file
# ---------------------------------------------------------------------------------------------------------------
PAIR_TEXT_VARIANTS = ["plain", "ids_differ_in_length", "crlf", "file2_shorter", "file2_longer", "file2_much_shorter", "bad_letter_in_2", "bad_letter_in_1",
                      "wrapped_in_2", "wrapped_in_1", "blank_line_in_2", "no_final_newline_1", "no_final_newline_2", "empty_mates", "fasta", "fasta_wrapped_in_2"]


def _pair_text_files(variant, sim_db, tmp_path):
    import numpy as np
    rng = np.random.default_rng(41)
    g = list(sim_db["targets"].values())
    n = 5000
    recs1, recs2 = [], []
    for i in range(n):
        src = g[i % len(g)] if i % 3 else "".join("ACGT"[x] for x in rng.integers(0, 4, size=400))
        p = int(rng.integers(0, len(src) - 310))
        L1, L2 = int(rng.integers(40, 152)), int(rng.integers(40, 152))
        if variant == "empty_mates" and i % 6 == 0:
            L2 = 0
        id2 = f"read{i}/2" if variant != "ids_differ_in_length" else f"mate_of_read_{i}_with_a_much_longer_identifier_{'x' * (i % 37)}/2"
        recs1.append((f"read{i} extra words", src[p:p + L1]))
        recs2.append((id2, src[p + 150:p + 150 + L2][::-1].translate(str.maketrans("ACGT", "TGCA"))))
    if variant == "bad_letter_in_2":
        recs2[3100] = (recs2[3100][0], recs2[3100][1][:10] + "!" + recs2[3100][1][11:])
    if variant == "bad_letter_in_1":
        recs1[2900] = (recs1[2900][0], recs1[2900][1][:10] + "!" + recs1[2900][1][11:])
    if variant == "file2_shorter":
        recs2 = recs2[:3700]
    if variant == "file2_much_shorter":
        recs2 = recs2[:40]
    if variant == "file2_longer":
        recs1 = recs1[:4100]
    fasta = variant.startswith("fasta")
    if fasta:
        t1 = _fasta_text(recs1, "plain")
        t2 = _fasta_text(recs2, "plain") if variant == "fasta" else _fasta_text(recs2[:2500], "plain") + _fasta_text(recs2[2500:2600], "wrapped") + _fasta_text(recs2[2600:], "plain")
    else:
        t1 = _fastq_text(recs1, wrap=50 if variant == "wrapped_in_1" else 0, crlf=variant == "crlf")
        t2 = _fastq_text(recs2, wrap=50 if variant == "wrapped_in_2" else 0, crlf=variant == "crlf", blank_after=2000 if variant == "blank_line_in_2" else None)
    if variant == "no_final_newline_1":
        t1 = t1[:-1]
    if variant == "no_final_newline_2":
        t2 = t2[:-1]
    ext = ".fa" if fasta else ".fq"
    f1, f2 = str(tmp_path / ("r1" + ext)), str(tmp_path / ("r2" + ext))
    open(f1, "w", newline="").write(t1)
    open(f2, "w", newline="").write(t2)
    return f1, f2


@pytest.mark.parametrize("variant", PAIR_TEXT_VARIANTS)
def test_pair_text_pieces_with_the_checker_backend(oracle_bin, sim_db, tmp_path, variant):
    """Both mate files as raw pieces: file 1's pieces with the matching lines of file 2 (line index), records found by the backend,
    the first piece that is not records from end to end in both files stops the pair there and the sequential readers go on --
    same bytes as the sequential reader whatever the piece size, the workers and the lanes."""
    f1, f2 = _pair_text_files(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1, f2], str(tmp_path / "seq"), True, {"GANON_HOST_PARSE_THREADS": "0"})
    assert seq_out[".all"].count(b"\n") > 300
    for raw, lanes, slab, dev in (("1", "2", "131072", "0,0,0"), ("1", "1", "400000", "0"), ("1", "3", "140000", "0,0"), ("0", "2", "131072", "0,0")):
        env = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": raw,
               "GANON_HOST_LANES": lanes, "GANON_HOST_SLAB_BYTES": slab, "GANON_HOST_BATCH_READS": "97", "GANON_HOST_POST_THREADS": "2",
               "GANON_HOST_PAIR_TEXT": "1"}
        err, out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1, f2], str(tmp_path / f"o{raw}{lanes}{slab}"), True, env, ("--device", dev))
        assert out == seq_out, (variant, raw, lanes, slab, dev)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant != "crlf":
            assert ("tokenised on the device" in err) == (raw == "1"), err[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", PAIR_TEXT_VARIANTS)
def test_device_tokenised_pairs_equal_sequential_reader(oracle_bin, sim_db, tmp_path, variant):
    """the same through the HIP binary: both texts tokenised on the device (gn_stream_upload_text_pair)"""
    f1, f2 = _pair_text_files(variant, sim_db, tmp_path)
    seq_err, seq_out = _run_reader_case(oracle_bin, sim_db["ibf"], [f1, f2], str(tmp_path / "seq"), True, {"GANON_HOST_PARSE_THREADS": "0"})
    common = {"GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_TIMING": "1", "GANON_HOST_DEVICE_FASTQ": "1",
              "GANON_HOST_PAIR_TEXT": "1"}
    for slab, extra in (("131072", ()), ("400000", ("--device", "0")), ("2097152", ("--device", "0,0")), ("131072", ("--device", "0,0"))):
        err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1, f2], str(tmp_path / ("dev" + slab + str(len(extra)))), True,
                                    dict(common, GANON_HOST_SLAB_BYTES=slab), extra)
        assert out == seq_out, (variant, slab)
        assert ("Error parsing" in err) == ("Error parsing" in seq_err)
        if variant != "crlf":
            assert "tokenised on the device" in err, err[-600:]
    err, out = _run_reader_case(cu.BIN_HIP, sim_db["ibf"], [f1, f2], str(tmp_path / "host"), True,
                                dict(common, GANON_HOST_SLAB_BYTES="131072", GANON_HOST_PAIR_TEXT="0"))
    assert out == seq_out and "tokenised on the device" not in err
