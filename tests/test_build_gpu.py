"""ganon-build on the GPU, following /root/reference/tests/ganon-build/GanonBuild.test.cpp SECTION by SECTION
(validate_filter / validate_elements), plus what the reference's tests cannot check: the header against the oracle's
sizing (oracle/build_params.py) and every bit of the filter against an oracle-built one."""
import gzip
import math
import os
import subprocess

import numpy as np
import pytest

import oracle
from oracle import build_params as bp
from test_build_cpu import BIN_BUILD, DATA, read_fasta_gz

pytestmark = pytest.mark.gpu

# the ten 80-mers of GanonBuild.test.cpp:100-109 and the shortened ones of :497-508 (data of the reference's tests)
SEQS = ["ACACTCTTTGAAAATGCATATAATATTGAACGTTATTTTGAAATAGATTAATTACTCATATCCATTTGCTAATCTTATCG",
        "TTTATTATATGTAATTATAAATTTATCGTTAAGCTTGACATAAGTGAGTGTATCTATGTTCTTAACAAATACATCGCGTT",
        "TTTTATTTTTATTTCTTATGCACAAGAATAAATTATATGCATATGATAATTTCTCATTCAATGCGGATGTACATTATGGT",
        "TATGGTAAGCTATTATGGCATGATAAAAAACCAGTCATATACCCATTGGCATCCTTATCTGATTATACTTATTATAACGA",
        "ATCCGACCCATTTGAAACGATTTATTATGTGGAGCAATACTATAAAATTAGCTTAAATGAGAGTAAGCGAATTCAAGAAC",
        "AAAAGGACATTTACGCACACCTTCAATTAAAACATAATAAATCATTAATTACAGCAAATGTAACGTTACATAATAAAAGT",
        "AATAGTTCGTATTATGTTCATCGGATGAATTTACCAGCAAACATCCATGAATCACCTTACTCTCCTTTGTGCAGTGGTTC",
        "TTTTTTAATCGTAACAAATAACATACGGTTAGATTATATAAGAAAAATTACATGCCGATTTGATTTGTGGATAAAAAAAT",
        "CTGACTGGATAGAAATATCACCCGGAGAAAAACTCTCATACACAGTAAATTTGAATGACTATTATGCTTTTCTCCCTGCG",
        "ATGCATCAATATGATATAGGAACTGTAGAGTTCACATTGGTAAATAGTAATTGGTTCTTAGAACAGCATATTTATGATCT"]
SEQS2 = [s[:n] for s, n in zip(SEQS, (80, 75, 70, 65, 60, 55, 50, 45, 40, 35))]


@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    assert os.path.exists(BIN_BUILD), "ganon-build is built by __graft_entry__.build()"
    return ganon_amd


def write_inputs(d, seqs, targets=None):
    """aux::SeqTarget (tests/aux/Aux.hpp:142-237): one FASTA per sequence, header SEQ<i>; default target = the file name"""
    files = []
    for i, s in enumerate(seqs):
        f = os.path.join(d, f"case.SEQ{i}.fasta")
        open(f, "w").write(f">SEQ{i}\n{s}\n")
        files.append(f)
    inp = os.path.join(d, "case_input.tsv")
    with open(inp, "w") as o:
        for i, f in enumerate(files):
            o.write(f if targets is None else f"{f}\t{targets[i]}")
            o.write("\n")
    names = [os.path.basename(f) for f in files] if targets is None else list(targets)
    return inp, files, names


def run_build(d, inp, extra=(), k=19, w=32, h=4, max_fp=0.05, expect=0):
    out = os.path.join(d, "case.ibf")
    args = [BIN_BUILD, "--input-file", inp, "--output-file", out, "--quiet", "--kmer-size", str(k), "--window-size", str(w),
            "--hash-functions", str(h)]
    if max_fp is not None:
        args += ["--max-fp", repr(max_fp)]
    p = subprocess.run(args + list(extra), capture_output=True, text=True)
    assert p.returncode == expect, p.stderr
    return out, p


def hashes_of(seq: str, k, w):
    r = oracle.to_ranks(seq.encode())
    if len(r) < k:
        return np.zeros(0, np.uint64)
    return oracle.minimiser_hash(r, k, min(w, len(r)))  # a range shorter than the window: the window shrinks to it


def check_filter(hip, path, seqs, names, k, w, h_req, max_fp, filter_size, mode="avg", min_length=0):
    """validate_filter + validate_elements of the reference's tests, then the stronger checks against the oracle"""
    from ganon_amd import ibf_file
    m = ibf_file.read_ibf_meta(path)
    cfg = m.config
    # validate_filter (GanonBuild.test.cpp:23-47)
    assert m.bins == len(m.bin_map) == cfg["n_bins"]
    assert m.hash_funs == cfg["hash_functions"]
    if h_req > 0:
        assert m.hash_funs == h_req
    if not filter_size:
        assert math.floor(cfg["true_max_fp"] * 100.0) / 100.0 <= math.floor(max_fp * 100.0) / 100.0
        assert math.floor(cfg["true_avg_fp"] * 100.0) / 100.0 <= math.floor(max_fp * 100.0) / 100.0
    # targets in first-appearance order, their hash sets (per file distinct, files behind each other)
    order, per_target = [], {}
    for s, t in zip(seqs, names):
        if t not in per_target:
            per_target[t] = []
            order.append(t)
        if len(s) >= min_length:
            per_target[t].append(np.unique(hashes_of(s, k, w)))
    counts = [int(sum(len(x) for x in per_target[t])) for t in order]
    assert m.hashes_count == list(zip(order, counts))
    # the header is what the oracle's restatement of optimal_hashes / true_false_positive computes from those counts
    exp = bp.optimal_hashes(0.0 if filter_size else max_fp, filter_size, counts, h_req, mode)
    exp.true_max_fp, exp.true_avg_fp = bp.true_false_positive(counts, exp.max_hashes_bin, exp.bin_size_bits, exp.hash_functions)
    for key in ("n_bins", "max_hashes_bin", "hash_functions", "bin_size_bits", "max_fp", "true_max_fp", "true_avg_fp"):
        assert cfg[key] == getattr(exp, key), (key, cfg[key], getattr(exp, key))
    assert (cfg["kmer_size"], cfg["window_size"]) == (k, w)
    # bin map and bits: an oracle filter filled with the same layout (create_bin_map_hash) is the file's payload
    spans = bp.create_bin_map(exp.max_hashes_bin, counts)
    assert m.bin_map == [(b, order[t]) for b, (t, _, _) in enumerate(spans)]
    ref = oracle.Ibf(m.bins, m.bin_size, m.hash_funs)
    for b, (t, a, z) in enumerate(spans):
        ref.emplace_many(np.concatenate(per_target[order[t]])[a:z + 1], b)
    payload = np.fromfile(path, dtype=np.uint64, offset=m.payload_offset).reshape(m.bin_size, m.bin_words)
    assert np.array_equal(payload, ref.data)
    # validate_elements (:49-85): every sequence finds all its minimisers in its target's bins -- through the device
    flt, _ = ibf_file.load_ibf(path)
    tb = {}
    for b, t in m.bin_map:
        tb.setdefault(t, []).append(b)
    ok = [(s, t) for s, t in zip(seqs, names) if len(s) >= max(min_length, w)]
    bases = np.frombuffer("".join(s for s, _ in ok).encode(), dtype=np.uint8)
    off = np.zeros(len(ok) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s, _ in ok])
    st = hip.HipStream(flt, len(ok), bases.size)
    st.submit(bases, off, None, k, w, 0.0)
    nh, status, _, _ = st.fetch()
    dense = st.dense_counts(0, len(ok), m.bins)
    for i, (s, t) in enumerate(ok):
        assert status[i] == 0 and nh[i] == len(hashes_of(s, k, w))
        found = int(dense[i][tb[t]].astype(np.int64).sum())
        # one bin: exactly its minimisers (:80-82); a target split over several bins can see a hash again in a sibling bin as
        # a false positive (the reference only runs this check on one-bin targets)
        good = found == nh[i] if len(tb[t]) == 1 else nh[i] <= found <= nh[i] * len(tb[t])
        if not good:  # (seen twice as the first GPU work on a fresh box, never again in the same session: say as much as possible)
            again = st.dense_counts(0, len(ok), m.bins)
            ho, hs = st.fetch_hashes()
            exp_h = hashes_of(s, k, w)
            got_h = hs[int(ho[i]):int(ho[i + 1])]
            rows_dev = flt.download_rows(0, 0, min(4, m.bin_size))
            rows_file = np.fromfile(path, dtype=np.uint64, offset=m.payload_offset, count=min(4, m.bin_size) * m.bin_words).reshape(-1, m.bin_words)
            raise AssertionError(dict(read=i, target=t, found=found, n_hashes=int(nh[i]), bins=len(tb[t]), dense_all_zero=bool((dense == 0).all()),
                                      dense_row_sum=int(dense[i].astype(np.int64).sum()), second_dense_equal=bool(np.array_equal(dense, again)),
                                      second_found=int(again[i][tb[t]].astype(np.int64).sum()), hashes_equal_oracle=bool(np.array_equal(got_h, exp_h)),
                                      first_rows_on_device_equal_file=bool(np.array_equal(rows_dev, rows_file))))
    st.destroy()
    flt.free()
    return m


def test_verbose_writes_a_log(hip, tmp_path):
    inp, _, names = write_inputs(str(tmp_path), SEQS)
    out, p = run_build(str(tmp_path), inp, extra=["--verbose"])
    p = subprocess.run([BIN_BUILD, "-i", inp, "-o", out, "-k", "19", "-w", "32", "-s", "4", "--verbose"], capture_output=True, text=True)
    assert p.returncode == 0
    for piece in ("--input-file        ", "ibf_config:", "n_bins         ", "Filter size: ", "Count/save hashes start: ",
                  "ganon-build processed 10 sequences / 10 files (0.0008 Mbp) in ", " - max. false positive: ", " - filter size: "):
        assert piece in p.stderr, piece
    check_filter(hip, out, SEQS, names, 19, 32, 4, 0.05, 0)


def test_input_file_one_and_two_columns(hip, tmp_path):
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(), d2.mkdir()
    inp, _, names = write_inputs(str(d1), SEQS)
    out, _ = run_build(str(d1), inp)
    check_filter(hip, out, SEQS, names, 19, 32, 4, 0.05, 0)
    targets = ["T1", "T9", "T1", "T8", "T1", "T1", "T1", "T1", "T4", "T1"]  # :186
    inp, _, names = write_inputs(str(d2), SEQS, targets)
    out, _ = run_build(str(d2), inp)
    m = check_filter(hip, out, SEQS, names, 19, 32, 4, 0.05, 0)
    assert [t for t, _ in m.hashes_count] == ["T1", "T9", "T8", "T4"]


def test_max_fp_and_filter_size(hip, tmp_path):
    sizes = {}
    for tag, kw in (("fp0.01", dict(max_fp=0.01)), ("fp0.5", dict(max_fp=0.5))):
        d = tmp_path / tag
        d.mkdir()
        inp, _, names = write_inputs(str(d), SEQS)
        out, _ = run_build(str(d), inp, **kw)
        check_filter(hip, out, SEQS, names, 19, 32, 4, kw["max_fp"], 0)
        sizes[tag] = os.path.getsize(out)
    assert sizes["fp0.01"] > sizes["fp0.5"]  # :259
    for tag, fs in (("fs0.1", 0.1), ("fs1", 1.0)):
        d = tmp_path / tag
        d.mkdir()
        inp, _, names = write_inputs(str(d), SEQS)
        out, _ = run_build(str(d), inp, extra=["--filter-size", repr(fs)])
        check_filter(hip, out, SEQS, names, 19, 32, 4, 0.05, fs)
        sizes[tag] = os.path.getsize(out)
    assert sizes["fs0.1"] < sizes["fs1"]  # :287


def test_filter_size_too_small_for_the_targets_fails_cleanly(hip, tmp_path):
    # a --filter-size that leaves no bit per bin: the reference dies in the seqan3 IBF constructor, this build says why
    d = tmp_path / "tiny"
    d.mkdir()
    inp, _, names = write_inputs(str(d), SEQS)
    p = subprocess.run([BIN_BUILD, "-i", inp, "-o", str(d / "x.ibf"), "--filter-size", "1e-9", "--max-fp", "0"], capture_output=True, text=True)
    assert p.returncode == 1 and ("bits per bin" in p.stderr or "filter" in p.stderr.lower()), p.stderr
    assert not (d / "x.ibf").exists()


def test_modes_on_the_reference_data_set(hip, tmp_path):
    # :290-352 with the reference's own mode_input.tsv and 25 gzipped genomes
    from ganon_amd import ibf_file
    inp = str(tmp_path / "mode_input.tsv")
    seqs, names = [], []
    with open(inp, "w") as o:
        for line in open(os.path.join(DATA, "mode_input.tsv")):
            f, t = line.rstrip("\n").split("\t")
            o.write(f"{os.path.join(DATA, f)}\t{t}\n")
            for s in read_fasta_gz(os.path.join(DATA, f)):
                seqs.append(s)
                names.append(t)

    def build(tag, extra, **kw):
        d = tmp_path / tag
        d.mkdir()
        out, _ = run_build(str(d), inp, extra=extra, **kw)
        return out

    avg = build("avg_fp", ["--mode", "avg"], max_fp=0.001)
    smallest = build("smallest_fp", ["--mode", "smallest"])  # (the reference's test leaves this one at 0.05, :311-313)
    check_filter(hip, avg, seqs, names, 19, 32, 4, 0.001, 0, "avg")
    check_filter(hip, smallest, seqs, names, 19, 32, 4, 0.05, 0, "smallest")
    assert os.path.getsize(smallest) < os.path.getsize(avg)
    metas = {}
    for mode in ("avg", "smallest", "fastest"):
        out = build(mode + "_fs", ["--mode", mode, "--filter-size", "1"])
        metas[mode] = check_filter(hip, out, seqs, names, 19, 32, 4, 0.05, 1.0, mode)
    assert metas["smallest"].config["max_fp"] < metas["avg"].config["max_fp"]
    assert metas["fastest"].config["n_bins"] < metas["avg"].config["n_bins"]


@pytest.mark.parametrize("h", [0, 2])
def test_hash_functions(hip, tmp_path, h):
    inp, _, names = write_inputs(str(tmp_path), SEQS)
    out, _ = run_build(str(tmp_path), inp, h=h)
    check_filter(hip, out, SEQS, names, 19, 32, h, 0.05, 0)


@pytest.mark.parametrize("k,w", [(19, 32), (21, 23), (27, 27)])
def test_window_and_kmer_sizes(hip, tmp_path, k, w):
    inp, _, names = write_inputs(str(tmp_path), SEQS)
    out, _ = run_build(str(tmp_path), inp, k=k, w=w)
    check_filter(hip, out, SEQS, names, k, w, 4, 0.05, 0)


def test_tmp_output_folder(hip, tmp_path):
    inp, _, names = write_inputs(str(tmp_path), SEQS)
    existing = tmp_path / "tmp_existing"
    existing.mkdir()
    out, _ = run_build(str(tmp_path), inp, extra=["--tmp-output-folder", str(existing) + "/"])
    check_filter(hip, out, SEQS, names, 19, 32, 4, 0.05, 0)
    os.remove(out)
    run_build(str(tmp_path), inp, extra=["--tmp-output-folder", str(tmp_path / "missing") + "/"], expect=1)
    assert not os.path.exists(out)


def test_min_length(hip, tmp_path):
    for ml in (0, 50):
        d = tmp_path / f"ml{ml}"
        d.mkdir()
        inp, _, names = write_inputs(str(d), SEQS2)
        out, p = run_build(str(d), inp, extra=["--min-length", str(ml)])
        m = check_filter(hip, out, SEQS2, names, 19, 32, 4, 0.05, 0, min_length=ml)
        kept = sum(1 for s in SEQS2 if len(s) >= ml)
        assert sum(1 for _, c in m.hashes_count if c > 0) == kept == (10 if ml == 0 else 7)  # :538-546


def test_long_multi_record_and_short_sequences(hip, tmp_path):
    # what the reference's tests do not reach: sequences far longer than a device piece (every window must still be seen
    # exactly as in the unsplit sequence), several records and several files per target (per-file sets are concatenated,
    # :236-238), gzip, bzip2, IUPAC letters, sequences shorter than the window, an unreadable file and a missing one
    rng = np.random.default_rng(3)
    k, w = 19, 31
    d = str(tmp_path)

    def rnd(n):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, size=n))

    big = rnd(300_000)
    shared = rnd(5000)
    files = {"g1.fasta": [("a", big[:150_000] + "NNNNRYKM" + big[150_000:])], "g2.fa.gz": [("b1", rnd(20_000)), ("b2", shared), ("b3", rnd(25))],
             "g3.fasta.bz2": [("c1", shared), ("c2", rnd(30)), ("c3", rnd(18))], "g4.fasta": [("d", rnd(2000))]}
    for name, recs in files.items():
        text = "".join(f">{i}\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n" for i, s in recs)
        if name.endswith(".gz"):
            gzip.open(os.path.join(d, name), "wt").write(text)
        elif name.endswith(".bz2"):
            import bz2
            bz2.open(os.path.join(d, name), "wt").write(text)
        else:
            open(os.path.join(d, name), "w").write(text)
    open(os.path.join(d, "bad.fasta"), "w").write(">x\nACGTACGTACGT!!ACGT\n")
    inp = os.path.join(d, "in.tsv")
    with open(inp, "w") as o:
        o.write(f"{d}/g1.fasta\tBIG\n{d}/g2.fa.gz\tMIX\n{d}/g3.fasta.bz2\tMIX\n{d}/missing.fasta\tGONE\n{d}/bad.fasta\tBAD\n{d}/g4.fasta\n")
    out = os.path.join(d, "x.ibf")
    p = subprocess.run([BIN_BUILD, "-i", inp, "-o", out, "-k", str(k), "-w", str(w), "-s", "3", "-p", "0.01", "-t", "3"], capture_output=True,
                       text=True)
    assert p.returncode == 0, p.stderr
    assert "WARNING: input file not found/empty: " in p.stderr and "Error parsing file [" in p.stderr
    assert " sequences / 6 files (" in p.stderr and " - 1 invalid files skipped" in p.stderr
    from ganon_amd import ibf_file
    m = ibf_file.read_ibf_meta(out)
    exp = {"BIG": [files["g1.fasta"]], "MIX": [files["g2.fa.gz"], files["g3.fasta.bz2"]], "BAD": [], "g4.fasta": [files["g4.fasta"]]}
    got = dict(m.hashes_count)
    assert list(got) == ["BIG", "MIX", "BAD", "g4.fasta"]
    sets = {}
    for t, per_file in exp.items():
        parts = []
        for recs in per_file:
            hs = [hashes_of(s, k, w) for _, s in recs]
            parts.append(np.unique(np.concatenate(hs)) if hs else np.zeros(0, np.uint64))
        sets[t] = parts
        assert got[t] == sum(len(x) for x in parts), t
    assert got["MIX"] > len(np.unique(np.concatenate(sets["MIX"])))  # the shared record is counted once per file
    # every hash of every target is found in that target's bins, nothing else of a random probe is (beyond the fp rate)
    flt, _ = ibf_file.load_ibf(out)
    ref = oracle.Ibf(m.bins, m.bin_size, m.hash_funs)
    counts = [got[t] for t in got]
    spans = bp.create_bin_map(m.config["max_hashes_bin"], counts)
    order = list(got)
    for b, (t, a, z) in enumerate(spans):
        ref.emplace_many(np.concatenate(sets[order[t]])[a:z + 1], b)
    payload = np.fromfile(out, dtype=np.uint64, offset=m.payload_offset).reshape(m.bin_size, m.bin_words)
    assert np.array_equal(payload, ref.data)
    flt.free()


def test_distinct_hashes_of_overlapping_pieces(hip):
    # the ABI call underneath: pieces that share w-1 bases have the hash SET of the whole sequence
    rng = np.random.default_rng(9)
    for k, w, stride in ((19, 31, 512), (21, 21, 512), (31, 200, 4096), (4, 4, 64)):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=70_000))
        pieces, at = [], 0
        while len(seq) - at >= w:
            pieces.append(seq[at:at + stride + w - 1])
            at += stride
        bases = np.frombuffer(b"".join(pieces), dtype=np.uint8)
        off = np.zeros(len(pieces) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in pieces])
        tmp = hip.HipFilter.ibf(None, 64, 64, 1)
        st = hip.HipStream(tmp, len(pieces), bases.size)
        st.upload(bases, off, None)
        st.minimisers(k, w)
        got = st.distinct_hashes()
        exp = np.unique(oracle.minimiser_hash(oracle.to_ranks(seq), k, w))
        assert np.array_equal(got, exp), (k, w)
        st.destroy()
        tmp.free()


def test_built_filter_classifies_with_the_binary(hip, tmp_path):
    # ganon-build -> ganon-classify: reads cut from the reference's 25 genomes go to their genome
    from test_build_cpu import DATA
    import cli_util as cu
    inp = str(tmp_path / "in.tsv")
    genomes = {}
    with open(inp, "w") as o:
        for line in open(os.path.join(DATA, "mode_input.tsv")):
            f, t = line.rstrip("\n").split("\t")
            o.write(f"{os.path.join(DATA, f)}\tG{t}\n")
            genomes[f"G{t}"] = "".join(read_fasta_gz(os.path.join(DATA, f)))
    out = str(tmp_path / "db.ibf")
    p = subprocess.run([BIN_BUILD, "-i", inp, "-o", out, "--quiet", "-p", "0.01", "-t", "4"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rng = np.random.default_rng(4)
    fq = str(tmp_path / "reads.fq")
    truth = {}
    with open(fq, "w") as o:
        for i in range(2000):
            t = f"G{i % 25}"
            g = genomes[t]
            s = int(rng.integers(0, len(g) - 150))
            o.write(f"@r{i}\n{g[s:s + 150]}\n+\n{'I' * 150}\n")
            truth[f"r{i}"] = t
    prefix = str(tmp_path / "res")
    cu.run(cu.BIN_HIP, ["--ibf", out, "--single-reads", fq, "-o", prefix, "--output-all", "--quiet", "--rel-cutoff", "0.9", "--skip-lca"])
    best = {}
    for line in open(prefix + ".all"):
        rid, t, c = line.rstrip("\n").split("\t")
        if rid not in best or int(c) > best[rid][1]:
            best[rid] = (t, int(c))
    assert len(best) == 2000
    assert all(best[r][0] == truth[r] for r in truth)


def test_nothing_to_build_and_all_short(hip, tmp_path):
    # every sequence below --min-length / below one k-mer: "No valid sequences to build", exit code 1, no file (:804-808)
    inp, _, _ = write_inputs(str(tmp_path), [s[:15] for s in SEQS[:3]])
    out, p = run_build(str(tmp_path), inp, expect=1)
    assert not os.path.exists(out)
    p = subprocess.run([BIN_BUILD, "-i", inp, "-o", out, "-k", "19", "-w", "32"], capture_output=True, text=True)
    assert p.returncode == 1 and "No valid sequences to build" in p.stderr
    inp, _, _ = write_inputs(str(tmp_path), SEQS)
    out, p = run_build(str(tmp_path), inp, extra=["--min-length", "1000"], expect=1)
    assert not os.path.exists(out)
    # an input file that lists nothing usable
    empty = tmp_path / "none.tsv"
    empty.write_text(f"{tmp_path}/does_not_exist.fasta\tT\n")
    p = subprocess.run([BIN_BUILD, "-i", str(empty), "-o", out], capture_output=True, text=True)
    assert p.returncode == 1 and "No valid input files" in p.stderr


def test_more_bins_than_one_classify_filter_takes(hip, tmp_path):
    # 33 000 one-bin targets: the builder's filter is storage-only (no bin-count limit); ganon-classify loads the file as
    # column parts and finds every target's own sequence
    rng = np.random.default_rng(12)
    n = 33000
    d = str(tmp_path)
    seqs = rng.integers(0, 4, size=(n, 120), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(os.path.join(d, "in.tsv"), "w") as tsv:
        for i in range(n):
            f = os.path.join(d, f"t{i}.fa")
            with open(f, "wb") as o:
                o.write(b">s\n" + lut[seqs[i]].tobytes() + b"\n")
            tsv.write(f"{f}\tT{i}\n")
    out = os.path.join(d, "wide.ibf")
    p = subprocess.run([BIN_BUILD, "-i", os.path.join(d, "in.tsv"), "-o", out, "-t", "16", "--quiet", "-p", "0.001", "-s", "3"], capture_output=True,
                       text=True)
    assert p.returncode == 0, p.stderr
    from ganon_amd import ibf_file
    m = ibf_file.read_ibf_meta(out)
    assert m.bins == n and m.bin_words == (n + 63) // 64
    import cli_util as cu
    fq = os.path.join(d, "r.fq")
    pick = rng.integers(0, n, size=500)
    with open(fq, "w") as o:
        for j, i in enumerate(pick):
            o.write(f"@r{j}\n{lut[seqs[i]].tobytes().decode()}\n+\n{'I' * 120}\n")
    prefix = os.path.join(d, "res")
    cu.run(cu.BIN_HIP, ["--ibf", out, "--single-reads", fq, "-o", prefix, "--output-all", "--quiet", "--rel-cutoff", "1", "--skip-lca"])
    hits = {}
    for line in open(prefix + ".all"):
        rid, t, c = line.rstrip("\n").split("\t")
        hits.setdefault(rid, set()).add(t)
    assert all(f"T{i}" in hits.get(f"r{j}", ()) for j, i in enumerate(pick))
