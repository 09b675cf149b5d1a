"""ganon_amd.ibf_file: Python-side .ibf reader/writer and mini build."""
import os

import numpy as np
import pytest

import cli_util as cu
import ganon_fixtures as gf
import gpu_util as gu
import oracle


def _built(seed=3, n_targets=9, glen=1200):
    rng = np.random.default_rng(seed)
    genomes = {f"T{i}": "".join("ACGT"[x] for x in rng.integers(0, 4, size=glen)) for i in range(n_targets)}
    return gf.build_ibf(genomes, 19, 31, max_fp=0.01, hash_functions=3), genomes, rng


@pytest.mark.parametrize("bv_header", ["wgb", "b", "wgq"])
def test_read_ibf_meta_matches_fixture_writer(tmp_path, bv_header):
    from ganon_amd import ibf_file
    built, _, _ = _built()
    path = str(tmp_path / "a.ibf")
    gf.write_ibf(path, built, bv_header=bv_header)
    m = ibf_file.read_ibf_meta(path)
    assert m.config["kmer_size"] == 19 and m.config["window_size"] == 31 and m.config["n_bins"] == built.ibf.bins
    assert (m.bins, m.bin_size, m.hash_funs, m.bin_words) == (built.ibf.bins, built.ibf.bin_size, built.ibf.hash_funs, built.ibf.bin_words)
    assert m.hashes_count == built.hashes_count and m.bin_map == built.bin_map
    assert m.payload_offset + m.payload_bytes == os.path.getsize(path)
    raw = np.fromfile(path, dtype="<u8", offset=m.payload_offset).reshape(m.bin_size, m.bin_words)
    assert np.array_equal(raw, built.ibf.data)
    open(path, "r+b").truncate(os.path.getsize(path) - 8)
    with pytest.raises(ibf_file.IbfFormatError):
        ibf_file.read_ibf_meta(path)


@pytest.mark.gpu
def test_load_ibf_whole_and_column_slices(tmp_path):
    import ganon_amd
    from ganon_amd import ibf_file
    from ganon_amd import partition as gp
    built, genomes, rng = _built(seed=5, n_targets=40, glen=3000)
    path = str(tmp_path / "b.ibf")
    gf.write_ibf(path, built)
    flt, m = ibf_file.load_ibf(path, chunk_bytes=1 << 16)   # many chunks
    assert np.array_equal(flt.download_rows(0, m.bin_size, m.bin_words), built.ibf.data)
    seqs = [g[100:250].encode() for g in list(genomes.values())[:20]] + [gu.random_seq(rng, 150) for _ in range(20)]
    bases, off1, _ = gu.pack_reads(seqs, None)
    st = ganon_amd.HipStream(flt, len(seqs), bases.size)
    st.submit(bases, off1, None, 19, 31, 0.5)
    nh, status, mo, full = st.fetch()
    assert len(full) >= 20
    names, b2t = m.targets()
    # every rank loads only its own columns from the file; the union of the slices' matches == the whole filter's
    for world in (2, 3):
        parts = []
        for sl in gp.plan_partition(b2t, m.bins, world):
            f2, _ = ibf_file.load_ibf(path, word_lo=sl.word_lo, word_hi=sl.word_hi, bin2target=sl.bin2target_local,
                                      n_targets=max(1, len(sl.targets_global)), chunk_bytes=1 << 15)
            exp = built.ibf.data[:, sl.word_lo:sl.word_hi].copy()
            if sl.bins_local & 63:
                exp[:, -1] &= np.uint64((1 << (sl.bins_local & 63)) - 1)
            assert np.array_equal(f2.download_rows(0, m.bin_size, sl.word_hi - sl.word_lo), exp)
            loc = gp.HipLocalFilter(f2, 0, own=True)
            _, _, _, mm = loc.classify(bases, off1, None, 19, 31, 0.5).fetch()
            g = mm.copy()
            if len(g):
                g["target"] = sl.targets_global[mm["target"]]
            parts.append(g)
            loc.close()
        got = np.concatenate(parts)
        got = got[np.lexsort((got["target"], got["read"]))]
        assert np.array_equal(got, full), world
    st.destroy()
    flt.free()


@pytest.mark.gpu
def test_device_build_save_and_classify_with_the_binary(tmp_path):
    # mini ganon-build on the device -> .ibf on disk -> the C++ binary loads it (streaming loader) and classifies; the
    # oracle re-derives the expected matches from the saved bits
    from ganon_amd import ibf_file
    rng = np.random.default_rng(8)
    genomes = {f"G{i}": [gu.random_seq(rng, 2500), gu.random_seq(rng, 800)] for i in range(12)}
    flt, kw = ibf_file.build_ibf(genomes, 19, 31, max_fp=0.01, hash_funs=3, max_hashes_bin=150)  # forces split bins
    path = str(tmp_path / "dev.ibf")
    ibf_file.save_ibf(path, flt, **kw)
    m = ibf_file.read_ibf_meta(path)
    assert m.bins == kw["bins"] > 12 and m.hashes_count == kw["hashes_count"]
    bits = flt.download_rows(0, m.bin_size, m.bin_words)
    assert np.array_equal(np.fromfile(path, dtype="<u8", offset=m.payload_offset).reshape(m.bin_size, m.bin_words), bits)
    flt.free()
    # no false negatives: every minimiser of a genome is found in one of its target's bins (GanonBuild.test.cpp:53-98)
    ibf = oracle.Ibf(m.bins, m.bin_size, m.hash_funs, bits)
    names, b2t = m.targets()
    for t, seqs in genomes.items():
        ti = names.index(t)
        for s in seqs:
            for hv in np.unique(oracle.minimiser_hash(oracle.to_ranks(s), 19, 31)).tolist():
                c = ibf.bulk_count(np.array([hv], dtype=np.uint64))
                assert c[b2t == ti].sum() >= 1
    recs = [(f"r{i}", genomes[f"G{i % 12}"][0][50 + i:200 + i].decode()) for i in range(60)] + \
           [(f"x{i}", gu.random_seq(rng, 150).decode()) for i in range(40)]
    fq = str(tmp_path / "r.fq")
    gf.write_fastq(fq, recs)
    out = str(tmp_path / "o")
    cu.run(cu.BIN_HIP, ["--ibf", path, "--single-reads", fq, "-o", out, "--output-all", "--rel-cutoff", "0.6", "--quiet"])
    res = cu.Res(out, lca_file=False, unc_file=False)
    for rid, s in recs:
        hh = oracle.minimiser_hash(oracle.to_ranks(s), 19, 31)
        exp, _ = gu.oracle_matches(ibf, b2t, len(names), hh, 0.6)
        assert res.all.get(rid, {}) == {names[t]: c for t, c in exp}, rid
    assert res.total_classified >= 60
