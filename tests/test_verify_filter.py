"""`ganon-classify --ibf F --verify-filter refs.tsv` (host/verify.cpp, gn_filter_probe): the membership check of the reference's
build test (validate_elements, /root/reference/tests/ganon-build/GanonBuild.test.cpp:53-98) against a filter FILE, on the
device.  Here against files of this repo's ganon-build, intact and deliberately wrong; scripts/first_contact.sh runs the same
command on a file written by the reference's ganon-build."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cli_util as cu
import gpu_util as gu
import ganon_fixtures as gf
import oracle
from test_build_gpu import SEQS, hip, run_build, write_inputs  # noqa: F401 (hip is a fixture)

pytestmark = pytest.mark.gpu


def _verify(ibf, tsv, *extra):
    p = subprocess.run([cu.BIN_HIP, "--ibf", ibf, "--verify-filter", tsv, *extra], capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


def test_filter_of_our_builder_passes_and_wrong_ones_fail(hip, tmp_path):
    rng = np.random.default_rng(11)
    seqs = [gu.random_seq(rng, int(rng.integers(2000, 30000))).decode() for _ in range(12)] + SEQS[:3]
    targets = [f"T{i // 2}" for i in range(len(seqs))]            # two files per target; large ones are split over several bins
    inp, files, names = write_inputs(str(tmp_path), seqs, targets)
    ibf, _ = run_build(str(tmp_path), inp, k=19, w=32, h=4, max_fp=0.01)
    rc, out, err = _verify(ibf, inp)
    assert rc == 0, out + err
    rows = [ln.split("\t") for ln in out.splitlines() if not ln.startswith(("#", "filter", "result", " "))]
    assert len(rows) == len(seqs) and all(r[-1] == "ok" and r[5] == "0" for r in rows)
    # distinct hashes per file == the oracle's, and hits >= distinct (a hash may sit in two of the target's bins by chance)
    for r, s in zip(rows, seqs):
        exp = len(np.unique(oracle.minimiser_hash(oracle.to_ranks(s.encode()), 19, 32)))
        assert int(r[3]) == exp and int(r[4]) >= exp
    assert out.splitlines()[-1].startswith("result\tok")

    # (1) the targets swapped in the list: the sequences are looked up in somebody else's bins
    swapped = str(tmp_path / "swapped.tsv")
    with open(swapped, "w") as o:
        for f, t in zip(files, targets[2:] + targets[:2]):
            o.write(f"{f}\t{t}\n")
    rc, out, _ = _verify(ibf, swapped)
    assert rc == 1 and "FAIL" in out and "first false negative: hash" in out and out.splitlines()[-1].startswith("result\tFAIL")
    # (2) a target the filter does not know
    unknown = str(tmp_path / "unknown.tsv")
    open(unknown, "w").write(f"{files[0]}\tnot_there\n")
    rc, out, _ = _verify(ibf, unknown)
    assert rc == 1 and "no such target" in out
    # (3) the file claims another k: other hashes, looked up in vain.  (A LARGER window would pass: the minimisers of a larger window are a
    # subset of the smaller window's; a smaller one fails like a smaller k.)
    for at, fmt, was, now in ((29, "<B", 19, 17), (30, "<H", 32, 24)):
        data = bytearray(open(ibf, "rb").read())
        assert struct.unpack_from(fmt, data, at)[0] == was
        struct.pack_into(fmt, data, at, now)
        wrong = str(tmp_path / f"wrong_{at}.ibf")
        open(wrong, "wb").write(bytes(data))
        rc, out, _ = _verify(wrong, inp)
        assert rc == 1 and "first false negative" in out, (at, out[-500:])
    # (4) one payload byte cleared where a hash of the first file lives: exactly the files that own the bit fail
    from ganon_amd import ibf_file
    m = ibf_file.read_ibf_meta(ibf)
    hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(seqs[0].encode()), 19, 32))
    row = int(oracle.SampledIbf(m.bins, m.bin_size, m.hash_funs, None).rows_of(hs[:1])[0])
    data = bytearray(open(ibf, "rb").read())
    at = m.payload_offset + row * m.bin_words * 8
    data[at:at + m.bin_words * 8] = bytes(m.bin_words * 8)
    cleared = str(tmp_path / "cleared.ibf")
    open(cleared, "wb").write(bytes(data))
    rc, out, _ = _verify(cleared, inp)
    assert rc == 1 and f"first false negative: hash {int(hs[0])} (index 0" in out and f"rows {row} " in out
    bad = [ln.split("\t") for ln in out.splitlines() if ln.endswith("FAIL") and not ln.startswith("result")]
    assert bad and bad[0][0] == "T0" and bad[0][1] == files[0]


def test_probe_abi_equals_the_oracle(hip):
    # gn_filter_probe == per-hash membership of the oracle's IBF: hits summed over the bins, misses, the first miss
    rng = np.random.default_rng(3)
    bins, rows, h = 200, 1009, 3
    ibf = gf.random_ibf(bins, rows, h, 0.2, seed=8)
    hashes = rng.integers(0, 1 << 38, size=5000, dtype=np.uint64)
    own = np.array([5, 64, 65, 199], dtype=np.uint32)
    for i, v in enumerate(hashes[:3000]):
        ibf.emplace_many(np.array([v], dtype=np.uint64), int(own[i % 4]))
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    for sel, lst in ((slice(0, 3000), own), (slice(0, 5000), own), (slice(2990, 5000), own[:1]), (slice(0, 0), own), (slice(0, 10), own[:0])):
        hh = hashes[sel]
        per = np.stack([ibf.bulk_count(hh[i:i + 1])[lst.astype(np.int64)] for i in range(len(hh))]) if len(hh) else np.zeros((0, len(lst)))
        miss = np.flatnonzero(per.sum(axis=1) == 0) if len(hh) else np.zeros(0, dtype=np.int64)
        hits, missing, first = flt.probe(hh, lst)
        assert (hits, missing, first) == (int(per.sum()), len(miss), int(miss[0]) if len(miss) else -1)
    with pytest.raises(hip.GanonHipError):
        flt.probe(hashes[:4], np.array([bins], dtype=np.uint32))
    flt.free()


def test_verify_filter_refuses_what_it_cannot_check(hip, tmp_path):
    p = subprocess.run([cu.BIN_HIP, "--ibf", "a.hibf", "--hibf", "--verify-filter", "x.tsv"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "flat .ibf" in p.stderr
    p = subprocess.run([cu.BIN_HIP, "--ibf", "a.ibf,b.ibf", "--verify-filter", "x.tsv"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "exactly one" in p.stderr
