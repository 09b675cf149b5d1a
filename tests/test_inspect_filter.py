"""`ganon-classify --inspect-filter F [--hibf]` (filter_io.cpp): the first-contact tool for filter FILES -- parse only, no device.
Every header field is printed with its offset, every redundancy is checked (S*W*8 against the bytes left, hash_shift ==
countl_zero(bin_size), technical_bins == 64*W, bit_vector size, map sizes), exit 1 names the first inconsistent field.
Exercised on files of this repo's writers -- every bit_vector header variant the loader accepts -- and on files damaged in
six different places; a file written by SeqAn3 / raptor would be run through exactly this (scripts/first_contact.sh)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cli_util as cu
import ganon_fixtures as gf
import oracle


@pytest.fixture(scope="module")
def binary():
    return cu.build_oracle_binary()      # same main.cpp / cli.cpp / filter_io.cpp as the product binary; needs no HIP runtime


@pytest.fixture(scope="module")
def built():
    rng = np.random.default_rng(5)
    targets = {f"t{i}.1": ["".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(300, 900))))] for i in range(7)}
    return gf.build_ibf(targets, k=19, w=31, max_fp=0.01)


def _inspect(binary, path, *extra):
    p = subprocess.run([binary, "--inspect-filter", path, *extra], capture_output=True, text=True, timeout=60)
    return p.returncode, p.stdout, p.stderr


def _field(out: str, name: str) -> str:
    for ln in out.splitlines():
        if ln.startswith("@") and ln[12:].startswith(name):
            return ln.split(" = ", 1)[1].split("   ")[0].strip()
    raise AssertionError(f"no field {name!r} in:\n{out}")


@pytest.mark.parametrize("bv_header", ["wgb", "b", "wb", "gb", "wgq", "q"])
def test_inspect_prints_every_field_of_our_own_ibf(binary, built, tmp_path, bv_header):
    path = str(tmp_path / "f.ibf")
    gf.write_ibf(path, built, bv_header=bv_header)
    rc, out, err = _inspect(binary, path)
    assert rc == 0, out + err
    ibf = built.ibf
    assert _field(out, "version") == "2.1.1"
    assert int(_field(out, "IBFConfig.n_bins")) == ibf.bins
    assert int(_field(out, "IBFConfig.kmer_size")) == 19 and int(_field(out, "IBFConfig.window_size")) == 31
    assert int(_field(out, "ibf.bins")) == ibf.bins and int(_field(out, "ibf.technical_bins")) == 64 * ibf.bin_words
    assert int(_field(out, "ibf.bin_size (rows S)")) == ibf.bin_size and int(_field(out, "ibf.hash_funs")) == ibf.hash_funs
    assert int(_field(out, "ibf.hash_shift")) == 64 - int(ibf.bin_size).bit_length()
    assert int(_field(out, "hashes_count")) == 7 and int(_field(out, "bin_map")) == ibf.bins
    payload = ibf.bin_size * ibf.bin_words * 8
    assert int(_field(out, "payload")) == payload and f"@{os.path.getsize(path) - payload}" in out
    assert "MISMATCH" not in out and out.count(" ok") >= 8
    assert out.splitlines()[-1].startswith("result      CONSISTENT: IBF, k=19 w=31, 1 IBF(s)")
    # the header variant is named
    how = _field(out, "sdsl bit_vector header")
    assert ("width byte" in how) == ("w" in bv_header) and ("growth factor" in how) == ("g" in bv_header)
    assert how.endswith("64-bit words" if "q" in bv_header else "bits")


def _damage(data: bytearray, what: str, built) -> bytearray:
    ibf = built.ibf
    # offsets of the fixed-size head: version 12 bytes, then IBFConfig
    if what == "truncated_payload":
        return data[:-100]
    if what == "trailing_bytes":
        return data + b"\0" * 16
    if what == "n_bins":                       # IBFConfig.n_bins disagrees with the stored IBF
        struct.pack_into("<Q", data, 12, ibf.bins + 1)
        return data
    if what == "kmer_size":
        data[12 + 17] = 40                     # IBFConfig.kmer_size > 32
        return data
    # the six IBF fields sit right before the bit_vector header: find them from the end
    payload = ibf.bin_size * ibf.bin_words * 8
    hdr = len(data) - payload - 13 - 48        # "wgb": 13 header bytes
    assert struct.unpack_from("<Q", data, hdr)[0] == ibf.bins
    if what == "hash_shift":
        struct.pack_into("<Q", data, hdr + 24, 63 - int(ibf.bin_size).bit_length())
    elif what == "technical_bins":
        struct.pack_into("<Q", data, hdr + 8, 64 * ibf.bin_words + 64)
    elif what == "bit_vector_size":
        struct.pack_into("<Q", data, hdr + 48 + 5, 12345)
    elif what == "string_length":              # first hashes_count string length runs past the file
        struct.pack_into("<Q", data, 12 + 52 + 8, 1 << 40)
    return data


@pytest.mark.parametrize("what,names", [("truncated_payload", ["truncated", "payload"]), ("trailing_bytes", ["bit_vector header"]),
                                        ("n_bins", ["IBFConfig"]), ("kmer_size", ["k/w"]), ("hash_shift", ["hash_shift"]),
                                        ("technical_bins", ["technical_bins"]), ("bit_vector_size", ["bit_vector header"]),
                                        ("string_length", ["string length"])])
def test_inspect_names_the_first_inconsistent_field_of_a_damaged_file(binary, built, tmp_path, what, names):
    good = str(tmp_path / "good.ibf")
    gf.write_ibf(good, built)
    data = _damage(bytearray(open(good, "rb").read()), what, built)
    bad = str(tmp_path / f"{what}.ibf")
    open(bad, "wb").write(bytes(data))
    rc, out, err = _inspect(binary, bad)
    assert rc == 1, out
    last = out.splitlines()[-1]
    assert last.startswith("result      INCONSISTENT -- first inconsistency:"), out
    assert all(n in last for n in names), last
    # and the loader proper refuses the same file the same way (nothing is classified against a misread filter)
    fq = str(tmp_path / "r.fq")
    gf.write_fastq(fq, [("r1", "ACGT" * 20)])
    p = subprocess.run([binary, "-i", bad, "-r", fq, "-o", str(tmp_path / "o")], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and any(n in p.stderr for n in names), p.stderr


def test_inspect_hibf_file(binary, tmp_path):
    h = gf.random_hibf(n_user_bins=40, tmax=8, max_depth=3, seed=4)
    path = str(tmp_path / "f.hibf")
    names = [[f"/x/ub{u}|||1.minimiser"] for u in range(h.n_user_bins)]
    gf.write_hibf(path, h, names, k=19, w=31, fpr=0.001)
    rc, out, err = _inspect(binary, path, "--hibf")
    assert rc == 0, out + err
    assert int(_field(out, "window_size")) == 31 and int(_field(out, "ibf_vector.size")) == len(h.ibfs)
    assert int(_field(out, "user_bin_filenames.size")) == h.n_user_bins and float(_field(out, "fpr")) == 0.001
    assert "--- ibf_vector[0]" in out and f"--- {len(h.ibfs)} IBFs parsed" in out
    assert out.splitlines()[-1].startswith(f"result      CONSISTENT: HIBF, k=19 w=31, {len(h.ibfs)} IBF(s), {h.n_user_bins} user bins")
    # read as a flat .ibf the same file is refused, with the field that gave it away
    rc, out, _ = _inspect(binary, path)
    assert rc == 1 and "first inconsistency" in out
    # cut in the middle of the tables that follow the matrices
    data = open(path, "rb").read()
    open(path, "wb").write(data[:-9])
    rc, out, _ = _inspect(binary, path, "--hibf")
    assert rc == 1 and ("truncated" in out.splitlines()[-1] or "implausible container size" in out.splitlines()[-1])
