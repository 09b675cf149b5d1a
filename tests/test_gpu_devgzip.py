"""`reads.fq.gz` inflated on the device and classified from there (host/devgzip.cpp + csrc/gn_inflate.hip): every output file must
be what the host inflater's run writes, byte for byte -- and what the oracle backend writes (same host code, CPU hot path, zlib).

Reference behaviour: parse_reads reads gzip input through one zlib stream (GanonClassify.cpp:1220-1287); a parse error reports,
keeps what was read and goes on with the next file (:1278-1283)."""
import gzip
import os
import zlib

import numpy as np
import pytest

import cli_util as cu
from test_cli_kat import _sim_reads, oracle_bin, sim_db  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

EXTS = (".all", ".one", ".unc", ".rep", ".sta")


def _records(n, seed=7, line_end="\n"):
    """n single-end records: the fixture reads (true matches in sim_db) under fresh ids, mixed with random reads"""
    r1, r2 = _sim_reads()
    pool = [s for _, s in r1] + [s for _, s in r2]
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        if rng.random() < 0.6:
            s = pool[int(rng.integers(0, len(pool)))]
        else:
            s = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(60, 151))))
        q = "".join(chr(33 + int(x)) for x in rng.choice([2, 11, 25, 37], size=len(s), p=[0.05, 0.1, 0.25, 0.6]))
        out.append(f"@read{i}:{int(rng.integers(1, 99999))} some description {i % 7}{line_end}{s}{line_end}+{line_end}{q}{line_end}")
    return out


def _run(binary, sim_db, reads, out_prefix, env=None, extra=()):
    args = ["--ibf", sim_db["ibf"], "--tax", sim_db["tax"], "--single-reads", reads, "-o", out_prefix, "--output-all", "--output-lca",
            "--output-unclassified", "--output-stats", "--rel-cutoff", "0.25", "--rel-filter", "0.1"] + (list(extra) if "--verbose" in extra else ["--quiet"] + list(extra))
    e = dict(os.environ)
    e.update(env or {})
    import subprocess
    p = subprocess.run([binary] + args, capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr
    return p


DEV = {"GANON_HOST_DEVICE_INFLATE_MIN": "0", "GANON_HOST_TIMING": "1"}
HOST = {"GANON_HOST_DEVICE_INFLATE": "0"}


def _same_files(a, b, exts=EXTS):
    for ext in exts:
        x, y = open(a + ext, "rb").read(), open(b + ext, "rb").read()
        if ext == ".sta":  # (the row holds run times)
            x, y = x.split(b"\n")[0], y.split(b"\n")[0]
        assert x == y, ext


def _device_path_taken(p):
    return "device inflate:" in p.stderr and "given up" not in p.stderr


@pytest.mark.parametrize("level,env_extra", [(6, {}), (1, {}), (9, {"GANON_HOST_DEVICE_INFLATE_CHUNK": "4096", "GANON_HOST_DEVICE_INFLATE_STEP": "262144"}),
                                             (6, {"GANON_HOST_SLAB_BYTES": "300000"})])
def test_device_inflate_outputs_equal_host_inflate(sim_db, oracle_bin, tmp_path, level, env_extra):
    text = "".join(_records(30000, seed=level)).encode()
    fq = str(tmp_path / "reads.fq.gz")
    with open(fq, "wb") as f:
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        f.write(co.compress(text) + co.flush())
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, dict(DEV, **env_extra))
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    assert _device_path_taken(pa), pa.stderr
    _same_files(a, b)
    res = cu.Res(a)
    assert res.total_classified > 1000 and res.total_classified + res.total_unclassified == 30000
    if level == 6 and not env_extra:
        c = str(tmp_path / "ora")
        _run(oracle_bin, sim_db, fq, c)
        _same_files(a, c, (".all", ".one", ".unc", ".rep"))


def test_members_and_fasta(sim_db, tmp_path):
    recs = _records(9000, seed=11)
    fq = str(tmp_path / "reads.fastq.gz")
    with open(fq, "wb") as f:  # three members, the cut inside a record
        t = "".join(recs).encode()
        for part in (t[:400_001], t[400_001:900_000], t[900_000:]):
            f.write(gzip.compress(part, 6))
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, DEV)
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    assert _device_path_taken(pa) and "3 members" in pa.stderr, pa.stderr
    _same_files(a, b)
    fa = str(tmp_path / "reads.fa.gz")
    with open(fa, "wb") as f:
        f.write(gzip.compress("".join(">" + r.split("\n")[0][1:] + "\n" + r.split("\n")[1] + "\n" for r in recs).encode(), 6))
    a, b = str(tmp_path / "deva"), str(tmp_path / "hosta")
    pa = _run(cu.BIN_HIP, sim_db, fa, a, DEV)
    _run(cu.BIN_HIP, sim_db, fa, b, HOST)
    assert _device_path_taken(pa), pa.stderr
    _same_files(a, b)


@pytest.mark.parametrize("case", ["no_final_newline", "wrapped_record", "crlf", "bad_letter", "truncated_gz"])
def test_irregular_input_ends_like_the_host_path(sim_db, tmp_path, case):
    recs = _records(6000, seed=23, line_end="\r\n" if case == "crlf" else "\n")
    if case == "wrapped_record":
        h, s, p, q = recs[3000].split("\n")[:4]
        recs[3000] = f"{h}\n{s[:40]}\n{s[40:]}\n{p}\n{q[:40]}\n{q[40:]}\n"
    if case == "bad_letter":
        h, s, p, q = recs[4000].split("\n")[:4]
        recs[4000] = f"{h}\n{s[:10]}!{s[11:]}\n{p}\n{q}\n"
    text = "".join(recs).encode()
    if case == "no_final_newline":
        text = text[:-1]
    gz = gzip.compress(text, 6)
    if case == "truncated_gz":
        gz = gz[:len(gz) * 2 // 3]
    fq = str(tmp_path / "reads.fq.gz")
    open(fq, "wb").write(gz)
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, DEV)
    pb = _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    _same_files(a, b)
    err = lambda p: [l for l in p.stderr.split("\n") if l.startswith("Error parsing")]  # noqa: E731
    assert err(pa) == err(pb)
    if case in ("bad_letter", "truncated_gz"):
        assert err(pa)


def _pair_records(n, seed=5):
    r1, r2 = _sim_reads()
    rng = np.random.default_rng(seed)
    a, b = [], []
    for i in range(n):
        if rng.random() < 0.6:
            j = int(rng.integers(0, len(r1)))
            s1, s2 = r1[j][1], r2[j][1]
        else:
            s1 = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(60, 151))))
            s2 = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(40, 151))))
        q1 = "".join(chr(33 + int(x)) for x in rng.choice([2, 11, 25, 37], size=len(s1)))
        q2 = "".join(chr(33 + int(x)) for x in rng.choice([2, 11, 25, 37], size=len(s2)))
        a.append(f"@pair{i}:{int(rng.integers(1, 99999))}/1\n{s1}\n+\n{q1}\n")
        b.append(f"@pair{i}:{int(rng.integers(1, 99999))}/2 mate\n{s2}\n+\n{q2}\n")
    return a, b


def _run_pair(binary, sim_db, f1, f2, out_prefix, env=None):
    import subprocess
    args = ["--ibf", sim_db["ibf"], "--tax", sim_db["tax"], "--paired-reads", f1 + "," + f2, "-o", out_prefix, "--output-all", "--output-lca",
            "--output-unclassified", "--output-stats", "--quiet", "--rel-cutoff", "0.25", "--rel-filter", "0.1"]
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary] + args, capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr
    return p


@pytest.mark.parametrize("case", ["equal", "small_steps", "mates_short", "mates_long", "mate_wrapped", "first_truncated_gz"])
def test_paired_gzip_files_from_the_device(sim_db, oracle_bin, tmp_path, case):
    a, b = _pair_records(24000, seed=len(case))
    if case == "mates_short":
        b = b[:15000]
    if case == "mates_long":
        b = b + _pair_records(500, seed=99)[1]
    if case == "mate_wrapped":
        h, s, p, q = b[9000].split("\n")[:4]
        b[9000] = f"{h}\n{s[:20]}\n{s[20:]}\n{p}\n{q[:20]}\n{q[20:]}\n"
    f1, f2 = str(tmp_path / "r.1.fq.gz"), str(tmp_path / "r.2.fq.gz")
    g1, g2 = gzip.compress("".join(a).encode(), 6), gzip.compress("".join(b).encode(), 3)
    if case == "first_truncated_gz":
        g1 = g1[:len(g1) * 3 // 5]
    open(f1, "wb").write(g1)
    open(f2, "wb").write(g2)
    env = dict(DEV)
    if case == "small_steps":
        env.update({"GANON_HOST_DEVICE_INFLATE_CHUNK": "4096", "GANON_HOST_DEVICE_INFLATE_STEP": "262144", "GANON_HOST_SLAB_BYTES": "400000"})
    x, y = str(tmp_path / "dev"), str(tmp_path / "host")
    px = _run_pair(cu.BIN_HIP, sim_db, f1, f2, x, env)
    py = _run_pair(cu.BIN_HIP, sim_db, f1, f2, y, HOST)
    assert "device inflate:" in px.stderr, px.stderr
    _same_files(x, y)
    err = lambda p: [l for l in p.stderr.split("\n") if l.startswith("Error parsing")]  # noqa: E731
    assert err(px) == err(py)
    if case == "equal":
        z = str(tmp_path / "ora")
        _run_pair(oracle_bin, sim_db, f1, f2, z)
        _same_files(x, z, (".all", ".one", ".unc", ".rep"))
        res = cu.Res(x)
        assert res.total_classified > 1000 and res.total_classified + res.total_unclassified == 24000


def test_no_room_on_the_device_means_the_host_inflater(sim_db, tmp_path):
    # the filters are loaded after the reader starts: an inflater must not take what they need ($GANON_HOST_DEVICE_INFLATE_ROOM = what to keep free for it)
    fq = str(tmp_path / "reads.fq.gz")
    open(fq, "wb").write(gzip.compress("".join(_records(8000, seed=3)).encode(), 6))
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, dict(DEV, GANON_HOST_DEVICE_INFLATE_ROOM=str(1 << 50)), extra=["--verbose"])
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    assert "device inflate:" not in pa.stderr and "inflated by the host" in pa.stderr
    _same_files(a, b)


def test_two_workers_on_one_device_take_the_pieces(sim_db, tmp_path):
    # --device 0,0: two device entries (two workers, each with streams of its own) share the inflater's pieces
    fq = str(tmp_path / "reads.fq.gz")
    open(fq, "wb").write(gzip.compress("".join(_records(20000, seed=13)).encode(), 6))
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, dict(DEV, GANON_HOST_SLAB_BYTES="500000"), extra=["--device", "0,0"])
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    assert _device_path_taken(pa), pa.stderr
    _same_files(a, b)


@pytest.mark.parametrize("paired", [False, True])
def test_two_hierarchy_levels_get_their_letters_from_the_device(sim_db, tmp_path, paired):
    # reads the first level leaves unclassified go on to the second with their letters (GanonClassify.cpp:811-820): for a text the host
    # never held they come back with the results (gn_stream_fetch_letters).  Level 1 = the HIBF (strict cutoff), level 2 = the flat filter.
    import subprocess
    if paired:
        a, b = _pair_records(12000, seed=21)
        f1, f2 = str(tmp_path / "r.1.fq.gz"), str(tmp_path / "r.2.fq.gz")
        open(f1, "wb").write(gzip.compress("".join(a).encode(), 6))
        open(f2, "wb").write(gzip.compress("".join(b).encode(), 6))
        reads = ["--paired-reads", f1 + "," + f2]
    else:
        fq = str(tmp_path / "reads.fq.gz")
        open(fq, "wb").write(gzip.compress("".join(_records(15000, seed=17)).encode(), 6))
        reads = ["--single-reads", fq]

    def run(prefix, env):
        args = ["--ibf", sim_db["ibf"] + "," + sim_db["ibf"], "--tax", sim_db["tax"] + "," + sim_db["tax"], "--hierarchy-labels", "first,second",
                "--rel-cutoff", "0.9,0.25", "--rel-filter", "0.1,0.1", "-o", prefix, "--output-all", "--output-lca", "--output-unclassified", "--quiet"] + reads
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([cu.BIN_HIP] + args, capture_output=True, text=True, timeout=600, env=e)
        assert p.returncode == 0, p.stderr
        return p

    x, y = str(tmp_path / "dev"), str(tmp_path / "host")
    px = run(x, DEV)
    run(y, HOST)
    assert "device inflate:" in px.stderr and "given up" not in px.stderr, px.stderr
    names = sorted(f for f in os.listdir(tmp_path) if f.startswith("dev."))
    assert any(".second." in f or "second" in f for f in names) or len(names) >= 3, names
    for f in names:
        assert open(os.path.join(tmp_path, f), "rb").read() == open(os.path.join(tmp_path, "host" + f[3:]), "rb").read(), f
    rep = open(x + ".rep").read()
    assert "second" in rep  # the second level classified reads: it got their letters


def test_several_files_in_one_run(sim_db, tmp_path):
    # gzip, plain, gzip again (three members), a file too small for the device path: every file's source is opened and ended in turn
    import subprocess
    recs = _records(26000, seed=29)
    files = []
    for i, (a, b, how) in enumerate([(0, 9000, "gz"), (9000, 14000, "plain"), (14000, 25990, "gz3"), (25990, 26000, "gz")]):
        t = "".join(recs[a:b]).encode()
        f = str(tmp_path / (f"part{i}.fq" + ("" if how == "plain" else ".gz")))
        if how == "plain":
            open(f, "wb").write(t)
        elif how == "gz3":
            open(f, "wb").write(gzip.compress(t[:len(t) // 3], 6) + gzip.compress(t[len(t) // 3:len(t) // 2], 1) + gzip.compress(t[len(t) // 2:], 9))
        else:
            open(f, "wb").write(gzip.compress(t, 6))
        files.append(f)
    x, y = str(tmp_path / "dev"), str(tmp_path / "host")
    px = _run(cu.BIN_HIP, sim_db, ",".join(files), x, dict(DEV, GANON_HOST_DEVICE_INFLATE_MIN="4096"))
    _run(cu.BIN_HIP, sim_db, ",".join(files), y, HOST)
    assert px.stderr.count("device inflate:") == 2, px.stderr
    _same_files(x, y)
    res = cu.Res(x)
    assert res.total_classified + res.total_unclassified == 26000


def test_blocked_gzip_input(sim_db, tmp_path):
    # bgzip's output (members of 64 KiB, BC field): on the device the members are the chunk starts; the host path has its own BGZF reader
    from test_gpu_inflate import _bgzf
    fq = str(tmp_path / "reads.fq.gz")
    open(fq, "wb").write(_bgzf("".join(_records(25000, seed=31)).encode()))
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, DEV)
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    assert _device_path_taken(pa) and "0 fix-ups" in pa.stderr, pa.stderr
    _same_files(a, b)


TURNS = {"GANON_HOST_DEVICE_INFLATE_CHUNK": "4096", "GANON_HOST_DEVICE_INFLATE_STEP": "131072", "GANON_HOST_SLAB_BYTES": "200000"}


@pytest.mark.parametrize("turns,devices", [("2", "0"), ("3", "0,0,0"), ("2", "0,0")])
def test_steps_of_a_file_inflated_in_turns_by_several_inflaters(sim_db, oracle_bin, tmp_path, turns, devices):
    """Every distinct device of a run gets an inflater of a .gz file and they take its steps in turn (gn_inflate_set_turns /
    gn_inflate_handoff; DESIGN 7: one inflating device capped a .gz run at 100-117 Mreads/s for any number of GPUs).  One GPU here:
    $GANON_HOST_DEVICE_INFLATE_TURNS puts several inflaters on it -- the hand-over of position, window, CRC and carried record is the
    same peer-copy code.  Outputs: the host inflater's, byte for byte, and the oracle backend's."""
    text = "".join(_records(40000, seed=41)).encode()
    fq = str(tmp_path / "reads.fq.gz")
    with open(fq, "wb") as f:   # two members, the cut inside a record: a member ends inside a step, its CRC is handed over
        f.write(gzip.compress(text[:3_000_001], 6) + gzip.compress(text[3_000_001:], 6))
    a, b, c = str(tmp_path / "turns"), str(tmp_path / "host"), str(tmp_path / "ora")
    pa = _run(cu.BIN_HIP, sim_db, fq, a, dict(DEV, GANON_HOST_DEVICE_INFLATE_TURNS=turns, **TURNS), extra=("--device", devices))
    _run(cu.BIN_HIP, sim_db, fq, b, HOST)
    _run(oracle_bin, sim_db, fq, c)
    assert _device_path_taken(pa) and f"{turns} inflaters taking turns" in pa.stderr and "2 members" in pa.stderr, pa.stderr
    _same_files(a, b)
    _same_files(a, c, (".all", ".one", ".unc", ".rep"))
    assert cu.Res(a).total_classified > 1000


def test_turns_with_irregular_input_and_damage_end_like_the_host_path(sim_db, tmp_path):
    recs = _records(20000, seed=29)
    h, s, p, q = recs[15000].split("\n")[:4]
    recs[15000] = f"{h}\n{s[:40]}\n{s[40:]}\n{p}\n{q[:40]}\n{q[40:]}\n"        # a wrapped record late in the file: the sequential reader takes over there
    gz = gzip.compress("".join(recs).encode(), 6)
    for tag, data in (("wrapped", gz), ("truncated", gz[:len(gz) * 3 // 5])):
        fq = str(tmp_path / f"{tag}.fq.gz")
        open(fq, "wb").write(data)
        a, b = str(tmp_path / f"{tag}_dev"), str(tmp_path / f"{tag}_host")
        pa = _run(cu.BIN_HIP, sim_db, fq, a, dict(DEV, GANON_HOST_DEVICE_INFLATE_TURNS="3", **TURNS))
        pb = _run(cu.BIN_HIP, sim_db, fq, b, HOST)
        _same_files(a, b)
        err = lambda p: [l for l in p.stderr.split("\n") if l.startswith("Error parsing")]  # noqa: E731
        assert err(pa) == err(pb)


@pytest.mark.parametrize("case,turns", [("equal", "2"), ("equal", "3"), ("mates_short", "2"), ("mate_wrapped", "3"), ("first_truncated_gz", "2")])
def test_paired_gzip_files_inflated_in_turns(sim_db, oracle_bin, tmp_path, case, turns):
    """Both files of a pair with their steps taken in turn by several inflaters (file 2 is cut where file 1's records end by number,
    GanonClassify.cpp:1240-1252): a pair's two pieces may lie on two devices (gn_stream_upload_text_pair_devices) -- on this box
    the inflaters share the one GPU.  Outputs and messages: the host inflater's run, byte for byte."""
    a, b = _pair_records(30000, seed=7 + len(case))
    if case == "mates_short":
        b = b[:19000]
    if case == "mate_wrapped":
        h, s, p, q = b[21000].split("\n")[:4]
        b[21000] = f"{h}\n{s[:20]}\n{s[20:]}\n{p}\n{q[:20]}\n{q[20:]}\n"
    f1, f2 = str(tmp_path / "r.1.fq.gz"), str(tmp_path / "r.2.fq.gz")
    g1, g2 = gzip.compress("".join(a).encode(), 6), gzip.compress("".join(b).encode(), 3)
    if case == "first_truncated_gz":
        g1 = g1[:len(g1) * 3 // 5]
    open(f1, "wb").write(g1)
    open(f2, "wb").write(g2)
    env = dict(DEV, GANON_HOST_DEVICE_INFLATE_TURNS=turns, GANON_HOST_DEVICE_INFLATE_CHUNK="4096", GANON_HOST_DEVICE_INFLATE_STEP="131072",
               GANON_HOST_SLAB_BYTES="300000")
    x, y = str(tmp_path / "dev"), str(tmp_path / "host")
    px = _run_pair(cu.BIN_HIP, sim_db, f1, f2, x, env)
    py = _run_pair(cu.BIN_HIP, sim_db, f1, f2, y, HOST)
    assert px.stderr.count(f"{turns} inflaters taking turns") == 2, px.stderr    # both files
    _same_files(x, y)
    err = lambda p: [l for l in p.stderr.split("\n") if l.startswith("Error parsing")]  # noqa: E731
    assert err(px) == err(py)
    if case == "equal":
        z = str(tmp_path / "ora")
        _run_pair(oracle_bin, sim_db, f1, f2, z)
        _same_files(x, z, (".all", ".one", ".unc", ".rep"))
        assert cu.Res(x).total_classified > 1000
