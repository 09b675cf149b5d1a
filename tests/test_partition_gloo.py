"""world_size-2 gloo tests (CPU) of the multi-GPU logic: bin-range partitioned filter with the sparse-match
all-to-all (BASELINE.json configs[4]) and read sharding with replicated filter (configs[1..3])."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import dist_worker as dw
import gpu_util as gu
import oracle
from ganon_amd import partition as gp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(mode, out, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), mode, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()


def _expected():
    ibf, b2t, n_targets, seqs = dw.make_case()
    recs = []
    for r, s in enumerate(seqs):
        if len(s) < dw.W:
            continue
        hh = oracle.minimiser_hash(oracle.to_ranks(s), dw.K, dw.W)
        m, _ = gu.oracle_matches(ibf, b2t, n_targets, hh, 0.25)
        recs += [(r, t, c) for t, c in m]
    return np.array(recs, dtype=gp.MATCH_DTYPE), len(seqs)


def test_plan_partition_properties():
    ibf, b2t, n_targets, _ = dw.make_case()
    for world in (2, 3, 8):
        plan = gp.plan_partition(b2t, ibf.bins, world)
        seen = np.zeros(n_targets, dtype=int)
        for sl in plan:
            seen[sl.targets_global] += 1
            # every bin of an owned target is inside the slice and mapped back to it
            for lt, g in enumerate(sl.targets_global):
                gb = np.nonzero(b2t == g)[0]
                assert gb.min() >= sl.word_lo * 64 and gb.max() < sl.word_hi * 64
                assert np.array_equal(np.nonzero(sl.bin2target_local == lt)[0] + sl.word_lo * 64, gb)
        assert (seen == 1).all()  # each target owned by exactly one rank


def test_partitioned_filter_gloo(tmp_path):
    out = str(tmp_path / "part")
    _run_world("partition", out)
    exp, n_reads = _expected()
    got = []
    for rank in range(2):
        m = np.load(f"{out}.{rank}.npy")
        lo, hi, wlo, whi = np.load(f"{out}.{rank}.range.npy")
        assert ((m["read"] >= lo) & (m["read"] < hi)).all()  # only reads this rank owns
        got.append(m)
    got = np.concatenate(got)
    assert len(exp) > 50
    assert np.array_equal(got, exp)  # owners hold ascending reads; concatenation == single-filter result


@pytest.mark.parametrize("world", [2, 3])
def test_rank_proof_helpers_gloo(tmp_path, world):
    """the collectives behind bench.py's `ranks` object (ranks_seen, devices, per-rank step times): every rank sees every rank's entry,
    in rank order"""
    import json
    out = str(tmp_path / "proof")
    _run_world("proof", out, world=world)
    for rank in range(world):
        r = json.load(open(f"{out}.{rank}.json"))
        assert r["ranks_seen"] == world and r["group_size"] == world
        assert r["texts"] == [f"0000:{i:02x}:00/rank{i}" for i in range(world)]
        assert r["times"] == [40.0 + i for i in range(world)]


def test_read_sharding_gloo(tmp_path):
    out = str(tmp_path / "shard")
    _run_world("shard", out)
    exp, n_reads = _expected()
    got = np.concatenate([np.load(f"{out}.{r}.npy") for r in range(2)])
    assert np.array_equal(got, exp)
    for r in range(2):
        lo, hi, total, slowest = np.load(f"{out}.{r}.range.npy")
        assert total == n_reads and slowest == 2  # sum / max over ranks
