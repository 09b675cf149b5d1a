"""Worker of the world-size-2 gloo tests (CPU).  The local hot path is the oracle here (test infrastructure); what is
under test is the distributed logic of ganon_amd.partition / ganon_amd.dist."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import ganon_fixtures as gf  # noqa: E402
import gpu_util as gu  # noqa: E402
import oracle  # noqa: E402
from ganon_amd import dist as gdist  # noqa: E402
from ganon_amd import partition as gp  # noqa: E402

K, W = 19, 31


def make_case(seed=5):
    rng = np.random.default_rng(seed)
    bins, rows, h = 700, 1500, 3
    ibf = gf.random_ibf(bins, rows, h, 0.35, seed=seed)
    # contiguous split bins: target t owns a run of 1..4 bins, a few bins unassigned
    b2t = np.full(bins, gp.NO_TARGET, dtype=np.uint32)
    b, t = 0, 0
    while b < bins:
        run = int(rng.integers(1, 5))
        if rng.random() < 0.05:
            b += 1
            continue
        b2t[b:b + run] = t
        b += run
        t += 1
    genomes = [gu.random_seq(rng, 1500) for _ in range(30)]
    for gi, g in enumerate(genomes):
        hv = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), K, W))
        tb = np.nonzero(b2t == (gi * 5) % t)[0]
        ibf.emplace_many(hv, int(tb[0]))
    seqs = []
    for i in range(120):
        if i % 2:
            g = genomes[i % 30]
            p = int(rng.integers(0, 1300))
            seqs.append(g[p:p + 150])
        else:
            seqs.append(gu.random_seq(rng, int(rng.integers(20, 200))))
    return ibf, b2t, t, seqs


def oracle_local_classify(rows, bins, bin_size, hash_funs, bin2target, n_targets, bases, off1, off2, k, w, rel_cutoff):
    ibf = oracle.Ibf(bins, bin_size, hash_funs, rows)
    n = len(off1) - 1
    nh = np.zeros(n, np.uint32)
    st = np.zeros(n, np.uint8)
    mo = np.zeros(n + 1, np.uint64)
    recs = []
    for r in range(n):
        s = bases[int(off1[r]):int(off1[r + 1])]
        if len(s) < w:
            st[r] = 1
        else:
            hh = oracle.minimiser_hash(oracle.to_ranks(s), k, w)
            nh[r] = len(hh)
            m, _ = gu.oracle_matches(ibf, bin2target, n_targets, hh, rel_cutoff)
            recs += [(r, t, c) for t, c in m]
        mo[r + 1] = len(recs)
    return nh, st, mo, np.array(recs, dtype=gp.MATCH_DTYPE) if recs else np.zeros(0, gp.MATCH_DTYPE)


class OracleLocal(gp.LocalFilter):
    """a rank's column slice with the oracle as the local hot path (CPU tests of the distributed logic)"""

    def __init__(self, rows, bins, bin_size, hash_funs, bin2target, n_targets):
        self.args = (rows, bins, bin_size, hash_funs, bin2target, n_targets)

    def classify(self, bases, off1, off2, k, w, rel_cutoff):
        return oracle_local_classify(*self.args, bases, off1, off2, k, w, rel_cutoff)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, _, world = gdist.env_rank_world()
    gdist.init("gloo")
    ibf, b2t, n_targets, seqs = make_case()
    bases, off1, _ = gu.pack_reads(seqs, None)
    if mode == "partition":
        part = gp.PartitionedIbf.from_host_rows(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, rank, world, OracleLocal)
        lo, hi, nh, st, mine = part.classify(bases, off1, None, K, W, 0.25)
        np.save(f"{out}.{rank}.npy", mine)
        np.save(f"{out}.{rank}.range.npy", np.array([lo, hi, part.slice.word_lo, part.slice.word_hi]))
        # the exchange step is timed per classify() (what bench.py prints as slice1t_exchange_ms)
        assert len(part.exchange_ms) == 1 and part.exchange_ms[0] > 0 and len(part.merge_ms) == 1
    elif mode == "shard":
        lo, hi = gdist.shard_range(len(seqs), rank, world)
        sb, so, _ = gu.pack_reads(seqs[lo:hi], None)
        nh, st, mo, m = oracle_local_classify(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets, sb, so, None, K, W, 0.25)
        m = m.copy()
        m["read"] += lo  # back to global read indices
        np.save(f"{out}.{rank}.npy", m)
        total = gdist.sum_over_ranks(hi - lo)
        slowest = gdist.max_over_ranks(float(rank + 1))
        np.save(f"{out}.{rank}.range.npy", np.array([lo, hi, total, int(slowest)]))
    elif mode == "proof":
        # what bench.py's record proves its ranks with (rank_proof): ranks that answered, every rank's short text, every rank's own clock
        import json
        texts = gdist.all_gather_text(f"0000:{rank:02x}:00/rank{rank}")
        times = gdist.all_gather_float(40.0 + rank)
        json.dump({"ranks_seen": gdist.sum_over_ranks(1), "group_size": gdist.group_size(), "texts": texts, "times": times},
                  open(f"{out}.{rank}.json", "w"))
    gdist.barrier()


if __name__ == "__main__":
    main()
