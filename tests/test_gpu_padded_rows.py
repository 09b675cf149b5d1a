"""The IBFs of an HIBF keep their rows on the device at a stride padded to whole 128-byte lines (gn_pad_row_words, gn_internal.h); the C
ABI still takes and returns rows of ceil(bins / 64) words.  Checks: what goes in comes out (upload, write_rows with a wider source,
download_rows, download_row_list), the device footprint is the padded one, emplace lands in the right words, and a filter created with
the switch `hibf_dense_rows` (the old layout) classifies the same reads to the same records."""
import numpy as np
import pytest

import gpu_util as gu
import ganon_fixtures as gf
import oracle

pytestmark = pytest.mark.gpu


def _pad(w):
    if w >= 16:
        return (w + 15) & ~15
    p = 1
    while p < w:
        p <<= 1
    return p


def _rows(rng, S, bins):
    W = (bins + 63) >> 6
    r = rng.integers(0, 1 << 63, size=(S, W), dtype=np.uint64) & rng.integers(0, 1 << 63, size=(S, W), dtype=np.uint64)
    if bins & 63:
        r[:, -1] &= np.uint64((1 << (bins & 63)) - 1)
    return r


@pytest.mark.parametrize("bins", [130, 700, 1100, 64, 1024])
def test_rows_round_trip(bins):
    import ganon_amd as hip
    rng = np.random.default_rng(bins)
    S, W = 4096, (bins + 63) >> 6
    top = _rows(rng, S, 64)
    child = _rows(rng, S, bins)
    # a two-level tree: one merged bin at the top leads to the child whose bins are the user bins
    nxt = [np.full(64, -1, dtype=np.int64), np.full(bins, -1, dtype=np.int64)]
    b2u = [np.full(64, -1, dtype=np.int64), np.arange(bins, dtype=np.int64)]
    nxt[0][0] = 1
    b2u[0][1:] = np.arange(bins, bins + 63)
    flt = hip.HipFilter.hibf([(top.reshape(-1), 64, S, 2), (child.reshape(-1), bins, S, 2)], nxt, b2u, bins + 63)
    assert flt.info()["device_bytes"] == S * 8 * (1 + _pad(W))
    assert np.array_equal(flt.download_rows(0, S, W, ibf_idx=1), child)
    assert np.array_equal(flt.download_rows(1000, 77, W, ibf_idx=1), child[1000:1077])
    idx = rng.integers(0, S, size=300).astype(np.uint64)
    assert np.array_equal(flt.download_row_list(idx, W, ibf_idx=1), child[idx.astype(np.int64)])
    # write_rows from a wider source (a column slice of it), in two pieces
    wide = rng.integers(0, 1 << 63, size=(S, W + 5), dtype=np.uint64)
    flt.write_rows(0, wide[:1500], word_lo=3, ibf_idx=1)
    flt.write_rows(1500, wide[1500:], word_lo=3, ibf_idx=1)
    flt.finalize()
    exp = wide[:, 3:3 + W].copy()
    if bins & 63:
        exp[:, -1] &= np.uint64((1 << (bins & 63)) - 1)
    assert np.array_equal(flt.download_rows(0, S, W, ibf_idx=1), exp)
    # emplace: the bit of (hash, bin) lands in word bin >> 6 of the hash's rows
    flt.write_rows(0, np.zeros((S, W), dtype=np.uint64), ibf_idx=1)
    hs = rng.integers(0, 1 << 63, size=50, dtype=np.uint64)
    bb = rng.integers(0, bins, size=50).astype(np.uint32)
    flt.emplace(hs, bb, ibf_idx=1)
    got = flt.download_rows(0, S, W, ibf_idx=1)
    ref = oracle.Ibf(bins, S, 2)
    for h, b in zip(hs.tolist(), bb.tolist()):
        ref.emplace(int(h), int(b))
    assert np.array_equal(got, ref.data.reshape(S, W))
    flt.free()


@pytest.mark.parametrize("seed,tmax", [(1, 192), (2, 640), (3, 320)])
def test_dense_and_padded_layout_same_records(seed, tmax):
    import ganon_amd as hip
    rng = np.random.default_rng(seed)
    k, w, n_ub = 19, 31, 900
    genomes = {ub: gu.random_seq(rng, 2600) for ub in range(0, n_ub, 15)}
    uh = {ub: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in genomes.items()}
    hb = gf.random_hibf(n_ub, tmax, 2, seed=seed, density=0.3, hash_funs=3, user_hashes=uh)
    keys = sorted(genomes)
    reads = []
    for i in range(400):
        g = genomes[keys[i % len(keys)]]
        p = int(rng.integers(0, len(g) - 150))
        reads.append(g[p:p + 150] if i % 4 else gu.random_seq(rng, 150))
    bases, off1, _ = gu.pack_reads(reads)
    outs, sizes = [], []
    for dense in (False, True):
        if dense:
            gu.SW.on("hibf_dense_rows")
        flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
        if dense:
            gu.SW.off("hibf_dense_rows")
        sizes.append(flt.info()["device_bytes"])
        st = hip.HipStream(flt, len(reads), bases.size)
        st.submit(bases, off1, None, k, w, 0.5)
        nh, status, mo, m = st.fetch()
        outs.append((mo.copy(), m.copy(), st.timings()["algo_bytes"]))
        st.destroy()
        flt.free()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]          # algorithmic bytes count the words that hold bins, in either layout
    assert sizes[0] >= sizes[1]              # ... and the padded tree is the larger one (equal when every width is a power of two)
    assert len(outs[0][1]) > 100
