/*
 * ganon_oracle.h -- CPU restatement of ganon's read-classification hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ganon_amd/ (the product) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported CPU baseline.
 *
 * The arithmetic of the path lives in SeqAn3 3.3.0 (seqan3::views::minimiser_hash,
 * seqan3::interleaved_bloom_filter, counting_agent) and raptor 3.0.1 (HIBF), which are
 * un-vendored submodules of the reference (/root/reference/.gitmodules:1-15,
 * CMakeLists.txt:119-120) and absent from this image, so the reference cannot be compiled here.
 * This file restates their published algorithms and anchors parity on the
 * reference's own call sites and known-answer tests:
 *   - tests/ganon-classify/GanonClassify.test.cpp (all per-read per-target counts; ported as
 *     data in tests/golden/kat_classify.json) -- every one is reproduced (tests/test_oracle_kat.py)
 *   - tests/utils/LCA.test.cpp (tests/golden/lca_*.tax)
 * Parity status: PINNED by the reference's KATs for minimiser semantics, seed, thresholds,
 * filters; the IBF hash constants are pinned only weakly (tiny KAT filters would show false
 * positives under a wrong hash); the .ibf/.hibf byte layout is unpinned in-tree.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef GANON_ORACLE_H
#define GANON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a-1 input semantics: char -> dna4 rank (src/utils/include/utils/dna4_traits.hpp:15-18,
 * SeqAn3 dna4 char_to_rank_table).  Returns 0..3; *legal=0 when the char is
 * not in the dna15 legal alphabet (the reference raises parse_error, GanonClassify.cpp:1278). */
uint8_t gno_char_to_rank(unsigned char c, int* legal);

/* ---- a-2 minimiser hash (call sites GanonClassify.cpp:647-650,693,698; seed
 * src/utils/include/utils/adjust_seed.hpp:33-37) */
uint64_t gno_adjust_seed(uint32_t k);
/* ranks[0..L) in 0..3.  Writes up to cap hashes, returns the number that WOULD be emitted. */
size_t gno_minimiser_hash(const uint8_t* ranks, size_t L, uint32_t k, uint32_t w, uint64_t* out, size_t cap);

/* ---- a-3 threshold (GanonClassify.cpp:492-495,720-724) */
uint64_t gno_threshold_rel(uint64_t n_hashes, double p);    /* ceil(n*p) in double */
uint64_t gno_threshold_cutoff(uint64_t n_hashes, double p); /* max(1, ceil(n*p)) */

/* ---- a-4/a-5 interleaved Bloom filter (SeqAn3 3.3.0 interleaved_bloom_filter) */
typedef struct
{
    uint64_t* data;       /* S * W little-endian u64 words, word(r, b) = data[r*W + b] */
    uint64_t  bins;       /* user-visible bin count B */
    uint64_t  bin_size;   /* rows S */
    uint64_t  bin_words;  /* W = ceil(B/64) */
    uint64_t  hash_shift; /* countl_zero(S) */
    uint32_t  hash_funs;  /* h <= 5 */
} gno_ibf;

extern const uint64_t GNO_IBF_SEEDS[5];
uint64_t gno_ibf_hash_shift(uint64_t bin_size);
/* row index (0..S) of hash function i for value v */
uint64_t gno_ibf_row(const gno_ibf* f, uint64_t v, uint32_t i);
void     gno_ibf_emplace(gno_ibf* f, uint64_t v, uint64_t bin);
void     gno_ibf_emplace_many(gno_ibf* f, const uint64_t* v, const uint32_t* bins, size_t n);
/* counts[0..B) u16, zeroed then += bulk_contains(v) for each hash (GanonClassify.cpp:514) */
void gno_ibf_bulk_count(const gno_ibf* f, const uint64_t* hashes, size_t n, uint16_t* counts);
/* same, over rows gathered elsewhere: gathered[(q*h + i)*W ..) = row gno_ibf_row(f, hashes[q], i) */
void gno_ibf_bulk_count_gathered(const uint64_t* gathered, size_t n, uint32_t hash_funs, uint64_t bin_words, uint64_t bins,
                                 uint16_t* counts);

/* ---- a-7 HIBF (src/ganon-classify/include/ganon-classify/hierarchical_interleaved_bloom_filter.hpp:432-460,506-523) */
typedef struct
{
    uint32_t        n_ibf;
    const gno_ibf*  ibfs;
    const int64_t** next_ibf_id;   /* [n_ibf][bins(ibf)] */
    const int64_t** bin_to_user;   /* [n_ibf][bins(ibf)]; -1 = merged bin */
    uint64_t        n_user_bins;
} gno_hibf;
/* result[0..n_user_bins) zeroed, then filled as counting_agent_type::bulk_count(values, threshold) */
void gno_hibf_bulk_count(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold, uint16_t* result);
/* sum over all visited IBFs of n*h*W*8 (algorithmic bytes) for the last call chain */
/* value_t = uint32_t: what the reference's -DLONGREADS build instantiates (GanonClassify.cpp:45-49, hibf.hpp:438) */
void gno_hibf_bulk_count_longreads(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold, uint32_t* result);
uint64_t gno_hibf_visited_bytes(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold);

/* ---- a-6 select_matches (GanonClassify.cpp:504-541 IBF, :543-577 HIBF) over ONE filter.
 * Targets of the filter are given as CSR over bins.  match_count[global]/match_fpr[global]
 * are the TMatches of the read (count 0 == absent).  max/min follow :531-537. */
typedef struct
{
    int             is_hibf;
    const gno_ibf*  ibf;
    const gno_hibf* hibf;
    uint32_t        n_targets;
    const uint32_t* tgt_bin_off;  /* n_targets+1 */
    const uint32_t* tgt_bins;     /* bin ids (IBF: technical bins; HIBF: user bins) */
    const uint32_t* tgt_global;   /* id in the level-wide target namespace */
    const double*   tgt_fpr;      /* per target (GanonClassify.cpp:968-982 / :932) */
    double          rel_cutoff;
} gno_filter;

void gno_select_matches(const gno_filter* flt, const uint64_t* hashes, size_t n_hashes, uint64_t threshold_cutoff,
                        uint64_t* match_count, double* match_fpr, uint64_t* max_count_read, uint64_t* min_count_read,
                        uint16_t* scratch_counts /* >= max(bins, n_user_bins) */);

/* ---- a-8 filter_matches (GanonClassify.cpp:579-613; binom :498-501).
 * Scans global targets in ascending id; keep[g]=1 kept, 2 discarded by rel-filter, 3 by fpr-query.
 * Returns number kept. */
double gno_binom(double n, double k);
size_t gno_filter_matches(const uint64_t* match_count, const double* match_fpr, size_t n_global, uint64_t n_hashes,
                          uint64_t threshold_filter, double fpr_query, uint8_t* keep);

/* per-target fpr for a flat IBF (GanonClassify.cpp:940-947,968-982) */
double gno_false_positive(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t n_hashes);
double gno_target_fpr(uint64_t count, uint64_t max_hashes_bin, uint64_t bin_size_bits, uint8_t hash_functions);

/* ---- whole read against one hierarchy level (GanonClassify.cpp:676-768), for KATs and the CPU
 * baseline.  seq1/seq2 are dna4 ranks; len2 = 0 for single-end.  Returns:
 *   0 classified-candidate evaluated (see outputs), 1 skipped small (:690,743), 2 skipped big (:706,737).
 * On return match_count/match_fpr hold TMatches, keep[] the filter_matches verdicts. */
typedef struct
{
    uint64_t n_hashes, max_count_read, min_count_read, threshold_filter, n_kept;
} gno_read_result;
int gno_classify_read(const gno_filter* filters, size_t n_filters, size_t n_global, const uint8_t* seq1, size_t len1,
                      const uint8_t* seq2, size_t len2, uint32_t k, uint32_t w, double rel_filter, double fpr_query,
                      uint64_t* match_count, double* match_fpr, uint8_t* keep, gno_read_result* res,
                      uint64_t* hash_scratch, size_t hash_cap, uint16_t* count_scratch);

/* ---- a-9 LCA (src/utils/include/utils/LCA.hpp): Euler tour + sparse-table RMQ over integer ids.
 * parent[i] for node i (root: parent == itself or -1).  children are visited in the order given
 * by child_order (indices of nodes sorted the way the caller wants; LCA result is order-free). */
typedef struct gno_lca gno_lca;
gno_lca* gno_lca_build(const int32_t* parent, int32_t n_nodes, int32_t root);
int32_t  gno_lca_query(const gno_lca* l, const int32_t* nodes, int32_t n);
void     gno_lca_free(gno_lca* l);

/* ---- CPU baseline: classify a batch with OpenMP threads (one agent per thread as
 * GanonClassify.cpp:652-660,1579-1597).  bases = concatenated dna4 ranks, off[n_reads+1].
 * Single flat IBF filter with identity/CSR targets.  Writes per read n_hashes and number of
 * matches >= cutoff; returns total matches.  Used by bench.py cpu_baseline (kind "port"). */
uint64_t gno_baseline_classify(const gno_filter* flt, const uint8_t* bases, const uint64_t* off, size_t n_reads,
                               uint32_t k, uint32_t w, int threads, uint32_t* n_hashes_out, uint32_t* n_matches_out,
                               uint64_t* checksum_out);

#ifdef __cplusplus
}
#endif
#endif
