/*
 * ganon_oracle.c -- CPU restatement of ganon's read-classification hot path (plain C11).
 * TEST INFRASTRUCTURE ONLY -- see ganon_oracle.h for scope, provenance and parity status.
 * Citations are relative to /root/reference.
 */
#include "ganon_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * a-1  char -> dna4 rank.  SeqAn3 dna4::char_to_rank_table restated: A C G T
 * exact (U==T), IUPAC collapse R->A Y->C S->C W->A K->G M->A B->C D->A H->A V->A N->A,
 * case-insensitive.  Legal alphabet = dna15 (dna4_traits.hpp:15-18 inherits
 * sequence_legal_alphabet = dna15); anything else is a parse error in the reference
 * (GanonClassify.cpp:1278-1283).  The reference's test literals use '-' for "unknown, replaced by
 * A" (tests/ganon-classify/GanonClassify.test.cpp:813) -- those go through the dna4 literal
 * operator, not the file parser, so rank 0 is returned with legal=0.
 * ------------------------------------------------------------------------------------------ */
uint8_t gno_char_to_rank(unsigned char c, int* legal)
{
    int     ok = 1;
    uint8_t r  = 0;
    switch (c)
    {
        case 'A': case 'a': r = 0; break;
        case 'C': case 'c': r = 1; break;
        case 'G': case 'g': r = 2; break;
        case 'T': case 't': case 'U': case 'u': r = 3; break;
        case 'R': case 'r': r = 0; break;
        case 'Y': case 'y': r = 1; break;
        case 'S': case 's': r = 1; break;
        case 'W': case 'w': r = 0; break;
        case 'K': case 'k': r = 2; break;
        case 'M': case 'm': r = 0; break;
        case 'B': case 'b': r = 1; break;
        case 'D': case 'd': r = 0; break;
        case 'H': case 'h': r = 0; break;
        case 'V': case 'v': r = 0; break;
        case 'N': case 'n': r = 0; break;
        default: r = 0; ok = 0; break;
    }
    if (legal)
        *legal = ok;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * a-2  minimiser hash
 * ------------------------------------------------------------------------------------------ */
/* src/utils/include/utils/adjust_seed.hpp:33-37 */
uint64_t gno_adjust_seed(uint32_t k)
{
    return 0x8F3F73B5CF1C9ADEULL >> (64u - 2u * k);
}

/* seqan3::views::minimiser_hash(shape{ungapped{k}}, window_size{w}, seed{adjust_seed(k)})
 * as called at GanonClassify.cpp:647-650,693,698.  Restated from SURVEY App. A.1 / App. D:
 *  v_i = min(fwd_i ^ seed, rc_i ^ seed); windows of K = w-k+1 consecutive v; first window emits
 *  its rightmost minimum; on each slide: if the remembered minimiser just left -> recompute the
 *  rightmost minimum and emit (even when equal); else if the entering value is strictly smaller
 *  -> emit it; else nothing.  Duplicates and order are kept. */
size_t gno_minimiser_hash(const uint8_t* ranks, size_t L, uint32_t k, uint32_t w, uint64_t* out, size_t cap)
{
    if (k == 0 || k > 32 || w < k || L < w)
        return 0;
    const uint64_t seed = gno_adjust_seed(k);
    const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const size_t   M    = L - k + 1; /* k-mers */
    const size_t   K    = (size_t)w - k + 1;

    uint64_t* v = (uint64_t*)malloc(M * sizeof(uint64_t));
    uint64_t  f = 0, r = 0;
    for (size_t i = 0; i < L; ++i)
    {
        const uint64_t b = ranks[i] & 3u;
        f                = ((f << 2) | b) & mask;
        r                = (r >> 2) | ((3ULL - b) << (2 * (k - 1)));
        if (i + 1 >= k)
        {
            const uint64_t a = f ^ seed, c = r ^ seed;
            v[i + 1 - k]     = a < c ? a : c;
        }
    }

    size_t   n   = 0;
    uint64_t m   = v[0];
    size_t   pos = 0; /* offset of the remembered minimiser inside the current window */
    for (size_t i = 1; i < K; ++i)
        if (v[i] <= m) /* less_equal -> rightmost */
        {
            m   = v[i];
            pos = i;
        }
    if (n < cap)
        out[n] = m;
    ++n;
    for (size_t j = K; j < M; ++j) /* window = v[j-K+1 .. j] */
    {
        if (pos == 0)
        {
            const uint64_t* win = v + (j - K + 1);
            m                   = win[0];
            pos                 = 0;
            for (size_t i = 1; i < K; ++i)
                if (win[i] <= m)
                {
                    m   = win[i];
                    pos = i;
                }
            if (n < cap)
                out[n] = m;
            ++n;
        }
        else if (v[j] < m)
        {
            m   = v[j];
            pos = K - 1;
            if (n < cap)
                out[n] = m;
            ++n;
        }
        else
        {
            --pos;
        }
    }
    free(v);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * a-3  thresholds
 * ------------------------------------------------------------------------------------------ */
/* GanonClassify.cpp:492-495 */
uint64_t gno_threshold_rel(uint64_t n_hashes, double p)
{
    return (uint64_t)ceil((double)n_hashes * p);
}
/* GanonClassify.cpp:720-724 */
uint64_t gno_threshold_cutoff(uint64_t n_hashes, double p)
{
    uint64_t t = gno_threshold_rel(n_hashes, p);
    return t == 0 ? 1 : t;
}

/* ------------------------------------------------------------------------------------------
 * a-4 / a-5  interleaved Bloom filter  (SeqAn3 3.3.0)
 * ------------------------------------------------------------------------------------------ */
const uint64_t GNO_IBF_SEEDS[5] = { 13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL,
                                    16499269484942379435ULL, 4893150838803335377ULL };
const uint64_t GNO_IBF_MULTIPLIER = 11400714819323198485ULL; /* floor(2^64 / golden ratio); tests/test_oracle_kat.py derives all six */

uint64_t gno_ibf_hash_shift(uint64_t bin_size)
{
    return bin_size ? (uint64_t)__builtin_clzll(bin_size) : 64;
}

/* hash_and_fit without the final "* technical_bins" (that product is the bit index of the row
 * start; word index = row * bin_words). */
uint64_t gno_ibf_row(const gno_ibf* f, uint64_t v, uint32_t i)
{
    uint64_t x = v * GNO_IBF_SEEDS[i];
    x ^= x >> f->hash_shift;
    x *= GNO_IBF_MULTIPLIER;
    return (uint64_t)(((__uint128_t)x * (__uint128_t)f->bin_size) >> 64);
}

/* interleaved_bloom_filter::emplace (call site src/ganon-build/GanonBuild.cpp:694) */
void gno_ibf_emplace(gno_ibf* f, uint64_t v, uint64_t bin)
{
    for (uint32_t i = 0; i < f->hash_funs; ++i)
    {
        const uint64_t row = gno_ibf_row(f, v, i);
        f->data[row * f->bin_words + (bin >> 6)] |= 1ULL << (bin & 63);
    }
}

void gno_ibf_emplace_many(gno_ibf* f, const uint64_t* v, const uint32_t* bins, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        gno_ibf_emplace(f, v[i], bins[i]);
}

/* counting_agent::bulk_count == for each value: counts += bulk_contains(value)
 * (call site GanonClassify.cpp:514).  Bits beyond `bins` in the last word are never set by
 * emplace and never counted (counting_vector has `bins` entries). */
void gno_ibf_bulk_count(const gno_ibf* f, const uint64_t* hashes, size_t n, uint16_t* counts)
{
    memset(counts, 0, f->bins * sizeof(uint16_t));
    for (size_t q = 0; q < n; ++q)
    {
        const uint64_t* rows[5];
        for (uint32_t i = 0; i < f->hash_funs; ++i)
            rows[i] = f->data + gno_ibf_row(f, hashes[q], i) * f->bin_words;
        for (uint64_t wd = 0; wd < f->bin_words; ++wd)
        {
            uint64_t t = ~0ULL;
            for (uint32_t i = 0; i < f->hash_funs; ++i)
                t &= rows[i][wd];
            while (t)
            {
                const uint64_t bin = wd * 64 + (uint64_t)__builtin_ctzll(t);
                t &= t - 1;
                if (bin < f->bins)
                    ++counts[bin];
            }
        }
    }
}

/* The same loop over PRE-GATHERED rows: gathered[(q*h + i)*W .. +W) is the row gno_ibf_row(f, hashes[q], i) of a
 * filter that is too large to hold on the host (the sampling parity checks of BASELINE configs 4/5 fetch exactly
 * the rows their reads touch from the device).  Row selection stays with gno_ibf_row; AND + count as above. */
void gno_ibf_bulk_count_gathered(const uint64_t* gathered, size_t n, uint32_t hash_funs, uint64_t bin_words, uint64_t bins,
                                 uint16_t* counts)
{
    memset(counts, 0, bins * sizeof(uint16_t));
    for (size_t q = 0; q < n; ++q)
    {
        const uint64_t* base = gathered + q * hash_funs * bin_words;
        for (uint64_t wd = 0; wd < bin_words; ++wd)
        {
            uint64_t t = ~0ULL;
            for (uint32_t i = 0; i < hash_funs; ++i)
                t &= base[i * bin_words + wd];
            while (t)
            {
                const uint64_t bin = wd * 64 + (uint64_t)__builtin_ctzll(t);
                t &= t - 1;
                if (bin < bins)
                    ++counts[bin];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a-7  HIBF counting agent
 * hierarchical_interleaved_bloom_filter.hpp:432-460 (bulk_count_impl), :506-523 (bulk_count)
 * ------------------------------------------------------------------------------------------ */
static void hibf_impl(const gno_hibf* h, const uint64_t* hashes, size_t n, int64_t ibf_idx, uint64_t threshold,
                      uint16_t* result, uint64_t* bytes)
{
    const gno_ibf* f      = &h->ibfs[ibf_idx];
    uint16_t*      counts = (uint16_t*)malloc((f->bins ? f->bins : 1) * sizeof(uint16_t));
    gno_ibf_bulk_count(f, hashes, n, counts); /* :435-436 */
    if (bytes)
        *bytes += (uint64_t)n * f->hash_funs * f->bin_words * 8;
    uint16_t sum = 0; /* value_t, wraps (:438) */
    for (uint64_t bin = 0; bin < f->bins; ++bin)
    {
        sum                 = (uint16_t)(sum + counts[bin]);
        const int64_t fidx = h->bin_to_user[ibf_idx][bin];
        if (fidx < 0) /* merged bin :445-450 */
        {
            if ((uint64_t)sum >= threshold)
                hibf_impl(h, hashes, n, h->next_ibf_id[ibf_idx][bin], threshold, result, bytes);
            sum = 0;
        }
        else if (bin + 1 == f->bins || fidx != h->bin_to_user[ibf_idx][bin + 1]) /* :451-458 */
        {
            if ((uint64_t)sum >= threshold && result)
                result[fidx] = sum;
            sum = 0;
        }
    }
    free(counts);
}

void gno_hibf_bulk_count(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold, uint16_t* result)
{
    memset(result, 0, h->n_user_bins * sizeof(uint16_t)); /* :518 */
    hibf_impl(h, hashes, n, 0, threshold, result, NULL);  /* :520 */
}

/* The same agent as the reference's -DLONGREADS build instantiates it (GanonClassify.cpp:45-49: TIntCount = uint32_t, so
 * value_t of counting_agent_type and `sum` at hibf.hpp:438 are 32 bits wide): per-bin counts and per-user-bin sums do not
 * wrap at 2^16.  Per-bin counts: the uint32 counting_agent (A.2 bulk_count with uint32 counters). */
static void ibf_bulk_count_u32(const gno_ibf* f, const uint64_t* hashes, size_t n, uint32_t* counts)
{
    memset(counts, 0, (f->bins ? f->bins : 1) * sizeof(uint32_t));
    const uint64_t* rows[5];
    for (size_t q = 0; q < n; ++q)
    {
        for (uint32_t i = 0; i < f->hash_funs; ++i)
            rows[i] = f->data + gno_ibf_row(f, hashes[q], i) * f->bin_words;
        for (uint64_t wd = 0; wd < f->bin_words; ++wd)
        {
            uint64_t t = ~0ULL;
            for (uint32_t i = 0; i < f->hash_funs; ++i)
                t &= rows[i][wd];
            while (t)
            {
                const uint64_t bin = wd * 64 + (uint64_t)__builtin_ctzll(t);
                t &= t - 1;
                if (bin < f->bins)
                    ++counts[bin];
            }
        }
    }
}

static void hibf_impl_u32(const gno_hibf* h, const uint64_t* hashes, size_t n, int64_t ibf_idx, uint64_t threshold, uint32_t* result)
{
    const gno_ibf* f      = &h->ibfs[ibf_idx];
    uint32_t*      counts = (uint32_t*)malloc((f->bins ? f->bins : 1) * sizeof(uint32_t));
    ibf_bulk_count_u32(f, hashes, n, counts); /* :435-436 */
    uint32_t sum = 0;                         /* value_t = uint32_t (:438) */
    for (uint64_t bin = 0; bin < f->bins; ++bin)
    {
        sum += counts[bin];
        const int64_t fidx = h->bin_to_user[ibf_idx][bin];
        if (fidx < 0) /* merged bin :445-450 */
        {
            if ((uint64_t)sum >= threshold)
                hibf_impl_u32(h, hashes, n, h->next_ibf_id[ibf_idx][bin], threshold, result);
            sum = 0;
        }
        else if (bin + 1 == f->bins || fidx != h->bin_to_user[ibf_idx][bin + 1]) /* :451-458 */
        {
            if ((uint64_t)sum >= threshold)
                result[fidx] = sum;
            sum = 0;
        }
    }
    free(counts);
}

void gno_hibf_bulk_count_longreads(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold, uint32_t* result)
{
    memset(result, 0, h->n_user_bins * sizeof(uint32_t)); /* :518 */
    hibf_impl_u32(h, hashes, n, 0, threshold, result);    /* :520 */
}

uint64_t gno_hibf_visited_bytes(const gno_hibf* h, const uint64_t* hashes, size_t n, uint64_t threshold)
{
    uint64_t bytes = 0;
    hibf_impl(h, hashes, n, 0, threshold, NULL, &bytes);
    return bytes;
}

/* ------------------------------------------------------------------------------------------
 * a-6  select_matches
 * ------------------------------------------------------------------------------------------ */
void gno_select_matches(const gno_filter* flt, const uint64_t* hashes, size_t n_hashes, uint64_t threshold_cutoff,
                        uint64_t* match_count, double* match_fpr, uint64_t* max_count_read, uint64_t* min_count_read,
                        uint16_t* counts)
{
    if (!flt->is_hibf)
    {
        /* GanonClassify.cpp:504-541 */
        gno_ibf_bulk_count(flt->ibf, hashes, n_hashes, counts);
        for (uint32_t t = 0; t < flt->n_targets; ++t)
        {
            uint64_t summed = 0;
            for (uint32_t j = flt->tgt_bin_off[t]; j < flt->tgt_bin_off[t + 1]; ++j)
                summed += counts[flt->tgt_bins[j]];
            if (summed > n_hashes)
                summed = n_hashes;
            if (summed >= threshold_cutoff)
            {
                const uint32_t g = flt->tgt_global[t];
                if (summed > match_count[g])
                {
                    match_count[g] = summed;
                    match_fpr[g]   = flt->tgt_fpr ? flt->tgt_fpr[t] : 0.0;
                    if (summed > *max_count_read)
                        *max_count_read = summed;
                    if (summed < *min_count_read)
                        *min_count_read = summed;
                }
            }
        }
    }
    else
    {
        /* GanonClassify.cpp:543-577 */
        gno_hibf_bulk_count(flt->hibf, hashes, n_hashes, threshold_cutoff, counts);
        for (uint32_t t = 0; t < flt->n_targets; ++t)
        {
            const uint32_t b0 = flt->tgt_bins[flt->tgt_bin_off[t]];
            if (counts[b0] > 0)
            {
                uint64_t summed = counts[b0];
                if (summed > n_hashes)
                    summed = n_hashes;
                const uint32_t g = flt->tgt_global[t];
                if (summed > match_count[g])
                {
                    match_count[g] = summed;
                    match_fpr[g]   = flt->tgt_fpr ? flt->tgt_fpr[t] : 0.0;
                    if (summed > *max_count_read)
                        *max_count_read = summed;
                    if (summed < *min_count_read)
                        *min_count_read = summed;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a-8  filter_matches
 * ------------------------------------------------------------------------------------------ */
/* GanonClassify.cpp:498-501 */
double gno_binom(double n, double k)
{
    return exp(lgamma(n + 1) - lgamma(n - k + 1) - lgamma(k + 1));
}

/* GanonClassify.cpp:579-613.  threshold_filter arrives as size_t and is compared as double
 * (the parameter type is double, :580). */
size_t gno_filter_matches(const uint64_t* match_count, const double* match_fpr, size_t n_global, uint64_t n_hashes,
                          uint64_t threshold_filter, double fpr_query, uint8_t* keep)
{
    size_t       kept = 0;
    const double thr  = (double)threshold_filter;
    for (size_t g = 0; g < n_global; ++g)
    {
        keep[g] = 0;
        if (match_count[g] == 0)
            continue;
        if ((double)match_count[g] >= thr)
        {
            if (fpr_query < 1.0)
            {
                double q = 1;
                for (uint64_t i = 0; i <= match_count[g]; i++)
                    q -= gno_binom((double)n_hashes, (double)i) * pow(match_fpr[g], (double)i)
                         * pow(1 - match_fpr[g], (double)(n_hashes - i));
                if (q > fpr_query)
                {
                    keep[g] = 3;
                    continue;
                }
            }
            keep[g] = 1;
            ++kept;
        }
        else
        {
            keep[g] = 2;
        }
    }
    return kept;
}

/* GanonClassify.cpp:940-947 */
double gno_false_positive(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t n_hashes)
{
    return pow(1 - exp(-hash_functions / (bin_size_bits / (double)n_hashes)), hash_functions);
}

/* GanonClassify.cpp:968-982 */
double gno_target_fpr(uint64_t count, uint64_t max_hashes_bin, uint64_t bin_size_bits, uint8_t hash_functions)
{
    uint64_t n_bins_target = (uint64_t)ceil(count / (double)max_hashes_bin);
    uint64_t n_hashes_bin  = (uint64_t)ceil(count / (double)n_bins_target);
    return 1.0 - pow(1.0 - gno_false_positive(bin_size_bits, hash_functions, n_hashes_bin), (double)n_bins_target);
}

/* ------------------------------------------------------------------------------------------
 * One read against one hierarchy level: GanonClassify.cpp:676-768
 * ------------------------------------------------------------------------------------------ */
int gno_classify_read(const gno_filter* filters, size_t n_filters, size_t n_global, const uint8_t* seq1, size_t len1,
                      const uint8_t* seq2, size_t len2, uint32_t k, uint32_t w, double rel_filter, double fpr_query,
                      uint64_t* match_count, double* match_fpr, uint8_t* keep, gno_read_result* res,
                      uint64_t* hashes, size_t hash_cap, uint16_t* count_scratch)
{
    memset(match_count, 0, n_global * sizeof(uint64_t));
    memset(match_fpr, 0, n_global * sizeof(double));
    memset(keep, 0, n_global);
    memset(res, 0, sizeof(*res));

    if (len1 < w) /* :690,743-747 */
        return 1;
    size_t n = gno_minimiser_hash(seq1, len1, k, w, hashes, hash_cap); /* :693 */
    if (len2 >= w)                                                     /* :695-700 */
    {
        const size_t room = n < hash_cap ? hash_cap - n : 0;
        n += gno_minimiser_hash(seq2, len2, k, w, hashes + (n < hash_cap ? n : hash_cap), room);
    }
    res->n_hashes       = n;
    res->min_count_read = n;  /* :704 */
    if (n > 65535 || n > hash_cap) /* :674,706,737-741 (TIntCount = uint16_t) */
        return 2;

    for (size_t i = 0; i < n_filters; ++i) /* :717-735 */
    {
        const uint64_t thr = gno_threshold_cutoff(n, filters[i].rel_cutoff);
        gno_select_matches(&filters[i], hashes, n, thr, match_count, match_fpr, &res->max_count_read,
                           &res->min_count_read, count_scratch);
    }
    if (res->max_count_read > 0) /* :753-762 */
    {
        res->threshold_filter =
            res->max_count_read - gno_threshold_rel(res->max_count_read - res->min_count_read, rel_filter);
        res->n_kept = gno_filter_matches(match_count, match_fpr, n_global, n, res->threshold_filter, fpr_query, keep);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a-9  LCA: Euler tour + sparse table, src/utils/include/utils/LCA.hpp:73-174 on integer ids
 * ------------------------------------------------------------------------------------------ */
struct gno_lca
{
    int32_t  n;
    int32_t  len;      /* euler length */
    int32_t  logn;
    int32_t* euler;
    int32_t* depth;
    int32_t* first;
    int32_t* M;        /* len * logn */
};

gno_lca* gno_lca_build(const int32_t* parent, int32_t n, int32_t root)
{
    /* children lists in node-id order */
    int32_t* cnt  = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    int32_t* head = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    for (int32_t i = 0; i < n; ++i)
        if (i != root && parent[i] >= 0 && parent[i] < n && parent[i] != i)
            cnt[parent[i]]++;
    for (int32_t i = 0; i < n; ++i)
        head[i + 1] = head[i] + cnt[i];
    int32_t* kids = (int32_t*)malloc((size_t)(head[n] ? head[n] : 1) * sizeof(int32_t));
    int32_t* fill = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int32_t i = 0; i < n; ++i)
        if (i != root && parent[i] >= 0 && parent[i] < n && parent[i] != i)
            kids[head[parent[i]] + fill[parent[i]]++] = i;

    gno_lca* l = (gno_lca*)calloc(1, sizeof(gno_lca));
    l->n       = n;
    l->euler   = (int32_t*)malloc((size_t)(2 * n + 1) * sizeof(int32_t));
    l->depth   = (int32_t*)malloc((size_t)(2 * n + 1) * sizeof(int32_t));
    l->first   = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    for (int32_t i = 0; i < n; ++i)
        l->first[i] = -1;
    /* iterative DFS (LCA.hpp:73-95 is recursive) */
    int32_t* stack_node = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t* stack_it   = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t  sp         = 0;
    stack_node[0]       = root;
    stack_it[0]         = 0;
    l->first[root]      = 0;
    l->euler[0]         = root;
    l->depth[0]         = 0;
    l->len              = 1;
    while (sp >= 0)
    {
        const int32_t u = stack_node[sp];
        if (stack_it[sp] < cnt[u])
        {
            const int32_t c = kids[head[u] + stack_it[sp]++];
            ++sp;
            stack_node[sp] = c;
            stack_it[sp]   = 0;
            if (l->first[c] < 0)
                l->first[c] = l->len;
            l->euler[l->len] = c;
            l->depth[l->len] = sp;
            l->len++;
        }
        else
        {
            --sp;
            if (sp >= 0)
            {
                l->euler[l->len] = stack_node[sp];
                l->depth[l->len] = sp;
                l->len++;
            }
        }
    }
    /* sparse table LCA.hpp:105-130 */
    int32_t logn = 1;
    while ((1 << logn) <= l->len)
        ++logn;
    l->logn = logn;
    l->M    = (int32_t*)malloc((size_t)l->len * (size_t)logn * sizeof(int32_t));
    for (int32_t i = 0; i < l->len; ++i)
        l->M[(size_t)i * logn] = i;
    for (int32_t j = 1; (1 << j) <= l->len; ++j)
        for (int32_t i = 0; i + (1 << j) - 1 < l->len; ++i)
        {
            const int32_t a = l->M[(size_t)i * logn + j - 1];
            const int32_t b = l->M[(size_t)(i + (1 << (j - 1))) * logn + j - 1];
            l->M[(size_t)i * logn + j] = l->depth[a] < l->depth[b] ? a : b;
        }
    free(cnt);
    free(head);
    free(kids);
    free(fill);
    free(stack_node);
    free(stack_it);
    return l;
}

static int32_t lca_pair(const gno_lca* l, int32_t u, int32_t v)
{
    if (u == v)
        return u;
    int32_t i = l->first[u], j = l->first[v];
    if (i > j)
    {
        int32_t t = i;
        i         = j;
        j         = t;
    }
    int32_t k = 0;
    while ((1 << (k + 1)) <= j - i + 1)
        ++k;
    const int32_t a = l->M[(size_t)i * l->logn + k];
    const int32_t b = l->M[(size_t)(j - (1 << k) + 1) * l->logn + k];
    return l->euler[l->depth[a] <= l->depth[b] ? a : b];
}

/* LCA.hpp:165-174 */
int32_t gno_lca_query(const gno_lca* l, const int32_t* nodes, int32_t n)
{
    int32_t r = nodes[0];
    for (int32_t i = 1; i < n; ++i)
        r = lca_pair(l, r, nodes[i]);
    return r;
}

void gno_lca_free(gno_lca* l)
{
    if (!l)
        return;
    free(l->euler);
    free(l->depth);
    free(l->first);
    free(l->M);
    free(l);
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline ("port"): the per-read loop of GanonClassify.cpp:676-735 over one flat IBF with
 * OpenMP threads standing in for the reference's std::async workers (:1579-1597).
 * ------------------------------------------------------------------------------------------ */
uint64_t gno_baseline_classify(const gno_filter* flt, const uint8_t* bases, const uint64_t* off, size_t n_reads,
                               uint32_t k, uint32_t w, int threads, uint32_t* n_hashes_out, uint32_t* n_matches_out,
                               uint64_t* checksum_out)
{
    uint64_t total = 0, checksum = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) reduction(+ : total, checksum)
#endif
    {
        const uint64_t bins   = flt->ibf->bins;
        uint16_t*      counts = (uint16_t*)malloc(bins * sizeof(uint16_t));
        size_t         hcap   = 1024;
        uint64_t*      hashes = (uint64_t*)malloc(hcap * sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (long long r = 0; r < (long long)n_reads; ++r)
        {
            const size_t len = (size_t)(off[r + 1] - off[r]);
            uint32_t     nm  = 0;
            size_t       n   = 0;
            if (len >= w)
            {
                if (len > hcap)
                {
                    hcap   = len;
                    hashes = (uint64_t*)realloc(hashes, hcap * sizeof(uint64_t));
                }
                n = gno_minimiser_hash(bases + off[r], len, k, w, hashes, hcap);
                if (n <= 65535)
                {
                    const uint64_t thr = gno_threshold_cutoff(n, flt->rel_cutoff);
                    /* software prefetch of every row the read will touch (h rows per minimiser, random over the whole
                     * filter): without it a thread waits out one DRAM/TLB miss after the other.  Same rows, same
                     * counting as gno_ibf_bulk_count below -- only the order in which the cache lines are requested. */
                    {
                        const gno_ibf* f   = flt->ibf;
                        const size_t   rb  = (size_t)f->bin_words * 8;
                        for (size_t q = 0; q < n; ++q)
                            for (uint32_t i = 0; i < f->hash_funs; ++i)
                            {
                                const char* row = (const char*)(f->data + gno_ibf_row(f, hashes[q], i) * f->bin_words);
                                for (size_t o = 0; o < rb; o += 64)
                                    __builtin_prefetch(row + o, 0, 0);
                            }
                    }
                    gno_ibf_bulk_count(flt->ibf, hashes, n, counts);
                    for (uint32_t t = 0; t < flt->n_targets; ++t)
                    {
                        uint64_t summed = 0;
                        for (uint32_t j = flt->tgt_bin_off[t]; j < flt->tgt_bin_off[t + 1]; ++j)
                            summed += counts[flt->tgt_bins[j]];
                        if (summed > n)
                            summed = n;
                        if (summed >= thr)
                        {
                            ++nm;
                            checksum += (uint64_t)(r + 1) * 0x9E3779B97F4A7C15ULL + (uint64_t)t * 1000003ULL + summed;
                        }
                    }
                }
            }
            if (n_hashes_out)
                n_hashes_out[r] = (uint32_t)n;
            if (n_matches_out)
                n_matches_out[r] = nm;
            total += nm;
        }
        free(counts);
        free(hashes);
    }
    if (checksum_out)
        *checksum_out = checksum;
    return total;
}
