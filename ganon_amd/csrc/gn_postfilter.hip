// gn_postfilter.hip -- device-side pre-pass of filter_matches (/root/reference/src/ganon-classify/GanonClassify.cpp:579-613
// with the threshold of :755-761), run on the matches of a batch after they were grouped by read.
//
// Why it exists: at a low cutoff (--rel-cutoff 0.2, the binary's default, with --rel-filter 0.1 --fpr-query 1e-5) a read has ~100
// chance matches on a 4096-bin filter (T = 4 of ~18 minimisers), and nearly all of them die in filter_matches.  Shipping
// them over PCIe and walking them on one host thread made the end-to-end rate 2 Mreads/s; dropping them here, in HBM,
// leaves the host the survivors only.
//
// What it computes, per read with matches (count_i, target_i), n = number of minimisers:
//   max = max_i count_i, min = min(n, min_i count_i)                         (:704 and select_matches' tracking)
//   threshold_filter = max - ceil((max - min) * rel_filter)                   (:757-758; IEEE double multiply + ceil)
//   count_i < threshold_filter                      -> dropped (--rel-filter), counted
//   fpr_query < 1 and q_i SURELY > fpr_query        -> dropped (--fpr-query), counted
//   everything else                                 -> kept, original order; if q_i is SURELY <= fpr_query the match is
//                                                      marked (GN_MATCH_FPR_OK in its count) and the host skips its own check
// The --rel-filter rule is exact integer/double arithmetic and is applied completely.  The --fpr-query rule
// (q = 1 - sum_{i<=count} C(n,i) p^i (1-p)^(n-i), libm lgamma/exp/pow on the host) is applied CONSERVATIVELY: the device
// evaluates q with a multiplicative recurrence and drops a match only if q > fpr_query*1.001 + 1e-9 (and marks it as
// passed only if q < fpr_query*0.999 - 1e-9), a margin far above both evaluations' error (<= ~1e-11 for n <= 4096, the only range the device decides; longer reads, degenerate p and
// underflowing terms are left alone).  The host applies the reference's own expression to every survivor, so the final
// result is the one the reference computes; the device only removes matches whose fate is not in doubt.
//
// One stream's pass (MODE 0) is valid only where one filter (one gn_filter) sees all matches of a read.  A hierarchy level
// with several filters merges their matches before thresholding (:716-735), and a filter cut into column parts spreads a
// read over several streams: gn_streams_postfilter_joint below runs the rule over all the streams of a level -- max/min
// per stream, combined, applied (filters with disjoint targets, column parts), or the reference's merge replayed per read
// (filters that share targets).  The count kernels take part too: what the --rel-filter rule is bound to drop they do not
// write in the first place (GnCountParams::pre_mode; seg_min / pre_ctr carry what this pass still needs of it).
#include "gn_internal.h"
#include <hipcub/hipcub.hpp>
#include "gn_scan.h"

#define GN_PF_SMALL 6u
#define GN_PF_MERGE_WAVE 512u  // matches of a read over a level's filters that one wave merges (LDS table of 1024 slots)
#define GN_PF_MERGE_CAP 4096u  // ... that a block merges (8192 slots); beyond that the caller does the read
#define GN_MATCH_REMOVED 0x40000000u // (internal: entry dropped by the level merge; compacted away before anything is fetched)

struct GnPostfilterParams
{
    gn_match*           m;          // grouped matches, filtered in place (survivors move to the front of the read's range)
    const uint64_t*     off;        // read r owns [off[r*stride], off[(r+1)*stride])
    const uint64_t*     begin;      // segmented result (gn_run_group): read r's off[r+1] - off[r] matches lie at m + begin[r]; nullptr: at m + off[r*stride]
    uint32_t            stride;
    uint32_t            n_reads;
    const uint32_t*     nh;         // minimisers per read
    double              rel_filter;
    double              fpr_query;  // >= 1: rule off
    const double*       tfpr;       // per target
    uint32_t*           keep;       // [n_reads+1] survivors per read (keep[n_reads] = 0)
    uint32_t*           maxc;       // [n_reads] max count before filtering (0 = no match)
    uint32_t*           minc;       // [n_reads] STATS mode: min count (starts at the read's minimiser count, :704)
    const uint32_t*     gmax;       // APPLY mode: the level's max / min per read (several filters, see gn_streams_postfilter_joint)
    const uint32_t*     gmin;
    unsigned long long* ctr;        // [0] dropped by rel_filter [1] dropped by fpr_query
    const unsigned long long* cursor; // match cursor and capacity: an overflowed batch holds no matches yet (gn_finish re-runs)
    uint64_t            cap;
    uint64_t            n_targets;
    const uint32_t*     seg_min;    // per (read, column slice): smallest count the count kernel left unwritten (nullptr: none did)
    const unsigned long long* pre_ctr; // how many it left unwritten: they count as dropped by rel_filter
};

// 0 = the host decides, 1 = q is surely above fpr_query (drop), 2 = q is surely not above it (keep, no host check needed)
__device__ __forceinline__ uint32_t gn_fpr_verdict(uint32_t n, uint32_t count, double p, double fq)
{
    if (!(p > 0.0) || !(p < 1.0) || n > 4096u || count > 1024u)
        return 0;
    double term = exp((double)n * log1p(-p)); // i = 0
    if (!(term > 1e-200))
        return 0;
    const double ratio = p / (1.0 - p);
    double       sum   = term;
    for (uint32_t i = 1; i <= count; ++i)
    {
        term *= (double)(n - i + 1) / (double)i * ratio;
        sum += term;
    }
    const double q = 1.0 - sum;
    if (q > fq * 1.001 + 1e-9)
        return 1;
    if (q < fq * 0.999 - 1e-9)
        return 2;
    return 0;
}

// MODE 0: max/min of the read's own matches, then the rules (one filter per level)
// MODE 1: max/min only (-> maxc, minc): the first half of the joint pass over a level's filters
// MODE 2: the rules with the level's max/min given (gmax, gmin): its second half
// MODE 3: sweep after gn_pf_merge_kernel (levels whose filters share targets): entries marked GN_MATCH_REMOVED go
__device__ __forceinline__ void gn_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_postfilter_kernel(GnPostfilterParams p)
{
    __shared__ gn_match stage[4][128]; // per wave: entries of a heavy read waiting for their --fpr-query verdict
    const uint32_t      wv = threadIdx.x >> 6;
    if (*p.cursor > p.cap)
        return;
    const uint64_t r     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane  = threadIdx.x & 63u;
    const bool     valid = r < p.n_reads;
    const bool     fpr_on = p.fpr_query < 1.0;
    uint64_t       o = 0;
    uint32_t       c = 0, n = 0;
    uint32_t       n_fil = 0, n_fpr = 0;
    if (valid)
    {
        o = p.off[r * p.stride];
        c = (uint32_t)(p.off[(r + 1) * p.stride] - o);
        if (p.begin)
            o = p.begin[r];
        n = p.nh[r];
    }
    if (valid && c <= GN_PF_SMALL)
    {
        uint32_t mx = 0, mn = n;
        if constexpr (MODE == 2)
        {
            mx = p.gmax[r];
            mn = p.gmin[r];
        }
        else if constexpr (MODE != 3)
        {
            for (uint32_t j = 0; j < c; ++j)
            {
                const uint32_t ct = p.m[o + j].count;
                mx = ct > mx ? ct : mx;
                mn = ct < mn ? ct : mn;
            }
            if (p.seg_min)
                for (uint32_t x = 0; x < p.stride; ++x)
                {
                    const uint32_t sm = p.seg_min[r * p.stride + x];
                    mn = sm < mn ? sm : mn;
                }
        }
        if constexpr (MODE == 1)
        {
            p.maxc[r] = mx;
            p.minc[r] = mn;
        }
        else if constexpr (MODE == 3) // the level merge marked what goes: sweep
        {
            uint32_t kept = 0;
            for (uint32_t j = 0; j < c; ++j)
            {
                const gn_match m = p.m[o + j];
                if (!(m.count & GN_MATCH_REMOVED))
                    p.m[o + kept++] = m;
            }
            p.keep[r] = kept;
        }
        else
        {
            uint32_t kept = 0;
            if (c)
            {
                const uint32_t thr = gn_pf_threshold(mx, mn, p.rel_filter);
                for (uint32_t j = 0; j < c; ++j)
                {
                    gn_match       m = p.m[o + j];
                    const uint32_t v = m.count >= thr && fpr_on
                                           ? gn_fpr_verdict(n, m.count, m.target < p.n_targets ? p.tfpr[m.target] : 0.0, p.fpr_query)
                                           : 0u;
                    if (m.count < thr)
                        ++n_fil;
                    else if (v == 1)
                        ++n_fpr;
                    else
                    {
                        m.count |= v == 2 ? GN_MATCH_FPR_OK : 0u;
                        p.m[o + kept++] = m;
                    }
                }
            }
            p.keep[r] = kept;
            p.maxc[r] = mx;
        }
    }
    if (MODE != 1 && r == p.n_reads)
    {
        p.keep[r] = 0;
        if (MODE != 3 && p.pre_ctr && *p.pre_ctr)
            atomicAdd(&p.ctr[0], *p.pre_ctr);
    }
    uint64_t heavy = __ballot(valid && c > GN_PF_SMALL);
    while (heavy)
    {
        const uint32_t L = (uint32_t)__builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const uint64_t rr = r - lane + L;
        const uint64_t oo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), (int)L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, (int)L);
        const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)L);
        const uint32_t nn = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)L);
        uint32_t mx = 0, mn = nn;
        if constexpr (MODE == 3)
        {
            uint32_t kept = 0;
            for (uint32_t j0 = 0; j0 < cc; j0 += 64)
            {
                const uint32_t j   = j0 + lane;
                const bool     act = j < cc;
                gn_match       m{};
                if (act)
                    m = p.m[oo + j];
                const bool     k  = act && !(m.count & GN_MATCH_REMOVED);
                const uint64_t km = __ballot(k);
                if (k)
                    p.m[oo + kept + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = m;
                kept += (uint32_t)__popcll(km);
            }
            if (lane == 0)
                p.keep[rr] = kept;
            continue;
        }
        else if constexpr (MODE == 2)
        {
            mx = p.gmax[rr];
            mn = p.gmin[rr];
        }
        else
        {
            for (uint32_t j = lane; j < cc; j += 64)
            {
                const uint32_t ct = p.m[oo + j].count;
                mx = ct > mx ? ct : mx;
                mn = ct < mn ? ct : mn;
            }
            if (p.seg_min)
                for (uint32_t x = lane; x < p.stride; x += 64)
                {
                    const uint32_t sm = p.seg_min[rr * p.stride + x];
                    mn = sm < mn ? sm : mn;
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
            {
                const uint32_t a = (uint32_t)__shfl_xor((int)mx, off), b = (uint32_t)__shfl_xor((int)mn, off);
                mx = a > mx ? a : mx;
                mn = b < mn ? b : mn;
            }
        }
        if constexpr (MODE == 1)
        {
            if (lane == 0)
            {
                p.maxc[rr] = mx;
                p.minc[rr] = mn;
            }
            continue;
        }
        const uint32_t thr  = gn_pf_threshold(mx, mn, p.rel_filter);
        uint32_t       kept = 0;
        if (fpr_on)
        {
            // the --fpr-query verdict is a loop of `count` steps: entries that pass the --rel-filter rule (a few of many) are
            // first gathered in LDS, then judged 64 at a time with every lane busy
            uint32_t ns = 0;
            auto     judge = [&](uint32_t take) {
                gn_wave_sync_lds();
                gn_match m{};
                bool     k = false;
                if (lane < take)
                {
                    m = stage[wv][lane];
                    const uint32_t v = gn_fpr_verdict(nn, m.count, m.target < p.n_targets ? p.tfpr[m.target] : 0.0, p.fpr_query);
                    if (v == 1)
                        ++n_fpr;
                    else
                    {
                        k = true;
                        m.count |= v == 2 ? GN_MATCH_FPR_OK : 0u;
                    }
                }
                gn_match rest{};
                if (take + lane < ns)
                    rest = stage[wv][take + lane];
                gn_wave_sync_lds();
                if (take + lane < ns)
                    stage[wv][lane] = rest;
                ns -= take;
                const uint64_t km = __ballot(k);
                if (k) // (kept never passes the number of entries loaded so far: in place is safe)
                    p.m[oo + kept + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = m;
                kept += (uint32_t)__popcll(km);
            };
            for (uint32_t j0 = 0; j0 < cc; j0 += 64)
            {
                const uint32_t j   = j0 + lane;
                const bool     act = j < cc;
                gn_match       m{};
                if (act)
                    m = p.m[oo + j];
                const bool pass = act && m.count >= thr;
                n_fil += act && !pass;
                const uint64_t pm = __ballot(pass);
                if (pass)
                    stage[wv][ns + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = m;
                ns += (uint32_t)__popcll(pm);
                if (ns >= 64)
                    judge(64);
            }
            if (ns)
                judge(ns);
        }
        else
            for (uint32_t j0 = 0; j0 < cc; j0 += 64)
            {
                const uint32_t j   = j0 + lane;
                const bool     act = j < cc;
                gn_match       m{};
                if (act)
                    m = p.m[oo + j];
                const bool k = act && m.count >= thr;
                n_fil += act && !k;
                const uint64_t km = __ballot(k);
                if (k) // all loads of this chunk are done (the store data depends on them) and kept <= j0: in place is safe
                    p.m[oo + kept + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = m;
                kept += (uint32_t)__popcll(km);
            }
        if (lane == 0)
        {
            p.keep[rr] = kept;
            p.maxc[rr] = mx;
        }
    }
    // batch totals: one atomic per wave and counter
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        n_fil += (uint32_t)__shfl_xor((int)n_fil, off);
        n_fpr += (uint32_t)__shfl_xor((int)n_fpr, off);
    }
    if (lane == 0)
    {
        if (n_fil)
            atomicAdd(&p.ctr[0], (unsigned long long)n_fil);
        if (n_fpr)
            atomicAdd(&p.ctr[1], (unsigned long long)n_fpr);
    }
}

// survivors of read r: in[off[r*stride] + i], i < keep[r]  ->  out[new_off[r] + i]
__global__ void gn_postfilter_compact_kernel(const gn_match* __restrict__ in, gn_match* __restrict__ out, const uint64_t* __restrict__ off,
                                             const uint64_t* __restrict__ begin, uint32_t stride, const uint32_t* __restrict__ keep, const uint64_t* __restrict__ new_off,
                                             uint32_t n_reads, const unsigned long long* __restrict__ cursor, uint64_t cap)
{
    if (*cursor > cap)
        return;
    const uint64_t r     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane  = threadIdx.x & 63u;
    const bool     valid = r < n_reads;
    uint64_t       o = 0, no = 0;
    uint32_t       c = 0;
    if (valid)
    {
        o  = begin ? begin[r] : off[r * stride];
        no = new_off[r];
        c  = keep[r];
    }
    if (valid && c <= GN_PF_SMALL)
        for (uint32_t j = 0; j < c; ++j)
            out[no + j] = in[o + j];
    uint64_t heavy = __ballot(valid && c > GN_PF_SMALL);
    while (heavy)
    {
        const uint32_t L = (uint32_t)__builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const uint64_t oo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), (int)L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, (int)L);
        const uint64_t nn = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(no >> 32), (int)L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)no, (int)L);
        const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)L);
        for (uint32_t j = lane; j < cc; j += 64)
            out[nn + j] = in[oo + j];
    }
}

// queued on the stream right after the grouping pass; results: s->d_matches (compacted survivors), s->d_slot_cnt (n+1
// offsets), s->d_pf_max, s->d_pf_ctr [0] dropped rel_filter [1] dropped fpr_query [2] survivors
static GnPostfilterParams gn_pf_params(gn_stream* s)
{
    GnPostfilterParams p{};
    // a segmented result is judged where it lies (one read of every match, no copy before the pre-pass)
    const bool seg = s->segmented && !s->compacted && !s->f->is_hibf;
    p.m          = seg ? s->d_matches : s->d_sorted;
    p.off        = s->d_seg_off;
    p.begin      = seg ? s->d_seg_begin : nullptr;
    p.stride     = s->f->is_hibf ? 1u : (uint32_t)s->f->geom.wpr;
    p.n_reads    = s->n_reads;
    p.nh         = s->v_nh;
    p.rel_filter = s->pf_rel_filter;
    p.fpr_query  = s->pf_fpr_query;
    p.tfpr       = s->d_pf_fpr;
    p.keep       = s->d_pf_keep;
    p.maxc       = s->d_pf_max;
    p.minc       = s->d_pf_min;
    p.ctr        = s->d_pf_ctr;
    p.cursor     = s->d_ctr;
    p.cap        = s->match_cap;
    p.n_targets  = s->f->is_hibf ? s->f->n_user_bins : s->f->n_targets;
    p.seg_min    = s->pf_predrop ? s->d_pf_segmin : nullptr;
    p.pre_ctr    = s->pf_predrop ? s->d_pf_pre : nullptr;
    return p;
}

// survivors' offsets (scan of keep[]) and their compaction into d_matches; totals to the pinned copy
static int gn_pf_finish(gn_stream* s, const GnPostfilterParams& p)
{
    const uint32_t n      = s->n_reads;
    const unsigned blocks = (unsigned)(((uint64_t)n + 1 + 255) / 256);
    size_t         tmp    = s->pf_scan_bytes;
    GN_HIP(gn_scan_counts(s->d_pf_scan, tmp, s->d_pf_keep, s->d_slot_cnt, (int)(n + 1), s->st));
    // survivors go to the buffer the pre-pass did not read: d_matches after a contiguous copy, d_sorted after segments
    gn_match* out = p.m == s->d_sorted ? s->d_matches : s->d_sorted;
    hipLaunchKernelGGL(gn_postfilter_compact_kernel, dim3(blocks), dim3(256), 0, s->st, p.m, out, s->d_seg_off, p.begin, p.stride,
                       s->d_pf_keep, s->d_slot_cnt, n, s->d_ctr, s->match_cap);
    s->pf_out = out;
    GN_HIP(hipGetLastError());
    GN_HIP(hipMemcpyAsync(s->d_pf_ctr + 2, s->d_slot_cnt + n, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->st));
    GN_HIP(hipMemcpyAsync(s->h_pf_ctr, s->d_pf_ctr, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->st));
    return GN_OK;
}

// queued on the stream right after the grouping pass; results: s->d_matches (compacted survivors), s->d_slot_cnt (n+1
// offsets), s->d_pf_max, s->d_pf_ctr [0] dropped rel_filter [1] dropped fpr_query [2] survivors
int gn_run_postfilter(gn_stream* s)
{
    if (!s->pf_on || s->pf_joint) // (a joint pass is run by gn_streams_postfilter_joint, after every stream of the level)
        return GN_OK;
    GN_HIP(hipMemsetAsync(s->d_pf_ctr, 0, 4 * sizeof(unsigned long long), s->st));
    const GnPostfilterParams p = gn_pf_params(s);
    const unsigned blocks = (unsigned)(((uint64_t)s->n_reads + 1 + 255) / 256);
    hipLaunchKernelGGL(gn_postfilter_kernel<0>, dim3(blocks), dim3(256), 0, s->st, p);
    GN_HIP(hipGetLastError());
    return gn_pf_finish(s, p);
}

struct GnPfLists
{
    const uint32_t* mx[GN_PF_MAX_JOINT];
    const uint32_t* mn[GN_PF_MAX_JOINT];
    uint32_t        k;
};
__global__ void gn_pf_combine_kernel(GnPfLists l, uint32_t n, uint32_t* __restrict__ gmax, uint32_t* __restrict__ gmin)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n)
        return;
    uint32_t a = l.mx[0][r], b = l.mn[0][r];
    for (uint32_t i = 1; i < l.k; ++i)
    {
        const uint32_t x = l.mx[i][r], y = l.mn[i][r];
        a = x > a ? x : a;
        b = y < b ? y : b;
    }
    gmax[r] = a;
    gmin[r] = b;
}

// ---- levels whose filters SHARE targets ----------------------------------------------------------------------------------
// The reference merges a read's matches filter after filter (GanonClassify.cpp:531-537): an entry replaces the target's entry
// only if its count is larger, and max_count_read / min_count_read are updated only by entries that got in.  Replayed here per
// read (gn_pf_merge_kernel below): in the end
//   inserted(e) = count(e) > max(count of the earlier filters' entries),   winner = the largest count, earliest filter on ties
// max = max over all counts, min = min(n_hashes, counts of inserted entries); the rules of filter_matches then apply to the
// winners (one per target -- what the merged map holds), everything else is removed.  A read with more than GN_PF_MERGE_CAP
// matches over all filters is left untouched and flagged (bit 31 of its max_count): the host does that read itself.

struct GnPfMergeParams
{
    gn_match*       m[GN_PF_MAX_JOINT];
    const uint64_t* off[GN_PF_MAX_JOINT];
    uint32_t        stride[GN_PF_MAX_JOINT];
    const uint32_t* gid[GN_PF_MAX_JOINT];  // device target -> level-wide target id
    const double*   fpr[GN_PF_MAX_JOINT];
    uint64_t        n_targets[GN_PF_MAX_JOINT];
    uint32_t*       maxc[GN_PF_MAX_JOINT];
    uint32_t        k;
    uint32_t        n_reads;
    const uint32_t* nh;
    double          rel_filter, fpr_query;
    unsigned long long* ctr; // stream 0's: [0] dropped rel_filter [1] dropped fpr_query [3] length of `big`
    uint32_t*           big; // reads with more matches than a wave's table takes: done by the block-per-read launch
};

// The merged map of the reference is an open-addressing table in LDS, keyed by level-wide target id: filter after filter
// (barrier in between) every entry looks its target up, gets in if its count beats what is there (count, entry index), and
// min follows the entries that got in; what the table holds in the end are the winners.  BLOCK = false: a wave per read, 1024
// slots, reads up to GN_PF_MERGE_WAVE matches; larger ones are listed for the BLOCK = true launch (a block per read, 8192
// slots, up to GN_PF_MERGE_CAP matches).
template <bool BLOCK>
__global__ __launch_bounds__(256) void gn_pf_merge_kernel(GnPfMergeParams p)
{
    constexpr uint32_t SLOTS = BLOCK ? 2u * GN_PF_MERGE_CAP : 2u * GN_PF_MERGE_WAVE, G = BLOCK ? 256u : 64u, NG = BLOCK ? 1u : 4u;
    __shared__ uint32_t           tkey[NG][SLOTS];
    __shared__ unsigned long long tval[NG][SLOTS];
    __shared__ uint32_t           red[2][4];
    const uint32_t lane = BLOCK ? threadIdx.x : (threadIdx.x & 63u), grp = BLOCK ? 0u : (threadIdx.x >> 6);
    const bool     fpr_on = p.fpr_query < 1.0;
    uint32_t       n_fil = 0, n_fpr = 0;
    auto           sync = [&]() {
        if constexpr (BLOCK)
            __syncthreads();
        else
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    const uint32_t n_items = BLOCK ? (uint32_t)p.ctr[3] : p.n_reads;
    for (uint32_t it = BLOCK ? blockIdx.x : blockIdx.x * 4 + grp; it < n_items; it += BLOCK ? gridDim.x : gridDim.x * 4)
    {
        const uint32_t r = BLOCK ? p.big[it] : it;
        // entries of the read: filter f's segment is [o_f, o_f + c_f)
        uint64_t o[GN_PF_MAX_JOINT];
        uint32_t c[GN_PF_MAX_JOINT], total = 0;
        for (uint32_t f = 0; f < p.k; ++f)
        {
            o[f] = p.off[f][(uint64_t)r * p.stride[f]];
            c[f] = (uint32_t)(p.off[f][(uint64_t)(r + 1) * p.stride[f]] - o[f]);
            total += c[f];
        }
        const uint32_t n = p.nh[r];
        if (total == 0)
        {
            if (lane == 0)
                for (uint32_t f = 0; f < p.k; ++f)
                    p.maxc[f][r] = 0;
            continue;
        }
        if (total > SLOTS / 2)
        {
            if (!BLOCK && lane == 0) // flagged for the caller unless the block-per-read launch takes it
            {
                for (uint32_t f = 0; f < p.k; ++f)
                    p.maxc[f][r] = 0x80000000u;
                p.big[atomicAdd(&p.ctr[3], 1ull)] = r;
            }
            continue;
        }
        uint32_t used = 64, shift = 26;
        while (used < 2 * total)
        {
            used <<= 1;
            --shift;
        }
        for (uint32_t x = lane; x < used; x += G)
            tkey[grp][x] = 0;
        sync();
        uint32_t mx = 0, mn = n, base = 0;
        for (uint32_t f = 0; f < p.k; ++f)
        {
            for (uint32_t j = lane; j < c[f]; j += G)
            {
                const gn_match mt = p.m[f][o[f] + j];
                const uint32_t g1 = p.gid[f][mt.target] + 1u;
                uint32_t       h = (g1 * 0x9E3779B1u) >> shift, existing = 0;
                for (uint32_t step = 0; step < used; ++step)
                {
                    const uint32_t prev = atomicCAS(&tkey[grp][h], 0u, g1);
                    if (prev == 0)
                        break;
                    if (prev == g1) // (an earlier filter's entry: a filter names a target once)
                    {
                        existing = (uint32_t)(tval[grp][h] >> 32);
                        break;
                    }
                    h = (h + 1) & (used - 1);
                }
                mx = mt.count > mx ? mt.count : mx;
                if (mt.count > existing)
                {
                    tval[grp][h] = ((unsigned long long)mt.count << 32) | (base + j);
                    mn           = mt.count < mn ? mt.count : mn;
                }
            }
            base += c[f];
            sync();
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1)
        {
            const uint32_t a = (uint32_t)__shfl_xor((int)mx, sh), b = (uint32_t)__shfl_xor((int)mn, sh);
            mx = a > mx ? a : mx;
            mn = b < mn ? b : mn;
        }
        if constexpr (BLOCK)
        {
            if ((threadIdx.x & 63u) == 0)
            {
                red[0][threadIdx.x >> 6] = mx;
                red[1][threadIdx.x >> 6] = mn;
            }
            __syncthreads();
            for (uint32_t x = 0; x < 4; ++x)
            {
                mx = red[0][x] > mx ? red[0][x] : mx;
                mn = red[1][x] < mn ? red[1][x] : mn;
            }
        }
        const uint32_t thr = gn_pf_threshold(mx, mn, p.rel_filter);
        base = 0;
        for (uint32_t f = 0; f < p.k; ++f)
        {
            for (uint32_t j = lane; j < c[f]; j += G)
            {
                gn_match*      rec = p.m[f] + o[f] + j;
                const gn_match mt = *rec;
                const uint32_t g1 = p.gid[f][mt.target] + 1u;
                uint32_t       h = (g1 * 0x9E3779B1u) >> shift;
                for (uint32_t step = 0; step < used && tkey[grp][h] != g1; ++step)
                    h = (h + 1) & (used - 1);
                const bool winner = (uint32_t)tval[grp][h] == base + j; // what the merged map holds for this target
                uint32_t   out = mt.count;
                if (!winner)
                    out |= GN_MATCH_REMOVED;
                else if (mt.count < thr)
                {
                    out |= GN_MATCH_REMOVED;
                    ++n_fil;
                }
                else if (fpr_on)
                {
                    const uint32_t v = gn_fpr_verdict(n, mt.count, mt.target < p.n_targets[f] ? p.fpr[f][mt.target] : 0.0, p.fpr_query);
                    if (v == 1)
                    {
                        out |= GN_MATCH_REMOVED;
                        ++n_fpr;
                    }
                    else if (v == 2)
                        out |= GN_MATCH_FPR_OK;
                }
                rec->count = out;
            }
            base += c[f];
        }
        if (lane == 0)
            for (uint32_t f = 0; f < p.k; ++f)
                p.maxc[f][r] = mx;
        sync(); // (the table is cleared for the next read)
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1)
    {
        n_fil += (uint32_t)__shfl_xor((int)n_fil, sh);
        n_fpr += (uint32_t)__shfl_xor((int)n_fpr, sh);
    }
    if ((threadIdx.x & 63u) == 0)
    {
        if (n_fil)
            atomicAdd(&p.ctr[0], (unsigned long long)n_fil);
        if (n_fpr)
            atomicAdd(&p.ctr[1], (unsigned long long)n_fpr);
    }
}

int gn_finish_batch(gn_stream* s); // gn_capi.hip: waits for the batch, re-runs it after a match-buffer overflow

// Several filters of one hierarchy level, the same batch classified against each (one stream per filter, one device):
// the reference merges their matches before it thresholds (GanonClassify.cpp:716-735,755-761), so max/min are the level's.
// With DISJOINT targets across the filters (no target is reported twice) the merge is a union and the rules can run per
// filter once every stream knows the level's max/min per read: max/min per stream -> combined -> rules per stream.
extern "C" int gn_streams_postfilter_joint(gn_stream* const* streams, uint32_t n_streams)
{
    if (!streams || n_streams == 0 || n_streams > GN_PF_MAX_JOINT * GN_PF_MAX_DEVICES)
        return gn_fail(GN_EINVAL, "gn_streams_postfilter_joint: 1..%u streams", (unsigned)(GN_PF_MAX_JOINT * GN_PF_MAX_DEVICES));
    gn_stream* s0 = streams[0];
    bool       one_device = true;
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        gn_stream* s = streams[i];
        if (!s || !s->pf_on || !s->pf_joint)
            return gn_fail(GN_EINVAL, "stream %u has no joint post-filter set", i);
        if (s->n_reads != s0->n_reads)
            return gn_fail(GN_EINVAL, "streams of a joint pass must hold the same batch");
        one_device = one_device && s->device == s0->device;
    }
    const uint32_t n      = s0->n_reads;
    const unsigned blocks = (unsigned)(((uint64_t)n + 1 + 255) / 256);
    bool           merge  = false;
    for (uint32_t i = 0; i < n_streams; ++i)
        merge = merge || streams[i]->pf_merge;
    if (merge)
    {
        // filters that share targets: one kernel replays the level's merge per read over all streams, then every stream
        // sweeps what was marked
        if (!one_device || n_streams > GN_PF_MAX_JOINT)
            return gn_fail(GN_EINVAL, "a merging joint pass (filters that share targets) takes up to %u streams on one device", (unsigned)GN_PF_MAX_JOINT);
        GN_HIP(hipSetDevice(s0->device));
        GnPfMergeParams mp{};
        mp.k = n_streams;
        for (uint32_t i = 0; i < n_streams; ++i)
        {
            gn_stream* s = streams[i];
            if (!s->pf_merge || !s->d_pf_gid)
                return gn_fail(GN_EINVAL, "stream %u of a merging joint pass has no target_gid table", i);
            int rc = gn_finish_batch(s);
            if (rc)
                return rc;
            GN_HIP(hipMemsetAsync(s->d_pf_ctr, 0, 4 * sizeof(unsigned long long), s->st));
            if ((rc = gn_result_compact(s)) != GN_OK) // (the merge kernel walks contiguous per-read ranges)
                return rc;
            mp.m[i]         = s->d_sorted;
            mp.off[i]       = s->d_seg_off;
            mp.stride[i]    = s->f->is_hibf ? 1u : (uint32_t)s->f->geom.wpr;
            mp.gid[i]       = s->d_pf_gid;
            mp.fpr[i]       = s->d_pf_fpr;
            mp.n_targets[i] = s->f->is_hibf ? s->f->n_user_bins : s->f->n_targets;
            mp.maxc[i]      = s->d_pf_max;
        }
        mp.n_reads    = n;
        mp.nh         = s0->v_nh;
        mp.rel_filter = s0->pf_rel_filter;
        mp.fpr_query  = s0->pf_fpr_query;
        mp.ctr        = s0->d_pf_ctr;
        mp.big        = s0->d_pf_min; // (a merging pass has no use for the per-stream minima: the buffer holds the list)
        for (uint32_t i = 0; i < n_streams; ++i)
            GN_HIP(hipStreamSynchronize(streams[i]->st));
        if (n)
        {
            unsigned mb = (n + 3) / 4;
            if (mb > 8192u)
                mb = 8192u;
            hipLaunchKernelGGL(gn_pf_merge_kernel<false>, dim3(mb), dim3(256), 0, s0->st, mp);
            hipLaunchKernelGGL(gn_pf_merge_kernel<true>, dim3(n < 2048u ? n : 2048u), dim3(256), 0, s0->st, mp);
        }
        GN_HIP(hipGetLastError());
        GN_HIP(hipStreamSynchronize(s0->st));
        for (uint32_t i = 0; i < n_streams; ++i)
        {
            gn_stream*               s = streams[i];
            const GnPostfilterParams p = gn_pf_params(s);
            hipLaunchKernelGGL(gn_postfilter_kernel<3>, dim3(blocks), dim3(256), 0, s->st, p);
            GN_HIP(hipGetLastError());
            int rc = gn_pf_finish(s, p);
            if (rc)
                return rc;
            s->pf_joint_done = true;
        }
        return GN_OK;
    }
    // Disjoint targets.  The streams may sit on several devices (the column parts of a bin-range partitioned filter,
    // SURVEY 8e): max/min per stream -> combined per device on that device's first stream (its leader) -> every leader
    // gets every other leader's pair of arrays, device to device (8 bytes per read and pair of devices over xGMI) ->
    // combined again -> rules per stream with the level's values.
    struct Group
    {
        gn_stream*            leader;
        std::vector<uint32_t> idx;
    };
    std::vector<Group> groups;
    const bool apart = gn_sw().joint_apart; // tests on one GPU: every stream is treated as a device of its own
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        size_t g = 0;
        while (g < groups.size() && (apart || groups[g].leader->device != streams[i]->device))
            ++g;
        if (g == groups.size())
            groups.push_back(Group{ streams[i], {} });
        groups[g].idx.push_back(i);
    }
    if (groups.size() > GN_PF_MAX_DEVICES)
        return gn_fail(GN_EINVAL, "a joint pass spans at most %u devices", (unsigned)GN_PF_MAX_DEVICES);
    for (auto const& g : groups)
        if (g.idx.size() > GN_PF_MAX_JOINT)
            return gn_fail(GN_EINVAL, "a joint pass takes at most %u streams per device", (unsigned)GN_PF_MAX_JOINT);
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        gn_stream* s  = streams[i];
        int        rc = gn_finish_batch(s); // (a match buffer that overflowed is grown and the batch re-run first)
        if (rc)
            return rc;
        GN_HIP(hipSetDevice(s->device));
        GN_HIP(hipMemsetAsync(s->d_pf_ctr, 0, 4 * sizeof(unsigned long long), s->st));
        hipLaunchKernelGGL(gn_postfilter_kernel<1>, dim3(blocks), dim3(256), 0, s->st, gn_pf_params(s));
        GN_HIP(hipGetLastError());
    }
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        GN_HIP(hipSetDevice(streams[i]->device));
        GN_HIP(hipStreamSynchronize(streams[i]->st));
    }
    const size_t   ng   = groups.size();
    const uint64_t slot = (uint64_t)n; // entries of one array
    for (auto& g : groups)
    {
        gn_stream* L = g.leader;
        GN_HIP(hipSetDevice(L->device));
        uint32_t *lmax = L->d_pf_gmax, *lmin = L->d_pf_gmin;
        if (ng > 1)
        {
            const uint64_t need = 2ull * ((uint64_t)L->max_reads + 1) * ng;
            if (L->pf_peer_cap < need)
            {
                if (L->d_pf_peer)
                    GN_HIP(hipFree(L->d_pf_peer));
                L->d_pf_peer   = nullptr;
                L->pf_peer_cap = 0;
                GN_HIP(hipMalloc(reinterpret_cast<void**>(&L->d_pf_peer), need * 4));
                L->pf_peer_cap = need;
            }
            lmax = L->d_pf_peer;
            lmin = L->d_pf_peer + slot;
        }
        GnPfLists lists{};
        lists.k = (uint32_t)g.idx.size();
        for (size_t j = 0; j < g.idx.size(); ++j)
        {
            lists.mx[j] = streams[g.idx[j]]->d_pf_max;
            lists.mn[j] = streams[g.idx[j]]->d_pf_min;
        }
        if (n)
            hipLaunchKernelGGL(gn_pf_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, L->st, lists, n, lmax, lmin);
        GN_HIP(hipGetLastError());
    }
    for (auto& g : groups)
    {
        GN_HIP(hipSetDevice(g.leader->device));
        GN_HIP(hipStreamSynchronize(g.leader->st));
    }
    if (ng > 1)
    {
        for (size_t a = 0; a < ng; ++a)
        {
            gn_stream* L = groups[a].leader;
            GN_HIP(hipSetDevice(L->device));
            GnPfLists lists{};
            lists.k     = (uint32_t)ng;
            lists.mx[0] = L->d_pf_peer;
            lists.mn[0] = L->d_pf_peer + slot;
            size_t k    = 1;
            for (size_t b = 0; b < ng; ++b)
            {
                if (b == a)
                    continue;
                gn_stream* R   = groups[b].leader;
                uint32_t*  dst = L->d_pf_peer + 2 * slot * k;
                gn_peer_enable(L->device, R->device);
                if (n) // [max | min] of the other device in one copy
                    GN_HIP(hipMemcpyPeerAsync(dst, L->device, R->d_pf_peer, R->device, 2 * slot * 4, L->st));
                lists.mx[k] = dst;
                lists.mn[k] = dst + slot;
                ++k;
            }
            if (n)
                hipLaunchKernelGGL(gn_pf_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, L->st, lists, n, L->d_pf_gmax, L->d_pf_gmin);
            GN_HIP(hipGetLastError());
        }
        for (auto& g : groups)
        {
            GN_HIP(hipSetDevice(g.leader->device));
            GN_HIP(hipStreamSynchronize(g.leader->st));
        }
    }
    for (auto& g : groups)
        for (uint32_t i : g.idx)
        {
            gn_stream* s = streams[i];
            GN_HIP(hipSetDevice(s->device));
            GnPostfilterParams p = gn_pf_params(s);
            p.gmax = g.leader->d_pf_gmax;
            p.gmin = g.leader->d_pf_gmin;
            hipLaunchKernelGGL(gn_postfilter_kernel<2>, dim3(blocks), dim3(256), 0, s->st, p);
            GN_HIP(hipGetLastError());
            int rc = gn_pf_finish(s, p);
            if (rc)
                return rc;
            s->pf_joint_done = true;
        }
    // (a leader's gmax/gmin are read by the other streams' kernels on its device: nothing reuses them before every stream
    //  is fetched, and a fetch waits for its stream)
    return GN_OK;
}

extern "C" int gn_stream_set_postfilter(gn_stream* s, const gn_postfilter* pf)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (!pf)
    {
        s->pf_on = false;
        return GN_OK;
    }
    if (!(pf->rel_filter >= 0.0 && pf->rel_filter <= 1.0))
        return gn_fail(GN_EINVAL, "rel_filter must be within [0,1]");
    if (!(pf->fpr_query >= 0.0))
        return gn_fail(GN_EINVAL, "fpr_query must not be negative");
    const uint64_t nt = s->f->is_hibf ? s->f->n_user_bins : s->f->n_targets;
    if (pf->fpr_query < 1.0 && !pf->target_fpr)
        return gn_fail(GN_EINVAL, "target_fpr is required when fpr_query < 1");
    GN_HIP(hipSetDevice(s->f->device));
    if (!s->d_pf_keep)
    {
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_keep), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_max), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_min), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_gmax), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_gmin), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_ctr), 4 * sizeof(unsigned long long)));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_pre), 2 * sizeof(unsigned long long)));
        s->pf_segmin_cap = ((uint64_t)s->max_reads + 1) * (s->f->is_hibf ? 1u : s->f->geom.wpr);
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_segmin), s->pf_segmin_cap * 4));
        if (s->f->is_hibf)
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_rmax), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_fpr), (nt ? nt : 1) * sizeof(double)));
        GN_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_pf_ctr), 4 * sizeof(unsigned long long), hipHostMallocDefault));
        size_t tmp = 0;
        gn_scan_counts(nullptr, tmp, s->d_pf_keep, s->d_slot_cnt, (int)(s->max_reads + 1), s->st);
        GN_HIP(hipMalloc(&s->d_pf_scan, tmp ? tmp : 1));
        s->pf_scan_bytes = tmp;
    }
    GN_HIP(hipStreamSynchronize(s->st)); // (a batch in flight still reads the previous table)
    if (pf->target_fpr && nt)
        GN_HIP(hipMemcpy(s->d_pf_fpr, pf->target_fpr, nt * sizeof(double), hipMemcpyHostToDevice));
    else if (nt)
    {
        // (stream-ordered: s->st is a non-blocking stream, a null-stream fill would be unordered against its kernels)
        GN_HIP(hipMemsetAsync(s->d_pf_fpr, 0, nt * sizeof(double), s->st));
        GN_HIP(hipStreamSynchronize(s->st));
    }
    s->pf_rel_filter = pf->rel_filter;
    s->pf_fpr_query  = pf->fpr_query;
    s->pf_joint      = pf->joint != 0;
    s->pf_merge      = pf->joint == 2;
    if (s->pf_merge)
    {
        if (!pf->target_gid)
            return gn_fail(GN_EINVAL, "joint = 2 needs target_gid");
        if (!s->d_pf_gid)
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_pf_gid), (nt ? nt : 1) * sizeof(uint32_t)));
        for (uint64_t t = 0; t < nt; ++t)
            if (pf->target_gid[t] >= (1u << 28))
                return gn_fail(GN_ERANGE, "level-wide target ids must be below 2^28");
        if (nt)
            GN_HIP(hipMemcpy(s->d_pf_gid, pf->target_gid, nt * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    s->pf_on         = true;
    return GN_OK;
}

void gn_postfilter_release(gn_stream* s)
{
    void* ptrs[] = { s->d_pf_keep, s->d_pf_max, s->d_pf_min, s->d_pf_gmax, s->d_pf_gmin, s->d_pf_ctr, s->d_pf_fpr, s->d_pf_scan, s->d_pf_gid,
                     s->d_pf_segmin, s->d_pf_pre, s->d_pf_rmax, s->d_pf_peer };
    for (void* q : ptrs)
        if (q)
            hipFree(q);
    if (s->h_pf_ctr)
        hipHostFree(s->h_pf_ctr);
    s->d_pf_keep = s->d_pf_max = s->d_pf_min = s->d_pf_gmax = s->d_pf_gmin = s->d_pf_gid = nullptr;
    s->d_pf_ctr  = nullptr;
    s->d_pf_segmin = nullptr;
    s->pf_segmin_cap = 0;
    s->d_pf_pre  = nullptr;
    s->d_pf_rmax = nullptr;
    s->d_pf_peer = nullptr;
    s->pf_peer_cap = 0;
    s->pf_predrop = false;
    s->d_pf_fpr  = nullptr;
    s->d_pf_scan = nullptr;
    s->h_pf_ctr  = nullptr;
}
