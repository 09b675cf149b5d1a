// gn_hibf.hip -- HIBF counting agent on the device.
//
// Reference semantics: raptor::hierarchical_interleaved_bloom_filter::counting_agent_type
//   /root/reference/src/ganon-classify/include/ganon-classify/hierarchical_interleaved_bloom_filter.hpp:432-460 (bulk_count_impl)
//   :506-523 (bulk_count) and select_matches(Filter<THIBF>) at /root/reference/src/ganon-classify/GanonClassify.cpp:543-577.
//
// The data-dependent recursion becomes a breadth-first work queue: level 0 = (read, ibf 0) for every counted
// read; a level kernel gives one wavefront to each (read, ibf) item, counts all of the read's minimisers in that
// IBF (h row words per hash fetched from HBM, AND-ed, set bits accumulated in LDS), evaluates the IBF's bin
// RUNS (a merged bin is a run of its own; a split user bin is a run of equal filename index -- exactly where the
// reference resets its running uint16 `sum`), and for runs with sum >= T either appends (read, child ibf) to
// the next level's queue or emits (read, user bin, sum).  Matches are finally radix-sorted by (read, user bin).
#include "gn_internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#define GN_WAVE 64

__constant__ uint64_t GN_HIBF_SEEDS[5] = { 13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL,
                                           16499269484942379435ULL, 4893150838803335377ULL };

__device__ __forceinline__ uint32_t gn_hibf_row(uint64_t v, uint32_t i, uint32_t shift, uint64_t S)
{
    uint64_t x = v * GN_HIBF_SEEDS[i];
    x ^= x >> shift;
    x *= 11400714819323198485ULL;
    return (uint32_t)__umul64hi(x, S);
}

__device__ __forceinline__ void gn_hibf_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GnHibfLevelParams
{
    const GnHibfIbfDev* ibfs;
    const uint64_t*     hashes;
    const uint64_t*     slot_off;
    const uint32_t*     n_hashes;
    double              rel_cutoff;
    const uint2*        work_in;
    uint32_t            n_work;
    uint2*              work_out;
    uint32_t            work_cap;
    unsigned long long* ctr;       // [0] match cursor, [2] algo bytes, [3] next-level work count
    uint64_t*           keys;      // (read << 32) | user_bin
    uint32_t*           vals;      // raw uint16 sum
    uint64_t            match_cap;
    uint32_t            lds_bins;  // LDS counters per wave
};

#define GN_HIBF_CHUNK 64u // wave-private slices of the work queue / match buffer (one global atomic per slice)

// Persistent: waves stride over the (read, ibf) items of this level.  Queue appends and matches go to wave-private
// chunks (a single counter address only sustains ~90 atomics/us); unused chunk tails hold sentinels
// (read = 0xFFFFFFFF / key = ~0) that the next level and the final sort ignore.
__global__ __launch_bounds__(256) void gn_hibf_level_kernel(GnHibfLevelParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gn_hl[];
    const int      lane   = threadIdx.x & (GN_WAVE - 1);
    const int      wave   = threadIdx.x >> 6; // (kept in a VGPR: making it an SGPR slowed this kernel, 5.2 -> 7.5 ms)
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    uint32_t*      cnt    = gn_hl + (size_t)wave * p.lds_bins;

    unsigned long long my_bytes = 0, my_matches = 0;
    unsigned long long wq_base = 0, mq_base = 0; // current chunk cursors (wave-uniform)
    uint32_t           wq_left = 0, mq_left = 0;

    for (uint32_t item = blockIdx.x * (blockDim.x >> 6) + wave; item < p.n_work; item += nwaves)
    {
        const uint2    wk   = p.work_in[item];
        const uint32_t read = wk.x;
        if (read == 0xFFFFFFFFu) // hole left by a chunked append of the previous level
            continue;
        const GnHibfIbfDev f  = p.ibfs[wk.y];
        const uint32_t     n  = p.n_hashes[read];
        const uint64_t*    hs = p.hashes + p.slot_off[read];
        const uint32_t     TB = f.W * 64;

        gn_hibf_wave_sync();
        for (uint32_t i = lane; i < TB; i += GN_WAVE)
            cnt[i] = 0;
        gn_hibf_wave_sync();

        // lanes = (hash sub-index, word): Gp lanes cover the W words of a row (W <= 64), or the row is walked in
        // 64-word chunks with one hash per iteration (W > 64)
        uint32_t gp_log2 = 0;
        while ((1u << gp_log2) < f.W && gp_log2 < 6)
            ++gp_log2;
        const uint32_t Gp   = 1u << gp_log2;
        const uint32_t H    = GN_WAVE >> gp_log2;
        const uint32_t gl   = lane & (Gp - 1);
        const uint32_t hsub = lane >> gp_log2;

        for (uint32_t q0 = 0; q0 < n; q0 += H)
        {
            const uint32_t q = q0 + hsub;
            if (q < n)
            {
                const uint64_t v = hs[q];
                uint32_t       rows[5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    rows[i] = (uint32_t)i < f.h ? gn_hibf_row(v, i, f.shift, f.S) : 0u;
                for (uint32_t wd = gl; wd < f.W; wd += Gp)
                {
                    uint64_t m = ~0ULL;
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if ((uint32_t)i < f.h)
                            m &= f.rows[(uint64_t)rows[i] * f.W + wd];
                    while (m)
                    {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        atomicAdd(&cnt[wd * 64 + b], 1u);
                    }
                }
            }
        }
        gn_hibf_wave_sync();

        // threshold_cutoff = max(1, ceil(n * rel_cutoff))  (GanonClassify.cpp:492-495,720-724); passed to bulk_count (:553)
        uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
        if (T == 0)
            T = 1;

        for (uint32_t r0 = 0; r0 < f.n_runs; r0 += GN_WAVE)
        {
            const uint32_t r = r0 + lane;
            bool           hit = false, merged = false;
            uint32_t       sum = 0;
            int32_t        tgt = 0;
            if (r < f.n_runs)
            {
                const uint4 run = f.runs[r]; // first bin, n bins, user bin (-1 merged), child ibf
                for (uint32_t b = 0; b < run.y; ++b)
                    sum = (sum + cnt[run.x + b]) & 0xFFFFu; // value_t = uint16_t wraps (hibf.hpp:438,442)
                merged = (int32_t)run.z < 0;
                tgt    = merged ? (int32_t)run.w : (int32_t)run.z;
                hit    = sum >= T; // :447 / :455
            }
            // merged bins -> next level queue
            const uint64_t mm = __ballot(hit && merged);
            if (mm)
            {
                const uint32_t need = (uint32_t)__popcll(mm);
                if (need > wq_left)
                {
                    const uint32_t     take = need > GN_HIBF_CHUNK ? need : GN_HIBF_CHUNK;
                    unsigned long long nb   = 0;
                    if (lane == 0)
                        nb = atomicAdd(&p.ctr[3], (unsigned long long)take);
                    nb = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(nb >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
                    for (uint32_t i = lane; i < take; i += GN_WAVE) // sentinels first; real entries overwrite them
                        if (nb + i < p.work_cap)
                            p.work_out[nb + i] = make_uint2(0xFFFFFFFFu, 0u);
                    wq_base = nb;
                    wq_left = take;
                }
                if (hit && merged)
                {
                    const unsigned long long o = wq_base + __popcll(mm & ((1ULL << lane) - 1ULL));
                    if (o < p.work_cap)
                        p.work_out[o] = make_uint2(read, (uint32_t)tgt);
                }
                wq_base += need;
                wq_left -= need;
            }
            // leaf runs -> matches
            const uint64_t lm = __ballot(hit && !merged);
            if (lm)
            {
                const uint32_t need = (uint32_t)__popcll(lm);
                if (need > mq_left)
                {
                    const uint32_t     take = need > GN_HIBF_CHUNK ? need : GN_HIBF_CHUNK;
                    unsigned long long nb   = 0;
                    if (lane == 0)
                        nb = atomicAdd(&p.ctr[0], (unsigned long long)take);
                    nb = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(nb >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
                    for (uint32_t i = lane; i < take; i += GN_WAVE)
                        if (nb + i < p.match_cap)
                            p.keys[nb + i] = ~0ULL; // sentinel: sorts last
                    mq_base = nb;
                    mq_left = take;
                }
                if (hit && !merged)
                {
                    const unsigned long long o = mq_base + __popcll(lm & ((1ULL << lane) - 1ULL));
                    if (o < p.match_cap)
                    {
                        p.keys[o] = ((uint64_t)read << 32) | (uint32_t)tgt;
                        p.vals[o] = sum;
                    }
                }
                mq_base += need;
                mq_left -= need;
                my_matches += need;
            }
        }
        my_bytes += (unsigned long long)n * f.h * f.W * 8ull; // algorithmic bytes of this visit
    }
    if (lane == 0)
    {
        if (my_bytes)
            atomicAdd(&p.ctr[2], my_bytes);
        if (my_matches)
            atomicAdd(&p.ctr[6], my_matches);
    }
}

__global__ void gn_hibf_seed_kernel(uint2* work, const uint8_t* status, uint32_t n_reads, unsigned long long* count)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool     ok = r < n_reads && status[r] == GN_READ_OK;
    const uint64_t bm = __ballot(ok);
    const int      lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0 && bm)
        base = atomicAdd(count, (unsigned long long)__popcll(bm));
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    if (ok)
        work[base + __popcll(bm & ((1ULL << lane) - 1ULL))] = make_uint2(r, 0u);
}

// sorted (key, raw sum) -> gn_match with the cap of select_matches (GanonClassify.cpp:561-564) + per-read histogram
__global__ void gn_hibf_finish_kernel(const uint64_t* keys, const uint32_t* vals, uint64_t n, const uint32_t* n_hashes,
                                      gn_match* out, uint32_t* seg_count)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (keys[i] == ~0ULL) // chunk hole (sorted to the end)
        return;
    const uint32_t read = (uint32_t)(keys[i] >> 32);
    const uint32_t nh   = n_hashes[read];
    gn_match m;
    m.read   = read;
    m.target = (uint32_t)keys[i];
    m.count  = vals[i] > nh ? nh : vals[i];
    out[i]   = m;
    atomicAdd(&seg_count[read], 1u);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int gn_hibf_build(gn_filter* f, uint32_t n_ibf, const gn_ibf_desc* ibfs, const int64_t* const* next_ibf_id,
                  const int64_t* const* bin2userbin, uint64_t n_user_bins)
{
    std::vector<GnHibfIbfDev> dev(n_ibf);
    uint32_t                  max_tb = 0;
    for (uint32_t i = 0; i < n_ibf; ++i)
    {
        const uint64_t B = ibfs[i].bins;
        std::vector<uint4> runs;
        uint64_t b = 0;
        while (b < B)
        {
            const int64_t u = bin2userbin[i][b];
            if (u >= (int64_t)n_user_bins)
                return gn_fail(GN_EINVAL, "ibf %u bin %llu: user bin %lld out of range", i, (unsigned long long)b, (long long)u);
            if (u < 0)
            {
                const int64_t c = next_ibf_id[i][b];
                if (c < 0 || c >= (int64_t)n_ibf || c == (int64_t)i)
                    return gn_fail(GN_EINVAL, "ibf %u bin %llu: merged bin without a valid child ibf (%lld)", i,
                                   (unsigned long long)b, (long long)c);
                runs.push_back(make_uint4((uint32_t)b, 1u, 0xFFFFFFFFu, (uint32_t)c));
                ++b;
            }
            else
            {
                uint64_t e = b + 1;
                while (e < B && bin2userbin[i][e] == u)
                    ++e;
                runs.push_back(make_uint4((uint32_t)b, (uint32_t)(e - b), (uint32_t)u, 0u));
                b = e;
            }
        }
        uint4* d_runs = nullptr;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&d_runs), std::max<size_t>(1, runs.size()) * sizeof(uint4)));
        f->hibf_allocs.push_back(d_runs);
        if (!runs.empty())
            GN_HIP(hipMemcpy(d_runs, runs.data(), runs.size() * sizeof(uint4), hipMemcpyHostToDevice));
        dev[i].rows   = f->ibfs[i].d_rows;
        dev[i].S      = f->ibfs[i].S;
        dev[i].W      = (uint32_t)f->ibfs[i].W;
        dev[i].B      = (uint32_t)f->ibfs[i].B;
        dev[i].shift  = f->ibfs[i].shift;
        dev[i].h      = f->ibfs[i].h;
        dev[i].runs   = d_runs;
        dev[i].n_runs = (uint32_t)runs.size();
        max_tb        = std::max(max_tb, dev[i].W * 64u);
    }
    if ((size_t)max_tb * 4 > 144 * 1024)
        return gn_fail(GN_ERANGE, "an IBF of the HIBF has %u technical bins; the level kernel supports up to 36864", max_tb);
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&f->d_hibf), n_ibf * sizeof(GnHibfIbfDev)));
    GN_HIP(hipMemcpy(f->d_hibf, dev.data(), n_ibf * sizeof(GnHibfIbfDev), hipMemcpyHostToDevice));
    f->n_user_bins = n_user_bins;
    f->max_bins    = max_tb;
    return GN_OK;
}

static int gn_hibf_ensure_sort_buffers(gn_stream* s)
{
    if (s->hibf_cap >= s->match_cap && s->d_keys[0])
        return GN_OK;
    for (int i = 0; i < 2; ++i)
    {
        if (s->d_keys[i])
            hipFree(s->d_keys[i]);
        if (s->d_vals[i])
            hipFree(s->d_vals[i]);
        s->d_keys[i] = nullptr;
        s->d_vals[i] = nullptr;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_keys[i]), std::max<uint64_t>(1, s->match_cap) * 8));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_vals[i]), std::max<uint64_t>(1, s->match_cap) * 4));
    }
    if (s->d_sort_tmp)
        hipFree(s->d_sort_tmp);
    s->d_sort_tmp = nullptr;
    size_t tmp    = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, s->d_keys[0], s->d_keys[1], s->d_vals[0], s->d_vals[1],
                                       (int)std::min<uint64_t>(s->match_cap, 0x7FFFFFFFull), 0, 64, s->st);
    s->sort_tmp_bytes = tmp + 256;
    GN_HIP(hipMalloc(&s->d_sort_tmp, s->sort_tmp_bytes));
    s->hibf_cap = s->match_cap;
    return GN_OK;
}

// Runs all levels, then sorts/group matches.  Synchronises the stream once per level (queue sizes are
// data dependent) -- HIBFs are a handful of levels deep.
int gn_hibf_classify(gn_stream* s, gn_filter* f, hipStream_t st)
{
    int rc = gn_hibf_ensure_sort_buffers(s);
    if (rc)
        return rc;
    const uint32_t n = s->n_reads;
    GN_HIP(hipMemsetAsync(s->d_ctr + 2, 0, 2 * sizeof(unsigned long long), st)); // algo bytes, work count
    GN_HIP(hipMemsetAsync(s->d_ctr + 6, 0, sizeof(unsigned long long), st));     // exact match count
    GN_HIP(hipMemsetAsync(s->d_seg_count, 0, ((size_t)n + 1) * 4, st));
    if (n)
        hipLaunchKernelGGL(gn_hibf_seed_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s->d_work[0], s->d_status, n,
                           s->d_ctr + 3);
    int      cur   = 0;
    uint32_t depth = 0;
    while (true)
    {
        GN_HIP(hipStreamSynchronize(st));
        GN_HIP(hipMemcpy(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        uint64_t n_work = s->h_ctr[3];
        if (n_work == 0)
            break;
        if (n_work > s->work_cap)
        {
            // the queue overflowed: grow both queues and restart the whole batch
            for (int i = 0; i < 2; ++i)
            {
                hipFree(s->d_work[i]);
                s->d_work[i] = nullptr;
            }
            s->work_cap = (uint32_t)std::min<uint64_t>(n_work + n_work / 4 + 1024, 0xFFFFFFF0ull);
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_work[0]), (size_t)s->work_cap * sizeof(uint2)));
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_work[1]), (size_t)s->work_cap * sizeof(uint2)));
            GN_HIP(hipMemsetAsync(s->d_ctr, 0, sizeof(unsigned long long), st));
            return gn_hibf_classify(s, f, st);
        }
        if (++depth > 64)
            return gn_fail(GN_EINVAL, "HIBF deeper than 64 levels (cycle in next_ibf_id?)");
        GN_HIP(hipMemsetAsync(s->d_ctr + 3, 0, sizeof(unsigned long long), st));
        GnHibfLevelParams p{};
        p.ibfs       = f->d_hibf;
        p.hashes     = s->d_hashes;
        p.slot_off   = s->d_slot_off;
        p.n_hashes   = s->d_nh;
        p.rel_cutoff = s->rel_cutoff;
        p.work_in    = s->d_work[cur];
        p.n_work     = (uint32_t)n_work;
        p.work_out   = s->d_work[cur ^ 1];
        p.work_cap   = s->work_cap;
        p.ctr        = s->d_ctr;
        p.keys       = s->d_keys[0];
        p.vals       = s->d_vals[0];
        p.match_cap  = s->match_cap;
        p.lds_bins   = f->max_bins;
        const uint32_t wpb = (size_t)f->max_bins * 4 * 4 <= 64 * 1024 ? 4 : 1; // waves per block by LDS need
        const size_t   lds = (size_t)f->max_bins * 4 * wpb;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gn_hibf_level_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        uint32_t blocks = (p.n_work + wpb - 1) / wpb;
        const uint32_t max_blocks = (uint32_t)f->n_cu * (wpb == 4 ? 8u : 16u);
        if (blocks > max_blocks)
            blocks = max_blocks;
        hipLaunchKernelGGL(gn_hibf_level_kernel, dim3(blocks), dim3(wpb * 64), lds, st, p);
        GN_HIP(hipGetLastError());
        cur ^= 1;
    }
    // group: radix sort by (read, user bin), cap counts, histogram per read, exclusive scan
    const uint64_t nm = s->h_ctr[0];
    if (nm > s->match_cap)
        return GN_OK; // gn_finish() sees the overflow, grows the buffers and re-runs
    if (nm)
    {
        if (nm > 0x7FFFFFFFull)
            return gn_fail(GN_ERANGE, "more than 2^31 matches in one batch");
        size_t tmp = s->sort_tmp_bytes;
        GN_HIP(hipcub::DeviceRadixSort::SortPairs(s->d_sort_tmp, tmp, s->d_keys[0], s->d_keys[1], s->d_vals[0], s->d_vals[1],
                                                  (int)nm, 0, 64, st));
        hipLaunchKernelGGL(gn_hibf_finish_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, s->d_keys[1],
                           s->d_vals[1], nm, s->d_nh, s->d_sorted, s->d_seg_count);
    }
    size_t tmp = s->scan_tmp_bytes;
    GN_HIP(hipcub::DeviceScan::ExclusiveSum(s->d_scan_tmp, tmp, s->d_seg_count, s->d_seg_off, (int)(n + 1), st));
    return GN_OK;
}

// dense tap: uint16[n_user_bins] per read == counting_agent_type::bulk_count(values, T) (raw, uncapped sums)
int gn_hibf_dense(gn_stream* s, uint32_t rb, uint32_t re, uint16_t* counts)
{
    gn_filter*            f  = s->f;
    const uint64_t        nm = s->n_matches;
    std::vector<uint64_t> keys(nm ? nm : 1);
    std::vector<uint32_t> vals(nm ? nm : 1);
    if (nm)
    {
        GN_HIP(hipMemcpy(keys.data(), s->d_keys[1], nm * 8, hipMemcpyDeviceToHost));
        GN_HIP(hipMemcpy(vals.data(), s->d_vals[1], nm * 4, hipMemcpyDeviceToHost));
    }
    std::fill(counts, counts + (size_t)(re - rb) * f->n_user_bins, (uint16_t)0);
    for (uint64_t i = 0; i < nm; ++i)
    {
        const uint32_t r = (uint32_t)(keys[i] >> 32);
        if (r >= rb && r < re)
            counts[(size_t)(r - rb) * f->n_user_bins + (uint32_t)keys[i]] = (uint16_t)vals[i];
    }
    return GN_OK;
}
