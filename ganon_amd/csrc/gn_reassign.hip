// gn_reassign.hip -- `ganon reassign` on the device: the EM over a classification's (read, target, count)
// table, /root/reference/src/ganon/reassign.py:96-145 (the loop) and :226-241 (get_top_match).
//
// The table is CSR over reads (entries of a read in the order the .all file lists them, targets numbered by first
// appearance) and stays in HBM for the whole run; what an iteration touches is
//     off[n_reads+1] (8 B / read)  +  target[n_entries] (4 B / entry)  +  prob[target] (a gather inside a table of
//     n_targets doubles, L2-resident)  +  one integer add per read with more than one entry,
// i.e. it is a streaming pass bound by HBM, 8 + 4*d bytes per read of d entries.  Nothing here is floating-point
// sensitive except the two places the reference's result depends on IEEE order, which are kept:
//   * prob = count / n  -- one correctly rounded double division per target (Python's int / int on values < 2^53);
//   * diff = sum_t |old_t - new_t| accumulated left to right in target numbering order (:125-129) -- done by ONE wave that
//     loads 64 terms at a time and adds them lane after lane, so that the stop rule `diff <= threshold` (:141) and the
//     logged value see the reference's bits.
// The choice of a read is the FIRST entry whose probability is strictly larger than every earlier one's and than 0, else
// the first entry (:226-241) -- an arg-max with first-listed tie-break; reads of up to GN_RA_LIGHT entries take one lane,
// longer ones one wave (list built once at creation).  Counts are integers and the passes compare counts, not probabilities (see
// gn_ra_pick_lane); from the second pass on a read only touches the counters when its choice changed.
#include "gn_internal.h"

#include <cstdlib>
#include <new>
#include <vector>

#define GN_RA_LIGHT 32u
#define GN_RA_BLOCK 256u

struct gn_reassign
{
    int         device = 0;
    hipStream_t st     = nullptr;
    uint64_t    n_reads = 0, n_entries = 0;
    uint32_t    n_targets = 0;
    uint64_t*   d_off    = nullptr; // n_reads + 1
    uint32_t*   d_target = nullptr; // n_entries
    uint32_t*   d_heavy  = nullptr; // reads with more than GN_RA_LIGHT entries
    unsigned long long* d_n_heavy = nullptr;
    uint64_t    n_heavy = 0;
    unsigned long long* d_uniq   = nullptr; // per target: reads that list it and nothing else (:96-103)
    unsigned long long* d_counts = nullptr; // reassigned_matches of the last iteration (:113-121), widened for the fetch call
    uint32_t*   d_w[2]   = { nullptr, nullptr }; // counts of the pass before (what a pass compares) / of the running pass
    uint32_t*   d_chosen = nullptr; // per read: the target it chose in the pass before
    double*     d_prob   = nullptr;
    double*     d_absd   = nullptr; // |old - new| per target
    double*     d_diff   = nullptr; // [0] the iteration's diff
    uint64_t*   d_choice = nullptr; // entry index per read
    uint64_t    n_unique = 0;       // reads with exactly one entry
    uint64_t    n_multi  = 0;       // reads with more than one
    uint32_t    iterations = 0;
    bool        ran = false;
    std::vector<double> diffs;
    float       ms_em = 0.f;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
};

template <typename T>
static hipError_t ra_malloc(T** p, size_t n)
{
    return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}

// ---- once per table ---------------------------------------------------------------------------------------------------
// unique counts, the heavy-read list, and the number of unique / multi reads
__global__ void __launch_bounds__(GN_RA_BLOCK) gn_ra_prepare_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ target,
                                                                   uint64_t n_reads, unsigned long long* __restrict__ uniq,
                                                                   uint32_t* __restrict__ heavy, unsigned long long* __restrict__ ctr)
{
    // ctr[0] heavy cursor, ctr[1] unique reads, ctr[2] multi reads
    uint64_t n_u = 0, n_m = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t b = off[r], e = off[r + 1];
        const uint64_t d = e - b;
        if (d == 1)
        {
            atomicAdd(&uniq[target[b]], 1ull);
            ++n_u;
        }
        else if (d > 1)
        {
            ++n_m;
            if (d > GN_RA_LIGHT)
                heavy[atomicAdd(&ctr[0], 1ull)] = (uint32_t)r;
        }
    }
    // wave totals
    for (int s = 32; s; s >>= 1)
    {
        n_u += __shfl_down(n_u, s);
        n_m += __shfl_down(n_m, s);
    }
    if ((threadIdx.x & 63) == 0)
    {
        if (n_u)
            atomicAdd(&ctr[1], (unsigned long long)n_u);
        if (n_m)
            atomicAdd(&ctr[2], (unsigned long long)n_m);
    }
}

__global__ void gn_ra_first_prob_kernel(const unsigned long long* __restrict__ uniq, uint32_t n_targets, double denom,
                                        double* __restrict__ prob)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_targets)
        prob[t] = __ddiv_rn((double)uniq[t], denom); // :106-107
}

// ---- the choice of a read (:226-241) ----------------------------------------------------------------------------------
// get_top_match compares probabilities; every probability of a pass is the same pass's count divided by one number (unique / number
// of unique reads before the first pass, count / number of reads afterwards), and a correctly rounded division by a constant keeps
// the order of integers below 2^32 apart (neighbouring quotients differ by a relative 2^-32 at least), zero stays zero.  So the
// passes compare the COUNTS the probabilities were made from -- 4-byte gathers instead of 8, no floating point in the pass.
// Entries in groups of four: the four target loads, then the four count gathers, are independent of each other, so a lane has
// eight loads in flight instead of a chain of two per entry (the pass is latency-bound otherwise).
__device__ __forceinline__ uint64_t gn_ra_pick_lane(const uint32_t* __restrict__ target, const uint32_t* __restrict__ weight, uint64_t b,
                                                    uint64_t e)
{
    uint32_t best = 0;
    uint64_t at   = b;
    uint64_t i    = b;
    for (; i + 4 <= e; i += 4)
    {
        uint32_t t[4], p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = target[i + j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            p[j] = weight[t[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (p[j] > best)
            {
                best = p[j];
                at   = i + j;
            }
    }
    if (i < e)
    {
        uint32_t t[3], p[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            t[j] = target[i + j < e ? i + j : i];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            p[j] = weight[t[j]];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (i + j < e && p[j] > best)
            {
                best = p[j];
                at   = i + j;
            }
    }
    return at;
}

// One add per chosen target and wave where lanes agree: up to GN_RA_AGG_ROUNDS times the lowest active lane's target is
// broadcast, the lanes holding the same one are counted with a ballot and leave; what is left adds on its own.  Abundance
// profiles are skewed (a few targets take most reads), which is exactly where per-lane atomics on one address serialise.
#define GN_RA_AGG_ROUNDS 4
__device__ __forceinline__ void gn_ra_add_aggregated(uint32_t* __restrict__ ctr, uint32_t t, bool active)
{
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int round = 0; round < GN_RA_AGG_ROUNDS && todo; ++round)
    {
        const int      lead = __builtin_ctzll(todo);
        const uint32_t lt   = (uint32_t)__shfl((int)t, lead);
        const unsigned long long same = __ballot(active && t == lt);
        if ((int)(threadIdx.x & 63u) == lead)
            atomicAdd(&ctr[lt], (uint32_t)__popcll(same));
        if (active && t == lt)
            active = false;
        todo &= ~same;
    }
    if (active)
        atomicAdd(&ctr[t], 1u);
}

// one wave, entries strided over its lanes: the first entry holding the read's maximum if that is positive
__device__ __forceinline__ uint64_t gn_ra_pick_wave(const uint32_t* __restrict__ target, const uint32_t* __restrict__ weight, uint64_t b,
                                                    uint64_t e, unsigned lane)
{
    uint32_t best = 0;
    uint64_t at   = ~0ull;
    for (uint64_t i = b + lane; i < e; i += 64)
    {
        const uint32_t p = weight[target[i]];
        if (p > best) // ascending i inside a lane: the lane's first entry at its maximum
        {
            best = p;
            at   = i;
        }
    }
    for (int s = 32; s; s >>= 1)
    {
        const uint32_t ob = (uint32_t)__shfl_xor((int)best, s);
        const uint64_t oa = __shfl_xor(at, s);
        if (ob > best || (ob == best && oa < at))
        {
            best = ob;
            at   = oa;
        }
    }
    return best > 0 ? at : b;
}

// MODE 0: an EM iteration (:115-121).  `counts` enters as a copy of the counts the pass before ended with (of the unique counts
//         before the first pass) and every read with several entries moves its ONE from the target it chose last time (chosen[r];
//         none before the first pass) to the one it chooses now -- after the first pass few reads change their mind, so the atomics,
//         which bound the first pass, all but vanish from the later ones.  The sums are what a recount from the unique counts gives.
// MODE 1: the final choice of every read (:153-181): the entry index; a read with one entry keeps it
template <int MODE>
__global__ void __launch_bounds__(GN_RA_BLOCK) gn_ra_pick_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ target,
                                                                const uint32_t* __restrict__ weight, uint64_t n_reads,
                                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ chosen, int first_pass,
                                                                uint64_t* __restrict__ choice)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n_reads + stride - 1) / stride; // every lane of a wave walks the same number of rounds (ballots below)
    for (uint64_t k = 0; k < rounds; ++k)
    {
        const uint64_t r  = k * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool     in = r < n_reads;
        const uint64_t b = in ? off[r] : 0, e = in ? off[r + 1] : 0;
        const uint64_t d = e - b;
        if (MODE == 0)
        {
            const bool     mine = d >= 2 && d <= GN_RA_LIGHT;
            const uint32_t t    = mine ? target[gn_ra_pick_lane(target, weight, b, e)] : 0u;
            if (first_pass)
            {
                gn_ra_add_aggregated(counts, t, mine);
                if (mine)
                    chosen[r] = t;
            }
            else if (mine)
            {
                const uint32_t was = chosen[r];
                if (was != t)
                {
                    atomicAdd(&counts[t], 1u);
                    atomicSub(&counts[was], 1u);
                    chosen[r] = t;
                }
            }
        }
        else if (in && d <= GN_RA_LIGHT)
            choice[r] = d <= 1 ? b : gn_ra_pick_lane(target, weight, b, e);
    }
}

template <int MODE>
__global__ void __launch_bounds__(GN_RA_BLOCK) gn_ra_pick_heavy_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ target,
                                                                      const uint32_t* __restrict__ weight, const uint32_t* __restrict__ heavy,
                                                                      uint64_t n_heavy, uint32_t* __restrict__ counts,
                                                                      uint32_t* __restrict__ chosen, int first_pass, uint64_t* __restrict__ choice)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_heavy; i += n_waves)
    {
        const uint32_t r  = heavy[i];
        const uint64_t at = gn_ra_pick_wave(target, weight, off[r], off[r + 1], lane);
        if (lane == 0)
        {
            if (MODE == 0)
            {
                const uint32_t t = target[at], was = first_pass ? 0xFFFFFFFFu : chosen[r];
                if (was != t)
                {
                    atomicAdd(&counts[t], 1u);
                    if (!first_pass)
                        atomicSub(&counts[was], 1u);
                    chosen[r] = t;
                }
            }
            else
                choice[r] = at;
        }
    }
}

// unique counts (64-bit, what the fetch call hands out) as the 32-bit weights of the first pass
__global__ void gn_ra_narrow_kernel(const unsigned long long* __restrict__ uniq, uint32_t n_targets, uint32_t* __restrict__ w)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_targets)
        w[t] = (uint32_t)uniq[t];
}

// ---- the update (:123-129) --------------------------------------------------------------------------------------------
__global__ void gn_ra_update_kernel(const uint32_t* __restrict__ counts, uint32_t n_targets, double total, double* __restrict__ prob,
                                    double* __restrict__ absd, unsigned long long* __restrict__ counts64)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_targets)
    {
        counts64[t]     = counts[t];
        const double np = __ddiv_rn((double)counts[t], total);
        absd[t] = fabs(__dsub_rn(prob[t], np));
        prob[t] = np;
    }
}

// diff += |old - new| for target 0, 1, 2, ... : one wave, 64 coalesced terms per step, added in lane order
__global__ void __launch_bounds__(64) gn_ra_sum_in_order_kernel(const double* __restrict__ absd, uint32_t n_targets, double* __restrict__ out)
{
    const unsigned lane = threadIdx.x;
    double         acc  = 0.0;
    for (uint32_t base = 0; base < n_targets; base += 64)
    {
        const double v = base + lane < n_targets ? absd[base + lane] : 0.0; // (+0.0 leaves a non-negative sum as it is)
#pragma unroll
        for (int j = 0; j < 64; ++j)
            acc = __dadd_rn(acc, __shfl(v, j));
    }
    if (lane == 0)
        out[0] = acc;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------------
static int ra_free(gn_reassign* g)
{
    if (!g)
        return GN_OK;
    hipSetDevice(g->device);
    if (g->st)
        hipStreamSynchronize(g->st);
    hipFree(g->d_off);
    hipFree(g->d_target);
    hipFree(g->d_heavy);
    hipFree(g->d_n_heavy);
    hipFree(g->d_uniq);
    hipFree(g->d_counts);
    hipFree(g->d_w[0]);
    hipFree(g->d_w[1]);
    hipFree(g->d_chosen);
    hipFree(g->d_prob);
    hipFree(g->d_absd);
    hipFree(g->d_diff);
    hipFree(g->d_choice);
    if (g->ev0)
        hipEventDestroy(g->ev0);
    if (g->ev1)
        hipEventDestroy(g->ev1);
    if (g->st)
        hipStreamDestroy(g->st);
    delete g;
    return GN_OK;
}

static unsigned ra_grid(uint64_t n, unsigned per_block)
{
    const uint64_t want = (n + per_block - 1) / per_block;
    return (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want)); // 16 workgroups per CU at most; the kernels stride
}

extern "C" int gn_reassign_create(int device, uint64_t n_reads, uint64_t n_entries, uint32_t n_targets, const uint64_t* off,
                                  const uint32_t* target, gn_reassign** out)
{
    if (!out)
        return gn_fail(GN_EINVAL, "gn_reassign_create: out is NULL");
    *out = nullptr;
    if (!off || (n_entries && !target))
        return gn_fail(GN_EINVAL, "gn_reassign_create: off / target is NULL");
    if (n_reads >= (1ull << 32))
        return gn_fail(GN_ERANGE, "gn_reassign_create: %llu reads (a table holds fewer than 2^32)", (unsigned long long)n_reads);
    if (off[0] != 0 || off[n_reads] != n_entries)
        return gn_fail(GN_EINVAL, "gn_reassign_create: off[0] = %llu, off[n_reads] = %llu, n_entries = %llu", (unsigned long long)off[0],
                       (unsigned long long)off[n_reads], (unsigned long long)n_entries);
    for (uint64_t r = 0; r < n_reads; ++r)
        if (off[r + 1] <= off[r]) // (the reference's dict holds a read only once it has a match: reassign.py reads .all lines)
            return gn_fail(GN_EINVAL, "gn_reassign_create: read %llu has %s", (unsigned long long)r,
                           off[r + 1] < off[r] ? "a descending offset" : "no entries (a read without matches is not part of the EM)");
    for (uint64_t i = 0; i < n_entries; ++i)
        if (target[i] >= n_targets)
            return gn_fail(GN_EINVAL, "gn_reassign_create: entry %llu names target %u of %u", (unsigned long long)i, target[i], n_targets);
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd)
        return gn_fail(GN_ENODEV, "gn_reassign_create: no HIP device %d (%d visible); there is no CPU fallback", device, nd);
    GN_HIP(hipSetDevice(device));
    gn_reassign* g = new (std::nothrow) gn_reassign();
    if (!g)
        return gn_fail(GN_ENOMEM, "gn_reassign_create: out of host memory");
    g->device    = device;
    g->n_reads   = n_reads;
    g->n_entries = n_entries;
    g->n_targets = n_targets;
    hipError_t e = hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking);
    auto       ok = [&](hipError_t x) {
        if (e == hipSuccess)
            e = x;
    };
    ok(hipEventCreate(&g->ev0));
    ok(hipEventCreate(&g->ev1));
    ok(ra_malloc(&g->d_off, n_reads + 1));
    ok(ra_malloc(&g->d_target, n_entries));
    ok(ra_malloc(&g->d_heavy, n_reads));
    ok(ra_malloc(&g->d_n_heavy, 4));
    ok(ra_malloc(&g->d_uniq, n_targets));
    ok(ra_malloc(&g->d_counts, n_targets));
    ok(ra_malloc(&g->d_w[0], n_targets));
    ok(ra_malloc(&g->d_w[1], n_targets));
    ok(ra_malloc(&g->d_chosen, n_reads));
    ok(ra_malloc(&g->d_prob, n_targets));
    ok(ra_malloc(&g->d_absd, n_targets));
    ok(ra_malloc(&g->d_diff, 1));
    ok(ra_malloc(&g->d_choice, n_reads));
    if (e != hipSuccess)
    {
        ra_free(g);
        return gn_fail(e == hipErrorOutOfMemory ? GN_ENOMEM : GN_ENODEV, "gn_reassign_create: %s", hipGetErrorString(e));
    }
    // everything below is queued on g->st, the stream every later call of this handle uses
    ok(hipMemcpyAsync(g->d_off, off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g->st));
    if (n_entries)
        ok(hipMemcpyAsync(g->d_target, target, n_entries * sizeof(uint32_t), hipMemcpyHostToDevice, g->st));
    ok(hipMemsetAsync(g->d_uniq, 0, (n_targets ? n_targets : 1) * sizeof(unsigned long long), g->st));
    ok(hipMemsetAsync(g->d_n_heavy, 0, 4 * sizeof(unsigned long long), g->st));
    if (e == hipSuccess && n_reads)
        hipLaunchKernelGGL(gn_ra_prepare_kernel, dim3(ra_grid(n_reads, GN_RA_BLOCK)), dim3(GN_RA_BLOCK), 0, g->st, g->d_off, g->d_target,
                           n_reads, g->d_uniq, g->d_heavy, g->d_n_heavy);
    unsigned long long ctr[4] = {0, 0, 0, 0};
    ok(hipMemcpyAsync(ctr, g->d_n_heavy, sizeof(ctr), hipMemcpyDeviceToHost, g->st));
    ok(hipStreamSynchronize(g->st));
    ok(hipGetLastError());
    if (e != hipSuccess)
    {
        ra_free(g);
        return gn_fail(GN_ENODEV, "gn_reassign_create: %s", hipGetErrorString(e));
    }
    g->n_heavy  = ctr[0];
    g->n_unique = ctr[1];
    g->n_multi  = ctr[2];
    *out        = g;
    return GN_OK;
}

// weight = the counts a pass compares; counts = the running pass's (MODE 0)
template <int MODE>
static void ra_launch_pick(gn_reassign* g, const uint32_t* weight, uint32_t* counts, bool first_pass)
{
    const unsigned grid = ra_grid(g->n_reads, GN_RA_BLOCK * 4);
    if (g->n_reads)
        hipLaunchKernelGGL((gn_ra_pick_kernel<MODE>), dim3(grid), dim3(GN_RA_BLOCK), 0, g->st, g->d_off, g->d_target, weight, g->n_reads, counts,
                           g->d_chosen, first_pass ? 1 : 0, g->d_choice);
    if (g->n_heavy)
        hipLaunchKernelGGL((gn_ra_pick_heavy_kernel<MODE>), dim3(ra_grid(g->n_heavy, GN_RA_BLOCK / 64)), dim3(GN_RA_BLOCK), 0, g->st, g->d_off,
                           g->d_target, weight, g->d_heavy, g->n_heavy, counts, g->d_chosen, first_pass ? 1 : 0, g->d_choice);
}

extern "C" int gn_reassign_run(gn_reassign* g, uint32_t max_iter, double threshold, uint32_t* iterations)
{
    if (!g)
        return gn_fail(GN_EINVAL, "gn_reassign_run: NULL handle");
    if (threshold != threshold) // any number is taken as the reference's argparse takes it: a negative one never stops before max_iter
        return gn_fail(GN_EINVAL, "gn_reassign_run: threshold is not a number");
    GN_HIP(hipSetDevice(g->device));
    g->diffs.clear();
    const unsigned tb = (g->n_targets + 255u) / 256u;
    // :96-107 -- prob = unique / max(1, number of unique reads)
    const double denom = g->n_unique ? (double)g->n_unique : 1.0;
    if (g->n_targets)
        hipLaunchKernelGGL(gn_ra_first_prob_kernel, dim3(tb), dim3(256), 0, g->st, g->d_uniq, g->n_targets, denom, g->d_prob);
    GN_HIP(hipEventRecord(g->ev0, g->st));
    // d_w[a]: the counts the running pass compares (before the first pass: the unique counts, whose order is that of :106-107's
    // probabilities), d_w[1 - a]: the counts it builds, starting as a copy
    int a = 0;
    if (g->n_targets)
        hipLaunchKernelGGL(gn_ra_narrow_kernel, dim3(tb), dim3(256), 0, g->st, g->d_uniq, g->n_targets, g->d_w[0]);
    uint32_t it = 0;
    for (;;)
    {
        GN_HIP(hipMemcpyAsync(g->d_w[1 - a], g->d_w[a], (size_t)g->n_targets * sizeof(uint32_t), hipMemcpyDeviceToDevice, g->st));
        ra_launch_pick<0>(g, g->d_w[a], g->d_w[1 - a], it == 0);
        if (g->n_targets)
            hipLaunchKernelGGL(gn_ra_update_kernel, dim3(tb), dim3(256), 0, g->st, g->d_w[1 - a], g->n_targets, (double)g->n_reads, g->d_prob,
                               g->d_absd, g->d_counts);
        hipLaunchKernelGGL(gn_ra_sum_in_order_kernel, dim3(1), dim3(64), 0, g->st, g->d_absd, g->n_targets, g->d_diff);
        double diff = 0.0;
        GN_HIP(hipMemcpyAsync(&diff, g->d_diff, sizeof(double), hipMemcpyDeviceToHost, g->st));
        GN_HIP(hipStreamSynchronize(g->st));
        GN_HIP(hipGetLastError());
        g->diffs.push_back(diff);
        a = 1 - a; // the counts just built are what the next pass (and the final choice) compares
        if (diff <= threshold) // :141
            break;
        if (max_iter > 0 && it == max_iter - 1) // :143
            break;
        ++it;
    }
    // :170-181 -- the choice under the probabilities of the last update
    ra_launch_pick<1>(g, g->d_w[a], nullptr, false);
    GN_HIP(hipEventRecord(g->ev1, g->st));
    GN_HIP(hipStreamSynchronize(g->st));
    GN_HIP(hipGetLastError());
    GN_HIP(hipEventElapsedTime(&g->ms_em, g->ev0, g->ev1));
    g->iterations = it + 1;
    g->ran        = true;
    if (iterations)
        *iterations = g->iterations;
    return GN_OK;
}

extern "C" int gn_reassign_diffs(const gn_reassign* g, double* diffs, uint32_t cap)
{
    if (!g || !g->ran || !diffs)
        return gn_fail(GN_EINVAL, "gn_reassign_diffs: no finished run");
    for (uint32_t i = 0; i < cap && i < g->diffs.size(); ++i)
        diffs[i] = g->diffs[i];
    return GN_OK;
}

extern "C" int gn_reassign_fetch(gn_reassign* g, uint64_t* counts, uint64_t* unique, double* prob, uint64_t* choice)
{
    if (!g || !g->ran)
        return gn_fail(GN_EINVAL, "gn_reassign_fetch: no finished run");
    GN_HIP(hipSetDevice(g->device));
    if (counts && g->n_targets)
        GN_HIP(hipMemcpyAsync(counts, g->d_counts, (size_t)g->n_targets * 8, hipMemcpyDeviceToHost, g->st));
    if (unique && g->n_targets)
        GN_HIP(hipMemcpyAsync(unique, g->d_uniq, (size_t)g->n_targets * 8, hipMemcpyDeviceToHost, g->st));
    if (prob && g->n_targets)
        GN_HIP(hipMemcpyAsync(prob, g->d_prob, (size_t)g->n_targets * 8, hipMemcpyDeviceToHost, g->st));
    if (choice && g->n_reads)
        GN_HIP(hipMemcpyAsync(choice, g->d_choice, (size_t)g->n_reads * 8, hipMemcpyDeviceToHost, g->st));
    GN_HIP(hipStreamSynchronize(g->st));
    return GN_OK;
}

extern "C" int gn_reassign_info(const gn_reassign* g, uint64_t* n_unique_reads, uint64_t* n_multi_reads, uint64_t* n_wave_reads,
                                float* ms_em, uint64_t* bytes_per_iteration)
{
    if (!g)
        return gn_fail(GN_EINVAL, "gn_reassign_info: NULL handle");
    if (n_unique_reads)
        *n_unique_reads = g->n_unique;
    if (n_multi_reads)
        *n_multi_reads = g->n_multi;
    if (n_wave_reads)
        *n_wave_reads = g->n_heavy;
    if (ms_em)
        *ms_em = g->ms_em;
    if (bytes_per_iteration) // algorithmic: every read's offset (8 B), 4 B per entry of a read with several, 4 B for its last choice
        *bytes_per_iteration = (g->n_reads + 1) * 8 + (g->n_entries - g->n_unique) * 4 + g->n_multi * 4;
    return GN_OK;
}

extern "C" int gn_reassign_free(gn_reassign* g)
{
    return ra_free(g);
}
