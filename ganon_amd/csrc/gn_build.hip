// gn_build.hip -- build-side device work behind the C ABI (the ganon-build half of the minimiser/IBF code):
//   gn_stream_distinct_hashes  the set of minimiser hashes of the sequences resident in a stream, sorted ascending
//                              (count_hashes' robin_hood::unordered_set per file, /root/reference/src/ganon-build/GanonBuild.cpp:184-249)
//   gn_filter_emplace_split    insert a target's hashes into its run of technical bins, equal shares per bin
//                              (create_bin_map_hash :619-653 + build :655-698: hash i of the target goes to bin
//                              first_bin + i / hashes_per_bin)
// HBM-bound integer work: a radix sort (hipCUB) over the hash array, a unique pass, an atomic-OR scatter.
#include "gn_internal.h"
#include <hipcub/hipcub.hpp>

// seqan3::interleaved_bloom_filter hash seeds and hash_and_fit, as in gn_kernels.hip
__constant__ uint64_t GN_BUILD_SEEDS[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST;   // include/ganon_ibf_hash.h
__device__ __forceinline__ uint32_t gn_build_row(uint64_t v, uint32_t i, uint32_t shift, uint64_t S)
{
    uint64_t x = v * GN_BUILD_SEEDS[i];
    x ^= x >> shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint32_t)__umul64hi(x, S);
}

// hashes live in per-read slots (one slot per window, gn_slot_count_kernel); read r used the first nh[r] of them
__global__ void gn_pack_hashes_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ slot_off,
                                      const uint32_t* __restrict__ nh, uint32_t n_reads, uint64_t* __restrict__ out,
                                      unsigned long long* __restrict__ cursor)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= n_reads)
        return;
    const uint32_t     c = nh[wave];
    const uint64_t     b = slot_off[wave];
    unsigned long long o = 0;
    if (lane == 0 && c)
        o = atomicAdd(cursor, (unsigned long long)c); // order is irrelevant: the array is sorted next
    o = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(o >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o);
    for (uint32_t j = lane; j < c; j += 64)
        out[o + j] = hashes[b + j];
}

static int gn_build_reserve(gn_stream* s, uint64_t n)
{
    if (s->build_cap >= n && s->d_build[0])
        return GN_OK;
    for (auto& p : s->d_build)
    {
        if (p)
            hipFree(p);
        p = nullptr;
    }
    if (s->d_build_tmp)
        hipFree(s->d_build_tmp);
    s->d_build_tmp = nullptr;
    const uint64_t cap = n + n / 8 + 1024;
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_build[0]), cap * 8));
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_build[1]), cap * 8));
    if (!s->d_build_ctr)
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_build_ctr), 2 * sizeof(unsigned long long)));
    size_t a = 0, b = 0;
    hipcub::DeviceRadixSort::SortKeys(nullptr, a, s->d_build[0], s->d_build[1], (int)cap, 0, 64, s->st);
    hipcub::DeviceSelect::Unique(nullptr, b, s->d_build[1], s->d_build[0], s->d_build_ctr + 1, (int)cap, s->st);
    s->build_tmp_bytes = a > b ? a : b;
    GN_HIP(hipMalloc(&s->d_build_tmp, s->build_tmp_bytes ? s->build_tmp_bytes : 1));
    s->build_cap = cap;
    return GN_OK;
}

void gn_build_release(gn_stream* s)
{
    for (auto& p : s->d_build)
    {
        if (p)
            hipFree(p);
        p = nullptr;
    }
    if (s->d_build_tmp)
        hipFree(s->d_build_tmp);
    if (s->d_build_ctr)
        hipFree(s->d_build_ctr);
    s->d_build_tmp = nullptr;
    s->d_build_ctr = nullptr;
    s->build_cap   = 0;
}

extern "C" int gn_stream_distinct_hashes(gn_stream* s, uint64_t* out, uint64_t cap, uint64_t* n_distinct)
{
    if (!s || !n_distinct)
        return gn_fail(GN_EINVAL, "null argument");
    if (!s->hashed)
        return gn_fail(GN_EINVAL, "no minimisers computed on this stream");
    GN_HIP(hipSetDevice(s->device));
    *n_distinct = 0;
    const uint32_t n = s->n_reads;
    if (n == 0)
        return GN_OK;
    if (s->build_distinct != ~0ull) // asked before for these hashes (size query, then fetch): the sorted set is still there
    {
        *n_distinct = s->build_distinct;
        if (!out || s->build_distinct == 0)
            return GN_OK;
        if (cap < s->build_distinct)
            return gn_fail(GN_EOVERFLOW, "hash buffer too small: need %llu", (unsigned long long)s->build_distinct);
        GN_HIP(hipMemcpy(out, s->d_build[0], s->build_distinct * 8, hipMemcpyDeviceToHost));
        return GN_OK;
    }
    // every read's count is at most its window count, so the slot total bounds the packed size
    uint64_t slots = 0;
    GN_HIP(hipMemcpyAsync(&slots, s->d_slot_off + n, 8, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    if (slots == 0)
    {
        s->build_distinct = 0;
        return GN_OK;
    }
    if (slots + slots / 8 + 1024 > 0x7FFFFFFFull) // (the sort / unique calls take their item count -- the reserved capacity -- as an int)
        return gn_fail(GN_ERANGE, "more than 1.9 * 10^9 minimiser windows in one batch");
    int rc = gn_build_reserve(s, slots);
    if (rc)
        return rc;
    GN_HIP(hipMemsetAsync(s->d_build_ctr, 0, 2 * sizeof(unsigned long long), s->st));
    hipLaunchKernelGGL(gn_pack_hashes_kernel, dim3((unsigned)(((uint64_t)n * 64 + 255) / 256)), dim3(256), 0, s->st, s->d_hashes,
                       s->d_slot_off, s->d_nh, n, s->d_build[0], s->d_build_ctr);
    GN_HIP(hipGetLastError());
    unsigned long long total = 0;
    GN_HIP(hipMemcpyAsync(&total, s->d_build_ctr, 8, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    if (total == 0)
    {
        s->build_distinct = 0;
        return GN_OK;
    }
    size_t    tmp      = s->build_tmp_bytes;
    const int end_bit  = (int)(2 * s->k > 64 ? 64 : 2 * s->k); // values are below 4^k
    GN_HIP(hipcub::DeviceRadixSort::SortKeys(s->d_build_tmp, tmp, s->d_build[0], s->d_build[1], (int)total, 0, end_bit, s->st));
    tmp = s->build_tmp_bytes;
    GN_HIP(hipcub::DeviceSelect::Unique(s->d_build_tmp, tmp, s->d_build[1], s->d_build[0], s->d_build_ctr + 1, (int)total, s->st));
    unsigned long long nd = 0;
    GN_HIP(hipMemcpyAsync(&nd, s->d_build_ctr + 1, 8, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    *n_distinct       = nd;
    s->build_distinct = nd;
    if (!out)
        return GN_OK;
    if (cap < nd)
        return gn_fail(GN_EOVERFLOW, "hash buffer too small: need %llu", nd);
    GN_HIP(hipMemcpy(out, s->d_build[0], nd * 8, hipMemcpyDeviceToHost));
    return GN_OK;
}

__global__ void gn_emplace_split_kernel(uint64_t* rows, uint64_t S, uint32_t W, uint32_t shift, uint32_t h,
                                        const uint64_t* __restrict__ hashes, uint64_t n, uint32_t first_bin, uint64_t per_bin,
                                        uint64_t index_base)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * h)
        return;
    const uint64_t q   = idx / h;
    const uint32_t i   = (uint32_t)(idx - q * h);
    const uint32_t bin = first_bin + (uint32_t)((index_base + q) / per_bin);
    const uint32_t row = gn_build_row(hashes[q], i, shift, S);
    atomicOr(reinterpret_cast<unsigned long long*>(rows + ((uint64_t)row * W + (bin >> 6))), 1ULL << (bin & 63));
}

extern "C" int gn_filter_emplace_split(gn_filter* f, const uint64_t* hashes, uint64_t n, uint32_t first_bin, uint64_t hashes_per_bin)
{
    if (!f || f->is_hibf)
        return gn_fail(GN_EINVAL, "gn_filter_emplace_split needs a flat IBF filter");
    if (n == 0)
        return GN_OK;
    if (!hashes || hashes_per_bin == 0)
        return gn_fail(GN_EINVAL, "bad argument");
    GnIbfHost& ib = f->ibf;
    if ((uint64_t)first_bin + (n - 1) / hashes_per_bin >= ib.B)
        return gn_fail(GN_EINVAL, "bins %u.. exceed the filter's %llu bins", first_bin, (unsigned long long)ib.B);
    GN_HIP(hipSetDevice(f->device));
    if (!f->load_st)
        GN_HIP(hipStreamCreateWithFlags(&f->load_st, hipStreamNonBlocking));
    // staged through a device buffer that stays with the filter (grown to the largest request so far), at most 32 M hashes
    // at a time
    const uint64_t step = n < (32ull << 20) ? n : (32ull << 20);
    if (f->emplace_stage_cap < step)
    {
        if (f->d_emplace_stage)
            hipFree(f->d_emplace_stage);
        f->d_emplace_stage   = nullptr;
        f->emplace_stage_cap = 0;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&f->d_emplace_stage), step * 8));
        f->emplace_stage_cap = step;
    }
    for (uint64_t done = 0; done < n; done += step)
    {
        const uint64_t c = n - done < step ? n - done : step;
        GN_HIP(hipMemcpyAsync(f->d_emplace_stage, hashes + done, c * 8, hipMemcpyHostToDevice, f->load_st));
        const uint64_t total  = c * ib.h;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(gn_emplace_split_kernel, dim3(blocks), dim3(256), 0, f->load_st, ib.d_rows, ib.S, (uint32_t)ib.Ws, ib.shift,
                           ib.h, f->d_emplace_stage, c, first_bin, hashes_per_bin, done);
        GN_HIP(hipGetLastError());
        GN_HIP(hipStreamSynchronize(f->load_st)); // (the staging buffer is reused, and `hashes` may be pageable)
    }
    return GN_OK;
}

// ---- gn_filter_probe: the check of the reference's build test (tests/ganon-build/GanonBuild.test.cpp:53-98) on the device ------
// For every hash: in how many of the given technical bins are all h of its bits set?  hits = the sum over hashes (what the
// reference compares with hashes.size() after bulk_count: the target's bins' counts added up), missing = hashes that are in
// none of the bins (a false negative: the filter was not built from this sequence, or it is read with the wrong row / bin
// arithmetic), first = the smallest index of such a hash.  One lane per hash; h row words per (hash, bin) -- a few MB of
// gathered words per genome, nothing to tune.
__global__ void gn_probe_kernel(const uint64_t* __restrict__ rows, uint64_t S, uint32_t W, uint32_t shift, uint32_t h,
                                const uint64_t* __restrict__ hashes, uint64_t n, const uint32_t* __restrict__ bins, uint32_t n_bins,
                                uint64_t index_base, unsigned long long* __restrict__ out /* hits, missing, first */)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n)
        return;
    uint32_t row[5];
    for (uint32_t i = 0; i < h; ++i)
        row[i] = gn_build_row(hashes[q], i, shift, S);
    uint32_t found = 0;
    for (uint32_t b = 0; b < n_bins; ++b)
    {
        const uint32_t bin = bins[b];
        uint64_t       all = 1;
        for (uint32_t i = 0; i < h; ++i)
            all &= rows[(uint64_t)row[i] * W + (bin >> 6)] >> (bin & 63);
        found += (uint32_t)(all & 1);
    }
    if (found)
        atomicAdd(out, (unsigned long long)found);
    else
    {
        atomicAdd(out + 1, 1ull);
        atomicMin(out + 2, (unsigned long long)(index_base + q));
    }
}

extern "C" int gn_filter_probe(gn_filter* f, const uint64_t* hashes, uint64_t n, const uint32_t* bins, uint32_t n_bins, uint64_t* hits,
                               uint64_t* missing, uint64_t* first_missing)
{
    if (!f || f->is_hibf)
        return gn_fail(GN_EINVAL, "gn_filter_probe needs a flat IBF filter");
    if (!hits || !missing || !first_missing || (n && !hashes) || (n_bins && !bins))
        return gn_fail(GN_EINVAL, "gn_filter_probe: null argument");
    *hits = *missing = 0;
    *first_missing = ~0ull;
    if (n == 0)
        return GN_OK;
    GnIbfHost& ib = f->ibf;
    for (uint32_t b = 0; b < n_bins; ++b)
        if (bins[b] >= ib.B)
            return gn_fail(GN_EINVAL, "gn_filter_probe: bin %u of a filter with %llu bins", bins[b], (unsigned long long)ib.B);
    GN_HIP(hipSetDevice(f->device));
    if (!f->load_st)
        GN_HIP(hipStreamCreateWithFlags(&f->load_st, hipStreamNonBlocking));
    const uint64_t step = n < (32ull << 20) ? n : (32ull << 20);
    if (f->emplace_stage_cap < step)
    {
        if (f->d_emplace_stage)
            hipFree(f->d_emplace_stage);
        f->d_emplace_stage   = nullptr;
        f->emplace_stage_cap = 0;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&f->d_emplace_stage), step * 8));
        f->emplace_stage_cap = step;
    }
    uint32_t*           d_bins = nullptr;
    unsigned long long* d_out  = nullptr;
    unsigned long long  init[3] = { 0, 0, ~0ull };
    hipError_t          e = hipMalloc(reinterpret_cast<void**>(&d_bins), (size_t)(n_bins ? n_bins : 1) * 4);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(init));
    if (e == hipSuccess && n_bins)
        e = hipMemcpyAsync(d_bins, bins, (size_t)n_bins * 4, hipMemcpyHostToDevice, f->load_st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, f->load_st);
    for (uint64_t done = 0; done < n && e == hipSuccess; done += step)
    {
        const uint64_t c = n - done < step ? n - done : step;
        e = hipMemcpyAsync(f->d_emplace_stage, hashes + done, c * 8, hipMemcpyHostToDevice, f->load_st);
        if (e != hipSuccess)
            break;
        hipLaunchKernelGGL(gn_probe_kernel, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, f->load_st, ib.d_rows, ib.S, (uint32_t)ib.Ws, ib.shift,
                           ib.h, f->d_emplace_stage, c, d_bins, n_bins, done, d_out);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipStreamSynchronize(f->load_st); // (the staging buffer is reused, and `hashes` may be pageable)
    }
    if (e == hipSuccess)
        e = hipMemcpy(init, d_out, sizeof(init), hipMemcpyDeviceToHost);
    if (d_bins)
        (void)hipFree(d_bins);
    if (d_out)
        (void)hipFree(d_out);
    if (e != hipSuccess)
        return gn_fail(GN_ENODEV, "gn_filter_probe: %s", hipGetErrorString(e));
    *hits          = init[0];
    *missing       = init[1];
    *first_missing = init[2];
    return GN_OK;
}
