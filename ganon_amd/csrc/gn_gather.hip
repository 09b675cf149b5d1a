// gn_gather.hip -- the exchange step of a bin-range partitioned flat IBF (BASELINE config 5).
//
// A filter too large for one GPU is cut by technical-bin range, at target boundaries, into column parts that live on
// different devices (host/backend_hip.cpp places them; every part is a flat IBF of its own, gn_filter_write_rows with
// word_lo).  Every device classifies every read of a batch against its parts and applies the cutoff locally -- a target's
// bins never straddle a cut, so the per-target sum of GanonClassify.cpp:516-527 is complete inside one part.  What is left
// of the reference's single select_matches call is to put a read's sparse matches back together: part after part, which
// is ascending target order, because targets ascend with the bins.
//
//   gn_gather_run:  every part's grouped matches + per-read offsets travel to the batch's OWNER device
//                   (hipMemcpyPeerAsync: device to device over xGMI, nothing through the host), and one kernel there
//                   concatenates them per read in part order and rewrites part-local target ids into the caller's.
//
// The owner then holds exactly the result an unpartitioned filter would have produced on that device, grouped by read;
// gn_gather_fetch hands it to the host in one copy.  If a filter_matches pre-pass is set (gn_streams_postfilter_joint,
// which runs over the parts' streams on all their devices first), the parts hold survivors only and only those travel.
// The same call serves parts that share a device (filters wider than one row group of the count kernels): then nothing is
// copied, the kernel reads the streams' buffers in place.
#include "gn_internal.h"

#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#define GN_GATHER_MAX_PARTS 64
#define GN_GATHER_LIGHT 6u

int gn_finish_batch(gn_stream* s); // gn_capi.hip

struct gn_gather
{
    int         device = 0;
    hipStream_t st     = nullptr;
    uint32_t    n_parts = 0;
    std::vector<uint32_t*> d_map; // per part: part-local target id -> caller's id (nullptr = the same)
    std::vector<uint32_t>  n_map;
    // landing buffers on `device` for parts that live elsewhere
    std::vector<uint64_t*> d_off;
    std::vector<uint64_t>  off_cap;
    std::vector<gn_match*> d_in;
    std::vector<uint64_t>  in_cap;
    // result
    uint64_t* d_moff   = nullptr;
    uint64_t  moff_cap = 0;
    gn_match* d_out    = nullptr;
    uint64_t  out_cap  = 0;
    uint32_t  n_reads  = 0;
    uint64_t  n_matches = 0;
    uint64_t  peer_bytes = 0; // bytes that crossed devices in the last run
    bool      ran = false;
};

struct GnGatherParams
{
    const uint64_t* off[GN_GATHER_MAX_PARTS]; // n_reads+1 offsets into in[i]
    const gn_match* in[GN_GATHER_MAX_PARTS];
    const uint32_t* map[GN_GATHER_MAX_PARTS];
    uint32_t        k;
    uint32_t        n_reads;
    uint64_t*       moff; // n_reads+1
    gn_match*       out;
};

// read r of the result = its segment of part 0, then of part 1, ...; moff[r] = sum of the parts' offsets (a sum of
// exclusive prefix sums is the exclusive prefix sum of the sums: no scan is needed).  Part i's read r is
// in[i][off[i][r] - off[i][0] .. off[i][r+1] - off[i][0]).
__global__ __launch_bounds__(256) void gn_gather_parts_kernel(GnGatherParams p)
{
    const uint64_t r     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane  = threadIdx.x & 63u;
    const bool     valid = r < p.n_reads;
    uint64_t       o = 0;
    uint32_t       c = 0;
    if (r <= p.n_reads)
    {
        for (uint32_t i = 0; i < p.k; ++i)
            o += p.off[i][r] - p.off[i][0]; // (a part's offsets may start anywhere: a slice of a longer offset array)
        p.moff[r] = o;
    }
    if (valid)
        for (uint32_t i = 0; i < p.k; ++i)
            c += (uint32_t)(p.off[i][r + 1] - p.off[i][r]);
    if (valid && c != 0 && c <= GN_GATHER_LIGHT)
    {
        uint64_t w = o;
        for (uint32_t i = 0; i < p.k; ++i)
        {
            const uint64_t base = p.off[i][0], b = p.off[i][r] - base, e = p.off[i][r + 1] - base;
            for (uint64_t j = b; j < e; ++j)
            {
                gn_match m = p.in[i][j];
                if (p.map[i])
                    m.target = p.map[i][m.target];
                p.out[w++] = m;
            }
        }
    }
    uint64_t heavy = __ballot(valid && c > GN_GATHER_LIGHT);
    while (heavy)
    {
        const uint32_t L = (uint32_t)__builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const uint64_t rr = r - lane + L;
        uint64_t       w  = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), (int)L) << 32) |
                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, (int)L);
        for (uint32_t i = 0; i < p.k; ++i)
        {
            const uint64_t  b   = p.off[i][rr] - p.off[i][0];
            const uint32_t  cs  = (uint32_t)(p.off[i][rr + 1] - p.off[i][rr]);
            const uint32_t* map = p.map[i];
            for (uint32_t j = lane; j < cs; j += 64)
            {
                gn_match m = p.in[i][b + j];
                if (map)
                    m.target = map[m.target];
                p.out[w + j] = m;
            }
            w += cs;
        }
    }
}

__global__ void gn_gather_pick_kernel(const uint64_t* seg_off, uint32_t wpr, uint32_t n_reads, uint64_t* match_off)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads)
        match_off[r] = seg_off[(size_t)r * wpr];
}

// direct device-to-device copies need peer access switched on once per ordered pair; without it hipMemcpyPeerAsync still
// works (the runtime stages the copy), so a refusal is not an error
// What happened per ordered device pair, for gn_peer_stats: [dst * n + src] = 0 never asked, 1 peer access on (direct copies
// over the link between the two), 2 not available (the runtime stages the copy through the host); bytes gn_gather moved.
static std::mutex            g_peer_mu;
static std::vector<uint8_t>  g_peer_state;
static std::vector<uint64_t> g_peer_bytes;
static int                   g_peer_n = 0;

static bool gn_peer_table(int need_a, int need_b)
{
    if (g_peer_n == 0)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            return false;
        g_peer_n = n;
        g_peer_state.assign((size_t)n * n, 0);
        g_peer_bytes.assign((size_t)n * n, 0);
    }
    return need_a >= 0 && need_b >= 0 && need_a < g_peer_n && need_b < g_peer_n;
}

void gn_peer_enable(int dst, int src)
{
    if (dst == src)
        return;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    if (!gn_peer_table(dst, src))
        return;
    const int n = g_peer_n;
    for (int a : { dst, src })
    {
        const int b = a == dst ? src : dst;
        if (g_peer_state[(size_t)a * n + b])
            continue;
        g_peer_state[(size_t)a * n + b] = 2;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can)
        {
            int cur = 0;
            hipGetDevice(&cur);
            hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled)
                g_peer_state[(size_t)a * n + b] = 1;
            (void)hipGetLastError();
            hipSetDevice(cur);
        }
    }
}

static void gn_peer_count(int dst, int src, uint64_t bytes)
{
    std::lock_guard<std::mutex> lk(g_peer_mu);
    if (gn_peer_table(dst, src))
        g_peer_bytes[(size_t)dst * g_peer_n + src] += bytes;
}

extern "C" int gn_peer_stats(int dst, int src, int* state, uint64_t* bytes)
{
    std::lock_guard<std::mutex> lk(g_peer_mu);
    if (!gn_peer_table(dst, src))
        return gn_fail(GN_EINVAL, "gn_peer_stats: no device pair (%d, %d)", dst, src);
    if (state)
        *state = g_peer_state[(size_t)dst * g_peer_n + src];
    if (bytes)
        *bytes = g_peer_bytes[(size_t)dst * g_peer_n + src];
    return GN_OK;
}

extern "C" int gn_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return gn_fail(GN_ENODEV, "no HIP device available (libganon_hip has no CPU fallback)");
    if (device < 0 || device >= c)
        return gn_fail(GN_EINVAL, "device %d out of range (%d devices)", device, c);
    GN_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    GN_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes)
        *free_bytes = f;
    if (total_bytes)
        *total_bytes = t;
    return GN_OK;
}

extern "C" int gn_gather_destroy(gn_gather* g)
{
    if (!g)
        return GN_OK;
    hipSetDevice(g->device);
    if (g->st)
        hipStreamSynchronize(g->st);
    for (auto* q : g->d_map)
        if (q)
            hipFree(q);
    for (auto* q : g->d_off)
        if (q)
            hipFree(q);
    for (auto* q : g->d_in)
        if (q)
            hipFree(q);
    if (g->d_moff)
        hipFree(g->d_moff);
    if (g->d_out)
        hipFree(g->d_out);
    if (g->st)
        hipStreamDestroy(g->st);
    delete g;
    return GN_OK;
}

extern "C" int gn_gather_create(int device, uint32_t n_parts, const uint32_t* const* target_map, const uint32_t* n_map, gn_gather** out)
{
    if (!out || n_parts == 0 || n_parts > GN_GATHER_MAX_PARTS)
        return gn_fail(GN_EINVAL, "gn_gather_create: 1..%d parts", GN_GATHER_MAX_PARTS);
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return gn_fail(GN_ENODEV, "no HIP device available (libganon_hip has no CPU fallback)");
    if (device < 0 || device >= c)
        return gn_fail(GN_EINVAL, "device %d out of range (%d devices)", device, c);
    GN_HIP(hipSetDevice(device));
    gn_gather* g = new (std::nothrow) gn_gather();
    if (!g)
        return gn_fail(GN_ENOMEM, "out of host memory");
    g->device  = device;
    g->n_parts = n_parts;
    g->d_map.assign(n_parts, nullptr);
    g->n_map.assign(n_parts, 0);
    g->d_off.assign(n_parts, nullptr);
    g->off_cap.assign(n_parts, 0);
    g->d_in.assign(n_parts, nullptr);
    g->in_cap.assign(n_parts, 0);
    hipError_t e = hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking);
    for (uint32_t i = 0; i < n_parts && e == hipSuccess; ++i)
    {
        if (!target_map || !target_map[i] || !n_map || n_map[i] == 0)
            continue;
        g->n_map[i] = n_map[i];
        e = hipMalloc(reinterpret_cast<void**>(&g->d_map[i]), (size_t)n_map[i] * 4);
        if (e == hipSuccess)
            e = hipMemcpy(g->d_map[i], target_map[i], (size_t)n_map[i] * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess)
    {
        gn_gather_destroy(g);
        return gn_fail(e == hipErrorOutOfMemory ? GN_ENOMEM : GN_ENODEV, "gn_gather_create: %s", hipGetErrorString(e));
    }
    *out = g;
    return GN_OK;
}

template <typename T>
static int gn_gather_reserve(T** p, uint64_t* cap, uint64_t need)
{
    if (*cap >= need && *p)
        return GN_OK;
    if (*p)
        GN_HIP(hipFree(*p));
    *p   = nullptr;
    *cap = 0;
    const uint64_t n = need + need / 4 + 1024;
    GN_HIP(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
    *cap = n;
    return GN_OK;
}

extern "C" int gn_gather_run(gn_gather* g, gn_stream* const* streams, uint32_t n_streams)
{
    if (!g || !streams || n_streams != g->n_parts)
        return gn_fail(GN_EINVAL, "gn_gather_run: the gather was created for %u parts", g ? g->n_parts : 0u);
    g->ran = false;
    const bool force_copy = gn_sw().gather_copy; // tests: same-device parts take the peer-copy path too
    // every part's batch is complete (a match buffer that overflowed is grown and the part re-run first)
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        gn_stream* s = streams[i];
        if (!s || s->f->is_hibf)
            return gn_fail(GN_EINVAL, "gn_gather_run: part %u is not a stream on a flat IBF", i);
        if (s->pf_on && s->pf_joint && !s->pf_joint_done)
            return gn_fail(GN_EINVAL, "gn_gather_run: part %u waits for gn_streams_postfilter_joint", i);
        int rc = gn_finish_batch(s);
        if (rc)
            return rc;
        if (s->n_reads != streams[0]->n_reads)
            return gn_fail(GN_EINVAL, "gn_gather_run: the parts hold different batches");
        total += s->n_matches;
    }
    const uint32_t n = streams[0]->n_reads;
    GnGatherParams p{};
    p.k       = n_streams;
    p.n_reads = n;
    g->peer_bytes = 0;
    // per part: its per-read offsets (n+1) and its grouped matches, where they are or copied over
    for (uint32_t i = 0; i < n_streams; ++i)
    {
        gn_stream*      s   = streams[i];
        const uint32_t  wpr = s->f->geom.wpr;
        const uint64_t* src_off;
        GN_HIP(hipSetDevice(s->device));
        if (s->pf_on)
            src_off = s->d_slot_cnt; // survivors' offsets (gn_pf_finish)
        else if (wpr == 1)
            src_off = s->d_seg_off;
        else
        {
            // (d_slot_cnt is free after the minimiser kernels)
            hipLaunchKernelGGL(gn_gather_pick_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, s->st, s->d_seg_off, wpr, n, s->d_slot_cnt);
            GN_HIP(hipGetLastError());
            GN_HIP(hipStreamSynchronize(s->st));
            src_off = s->d_slot_cnt;
        }
        int rc_c = gn_result_compact(s); // (a segmented result that no pre-pass has compacted)
        if (rc_c)
            return rc_c;
        if (s->cmp_timed)
            GN_HIP(hipStreamSynchronize(s->st)); // (the copy was queued on the part's stream just now; the merge runs on the gather's)
        const gn_match* src_m = gn_result_matches(s);
        if (s->device == g->device && !force_copy)
        {
            p.off[i] = src_off;
            p.in[i]  = src_m;
        }
        else
        {
            GN_HIP(hipSetDevice(g->device));
            int rc = gn_gather_reserve(&g->d_off[i], &g->off_cap[i], (uint64_t)n + 1);
            if (rc)
                return rc;
            rc = gn_gather_reserve(&g->d_in[i], &g->in_cap[i], s->n_matches);
            if (rc)
                return rc;
            gn_peer_enable(g->device, s->device);
            GN_HIP(hipMemcpyPeerAsync(g->d_off[i], g->device, src_off, s->device, ((size_t)n + 1) * 8, g->st));
            if (s->n_matches)
                GN_HIP(hipMemcpyPeerAsync(g->d_in[i], g->device, src_m, s->device, s->n_matches * sizeof(gn_match), g->st));
            g->peer_bytes += ((uint64_t)n + 1) * 8 + s->n_matches * sizeof(gn_match);
            gn_peer_count(g->device, s->device, ((uint64_t)n + 1) * 8 + s->n_matches * sizeof(gn_match));
            p.off[i] = g->d_off[i];
            p.in[i]  = g->d_in[i];
        }
        p.map[i] = g->d_map[i];
    }
    GN_HIP(hipSetDevice(g->device));
    int rc = gn_gather_reserve(&g->d_moff, &g->moff_cap, (uint64_t)n + 1);
    if (rc)
        return rc;
    rc = gn_gather_reserve(&g->d_out, &g->out_cap, total);
    if (rc)
        return rc;
    p.moff = g->d_moff;
    p.out  = g->d_out;
    hipLaunchKernelGGL(gn_gather_parts_kernel, dim3((unsigned)(((uint64_t)n + 1 + 255) / 256)), dim3(256), 0, g->st, p);
    GN_HIP(hipGetLastError());
    g->n_reads   = n;
    g->n_matches = total;
    g->ran       = true;
    return GN_OK;
}

// The same concatenation over parts that are plain device buffers on the gather's device -- the receive buffers of the
// multi-process exchange (ganon_amd/partition.py: one process per GPU, the parts' records arrive through RCCL's
// all-to-all): d_off[i] = n_reads+1 offsets of part i (any origin: read r = d_matches[i][d_off[i][r]-d_off[i][0] ...)).
// The caller makes sure the buffers are complete (its own stream is synchronised) before the call.
extern "C" int gn_gather_run_buffers(gn_gather* g, const uint64_t* const* d_off, const gn_match* const* d_matches, const uint64_t* n_matches,
                                     uint32_t n_parts, uint32_t n_reads)
{
    if (!g || !d_off || !d_matches || !n_matches || n_parts != g->n_parts)
        return gn_fail(GN_EINVAL, "gn_gather_run_buffers: the gather was created for %u parts", g ? g->n_parts : 0u);
    g->ran = false;
    GN_HIP(hipSetDevice(g->device));
    GnGatherParams p{};
    p.k       = n_parts;
    p.n_reads = n_reads;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_parts; ++i)
    {
        if (!d_off[i] || (!d_matches[i] && n_matches[i]))
            return gn_fail(GN_EINVAL, "gn_gather_run_buffers: part %u has no buffers", i);
        p.off[i] = d_off[i];
        p.in[i]  = d_matches[i];
        p.map[i] = g->d_map[i];
        total += n_matches[i];
    }
    int rc = gn_gather_reserve(&g->d_moff, &g->moff_cap, (uint64_t)n_reads + 1);
    if (rc)
        return rc;
    rc = gn_gather_reserve(&g->d_out, &g->out_cap, total);
    if (rc)
        return rc;
    p.moff = g->d_moff;
    p.out  = g->d_out;
    hipLaunchKernelGGL(gn_gather_parts_kernel, dim3((unsigned)(((uint64_t)n_reads + 1 + 255) / 256)), dim3(256), 0, g->st, p);
    GN_HIP(hipGetLastError());
    g->n_reads    = n_reads;
    g->n_matches  = total;
    g->peer_bytes = 0;
    g->ran        = true;
    return GN_OK;
}

// per-read offsets (n_reads+1) of the stream's grouped matches, in device memory: what gn_stream_device_matches leaves out
extern "C" int gn_stream_device_offsets(gn_stream* s, const uint64_t** d_match_off)
{
    if (!s || !d_match_off)
        return gn_fail(GN_EINVAL, "null argument");
    int rc = gn_finish_batch(s);
    if (rc)
        return rc;
    GN_HIP(hipSetDevice(s->device));
    const uint32_t wpr = s->f->is_hibf ? 1u : (uint32_t)s->f->geom.wpr, n = s->n_reads;
    if (s->pf_on && (!s->pf_joint || s->pf_joint_done))
        *d_match_off = s->d_slot_cnt;
    else if (wpr == 1)
        *d_match_off = s->d_seg_off;
    else
    {
        hipLaunchKernelGGL(gn_gather_pick_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, s->st, s->d_seg_off, wpr, n, s->d_slot_cnt);
        GN_HIP(hipGetLastError());
        GN_HIP(hipStreamSynchronize(s->st));
        *d_match_off = s->d_slot_cnt;
    }
    return GN_OK;
}

extern "C" int gn_gather_fetch(gn_gather* g, uint64_t* match_off, gn_match* matches, uint64_t cap, uint64_t* n_matches)
{
    if (!g)
        return gn_fail(GN_EINVAL, "null gather");
    if (!g->ran)
        return gn_fail(GN_EINVAL, "gn_gather_fetch: nothing gathered yet");
    GN_HIP(hipSetDevice(g->device));
    if (n_matches)
        *n_matches = g->n_matches;
    if (match_off)
        GN_HIP(hipMemcpyAsync(match_off, g->d_moff, ((size_t)g->n_reads + 1) * 8, hipMemcpyDeviceToHost, g->st));
    if (matches)
    {
        if (cap < g->n_matches)
        {
            hipStreamSynchronize(g->st);
            return gn_fail(GN_EOVERFLOW, "match buffer too small: need %llu, have %llu", (unsigned long long)g->n_matches,
                           (unsigned long long)cap);
        }
        if (g->n_matches)
            GN_HIP(hipMemcpyAsync(matches, g->d_out, g->n_matches * sizeof(gn_match), hipMemcpyDeviceToHost, g->st));
    }
    GN_HIP(hipStreamSynchronize(g->st));
    return GN_OK;
}

extern "C" int gn_gather_device_matches(gn_gather* g, const gn_match** d_matches, const uint64_t** d_match_off, uint64_t* n_matches,
                                        uint64_t* peer_bytes)
{
    if (!g || !g->ran)
        return gn_fail(GN_EINVAL, "gn_gather_device_matches: nothing gathered yet");
    GN_HIP(hipSetDevice(g->device));
    GN_HIP(hipStreamSynchronize(g->st));
    if (d_matches)
        *d_matches = g->d_out;
    if (d_match_off)
        *d_match_off = g->d_moff;
    if (n_matches)
        *n_matches = g->n_matches;
    if (peer_bytes)
        *peer_bytes = g->peer_bytes;
    return GN_OK;
}
