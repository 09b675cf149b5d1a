// gn_capi.hip -- C ABI of libganon_hip.so (include/ganon_hip.h): filter upload, batch submit, sparse-match fetch.
// Host-side HIP runtime code only; the kernels live in gn_kernels.hip / gn_hibf.hip.
#include "gn_internal.h"

#include <hipcub/hipcub.hpp>
#include "gn_scan.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include <sys/mman.h>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int gn_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* gn_last_error(void)
{
    return g_err;
}

// ---- switches: $GANON_HIP_ABLATE, parsed once at load; gn_ablate() for in-process cross-checks (gn_internal.h) ----------------
static GnSwitches g_sw;

static int gn_parse_switches(const char* list, GnSwitches* out, char* bad, size_t bad_len)
{
    GnSwitches sw;
    struct
    {
        const char* name;
        bool GnSwitches::* flag;
    } static const flags[] = {{"early_exit", &GnSwitches::early_exit},       {"cand_select", &GnSwitches::cand_select},
                              {"csr_identity", &GnSwitches::csr_identity},   {"uniform_select", &GnSwitches::uniform_select},
                              {"run_select", &GnSwitches::run_select},       {"max_first", &GnSwitches::max_first},
                              {"const_nb", &GnSwitches::const_nb},           {"split_kernel", &GnSwitches::split_kernel},
                              {"predrop", &GnSwitches::predrop},             {"deferred_grids", &GnSwitches::deferred_grids}, {"on_demand", &GnSwitches::on_demand}, {"hibf_dense_rows", &GnSwitches::hibf_dense_rows},
                              {"hibf_reg", &GnSwitches::hibf_reg},           {"hibf_pack", &GnSwitches::hibf_pack},
                              {"hibf_one_pack", &GnSwitches::hibf_one_pack}, {"hibf_persistent", &GnSwitches::hibf_persistent}, {"hibf_stage", &GnSwitches::hibf_stage}, {"hibf_nsort", &GnSwitches::hibf_nsort}, {"hibf_reread", &GnSwitches::hibf_reread},
                              {"hibf_fake_hashes", &GnSwitches::hibf_fake_hashes}, {"gather_copy", &GnSwitches::gather_copy},     {"joint_apart", &GnSwitches::joint_apart},
                              {"pinned_malloc", &GnSwitches::pinned_malloc}, {"inflate_ahead", &GnSwitches::inflate_ahead}, {"debug", &GnSwitches::debug},
                              {"fake_count", &GnSwitches::fake_count}, {"seg_result", &GnSwitches::seg_result}};
    for (const char* p = list ? list : ""; *p;)
    {
        const char* e = strchr(p, ',');
        size_t      n = e ? (size_t)(e - p) : strlen(p);
        while (n && (*p == ' ' || *p == '\t'))
            ++p, --n;
        while (n && (p[n - 1] == ' ' || p[n - 1] == '\t'))
            --n;
        bool known = n == 0;
        for (const auto& f : flags)
            if (!known && strlen(f.name) == n && !strncmp(p, f.name, n))
                sw.*(f.flag) = true, known = true;
        if (!known && n > 6 && !strncmp(p, "chunk=", 6))
            sw.chunk = (uint32_t)strtoul(p + 6, nullptr, 10), known = true;
        if (!known && n > 11 && !strncmp(p, "emit_probe=", 11))
            sw.emit_probe = (uint32_t)strtoul(p + 11, nullptr, 10) & (64u | 128u), known = true;
        if (!known && n > 9 && !strncmp(p, "hibf_bpc=", 9))
            sw.hibf_bpc = (uint32_t)strtoul(p + 9, nullptr, 10), known = true;
        if (!known && n > 16 && !strncmp(p, "hibf_pair_limit=", 16))
            sw.hibf_pair_limit = std::max<uint64_t>(64, strtoull(p + 16, nullptr, 10)), known = true;
        if (!known && n > 5 && !strncmp(p, "sync=", 5))
        {
            const char* m = p + 5;
            const size_t l = n - 5;
            sw.sync_mode = (l == 4 && !strncmp(m, "spin", 4)) ? 1 : (l == 5 && !strncmp(m, "yield", 5)) ? 2 : (l == 5 && !strncmp(m, "block", 5)) ? 3 : -1;
            known = sw.sync_mode > 0;
        }
        if (!known)
        {
            snprintf(bad, bad_len, "%.*s", (int)std::min<size_t>(n, 60), p);
            return GN_EINVAL;
        }
        p = e ? e + 1 : p + strlen(p);
    }
    *out = sw;
    return GN_OK;
}

namespace
{
struct GnSwitchesInit
{
    GnSwitchesInit()
    {
        char bad[64];
        if (gn_parse_switches(getenv("GANON_HIP_ABLATE"), &g_sw, bad, sizeof(bad)) != GN_OK)
            fprintf(stderr, "libganon_hip: $GANON_HIP_ABLATE names an unknown switch '%s' -- the whole list is ignored\n", bad);
    }
} g_sw_init;
} // namespace

const GnSwitches& gn_sw()
{
    return g_sw;
}

extern "C" int gn_ablate(const char* list)
{
    char bad[64];
    if (gn_parse_switches(list, &g_sw, bad, sizeof(bad)) != GN_OK)
        return gn_fail(GN_EINVAL, "gn_ablate: unknown switch '%s'", bad);
    return GN_OK;
}

// How host threads wait for the device.  HIP's default spins: a worker thread that waits for its batch burns a core, and the
// host pipeline (parsers, post pool) has a use for that core (switch sync=spin|yield|block; read before anything touches a device).
static void gn_apply_sync_mode(int n_devices)
{
    static bool done = false;
    if (done)
        return;
    done = true;
    if (gn_sw().sync_mode <= 0)
        return;
    const unsigned flag = gn_sw().sync_mode == 3 ? hipDeviceScheduleBlockingSync : gn_sw().sync_mode == 2 ? hipDeviceScheduleYield : hipDeviceScheduleSpin;
    for (int d = 0; d < n_devices; ++d)
        if (hipSetDevice(d) == hipSuccess)
            (void)hipSetDeviceFlags(flag);
    (void)hipGetLastError();
}

extern "C" int gn_ibf_hash_constants(uint64_t seeds[5], uint64_t* multiplier)
{
    if (!seeds || !multiplier)
        return gn_fail(GN_EINVAL, "gn_ibf_hash_constants: null argument");
    static const uint64_t list[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST;
    for (int i = 0; i < GN_IBF_MAX_HASH_FUNS; ++i)
        seeds[i] = list[i];
    *multiplier = GN_IBF_MULTIPLIER;
    return GN_OK;
}

extern "C" int gn_device_count(int* n)
{
    if (!n)
        return gn_fail(GN_EINVAL, "gn_device_count: null argument");
    int        c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e == hipSuccess)
        gn_apply_sync_mode(c);
    if (e != hipSuccess)
    {
        *n = 0;
        return gn_fail(GN_ENODEV, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *n = c;
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------
static int gn_set_device(int dev)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return gn_fail(GN_ENODEV, "no HIP device available (libganon_hip has no CPU fallback)");
    if (dev < 0 || dev >= c)
        return gn_fail(GN_EINVAL, "device %d out of range (%d devices)", dev, c);
    GN_HIP(hipSetDevice(dev));
    return GN_OK;
}

// Padding bins (>= bins) of the last word are never set by emplace and never reported by the reference
// (counting_vector has `bins` entries); clear them in the device copy so kernels need no per-word mask.
__global__ void gn_clear_padding_kernel(uint64_t* rows, uint64_t S, uint64_t W, uint64_t Ws, uint64_t mask)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < S)
        rows[r * Ws + (W - 1)] &= mask;
}

extern "C" uint64_t gn_hibf_row_stride_words(uint64_t bin_words)
{
    return gn_sw().hibf_dense_rows ? bin_words : gn_pad_row_words(bin_words);
}

static int gn_upload_ibf_rows(const gn_ibf_desc* d, GnIbfHost* out, uint64_t* bytes_acc, bool pad_rows = false)
{
    if (!d || d->bin_size == 0 || d->bins == 0)
        return gn_fail(GN_EINVAL, "empty IBF description");
    if (d->bin_words != ((d->bins + 63) >> 6))
        return gn_fail(GN_EINVAL, "bin_words (%llu) != ceil(bins/64) (%llu)", (unsigned long long)d->bin_words,
                       (unsigned long long)((d->bins + 63) >> 6));
    if (d->hash_funs < 1 || d->hash_funs > 5)
        return gn_fail(GN_EINVAL, "hash_funs %u outside 1..5", d->hash_funs);
    if (d->hash_shift != (uint32_t)__builtin_clzll(d->bin_size))
        return gn_fail(GN_EINVAL, "hash_shift %u != countl_zero(bin_size) %d", d->hash_shift,
                       __builtin_clzll(d->bin_size));
    if (d->bin_size > 0xFFFFFFFFull)
        return gn_fail(GN_ERANGE, "bin_size > 2^32 rows is not supported");
    if (d->bin_words > 0xFFFFFFFFull || d->bins > 0xFFFFFFF0ull)
        return gn_fail(GN_ERANGE, "too many bins");
    const uint64_t Ws    = pad_rows ? gn_pad_row_words(d->bin_words) : d->bin_words;
    const uint64_t bytes = d->bin_size * Ws * 8ull;
    uint64_t*      dp    = nullptr;
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&dp), bytes + 64)); // +64: tail pad for 16-byte loads
    if (d->rows && Ws != d->bin_words)
    {
        // padded rows: zero everything, then the source's words row by row (in rounds of <= 1 GiB of source)
        GN_HIP(hipMemsetAsync(dp, 0, bytes + 64, nullptr));
        GN_HIP(hipDeviceSynchronize());
        const uint64_t per = std::max<uint64_t>(1, (1ull << 30) / (d->bin_words * 8));
        for (uint64_t r = 0; r < d->bin_size; r += per)
        {
            const uint64_t nr = std::min(per, d->bin_size - r);
            hipError_t     e  = hipMemcpy2D(dp + r * Ws, Ws * 8, d->rows + r * d->bin_words, d->bin_words * 8, d->bin_words * 8, nr, hipMemcpyHostToDevice);
            if (e != hipSuccess)
            {
                hipFree(dp);
                return gn_fail(GN_ENODEV, "filter upload failed: %s", hipGetErrorString(e));
            }
        }
    }
    else if (d->rows)
    {
        // chunked copy keeps pinned-staging pressure low for multi-GiB filters
        const uint64_t chunk = 1ull << 30;
        for (uint64_t o = 0; o < bytes; o += chunk)
        {
            const uint64_t nb = std::min(chunk, bytes - o);
            hipError_t     e  = hipMemcpy(reinterpret_cast<uint8_t*>(dp) + o, reinterpret_cast<const uint8_t*>(d->rows) + o,
                                          nb, hipMemcpyHostToDevice);
            if (e != hipSuccess)
            {
                hipFree(dp);
                return gn_fail(GN_ENODEV, "filter upload failed: %s", hipGetErrorString(e));
            }
        }
        GN_HIP(hipMemsetAsync(reinterpret_cast<uint8_t*>(dp) + bytes, 0, 64, nullptr));
    }
    else
    {
        GN_HIP(hipMemsetAsync(dp, 0, bytes + 64, nullptr));
    }
    if (d->rows && (d->bins & 63))
        hipLaunchKernelGGL(gn_clear_padding_kernel, dim3((unsigned)((d->bin_size + 255) / 256)), dim3(256), 0, nullptr, dp,
                           d->bin_size, d->bin_words, Ws, (1ull << (d->bins & 63)) - 1ull);
    // A memset of device memory returns before the fill has run, and everything that touches the filter afterwards --
    // gn_filter_write_rows' load stream, every gn_stream -- runs on streams created hipStreamNonBlocking, which take no
    // implicit order against the null stream the fill is queued on.  Without this wait the zero fill of a streamed filter
    // could land AFTER the first row chunks (round 3's intermittent "0 of 7 minimisers found": DESIGN 7-5b).
    GN_HIP(hipDeviceSynchronize());
    out->d_rows = dp;
    out->S      = d->bin_size;
    out->W      = d->bin_words;
    out->Ws     = Ws;
    out->B      = d->bins;
    out->h      = d->hash_funs;
    out->shift  = d->hash_shift;
    *bytes_acc += bytes;
    return GN_OK;
}

extern "C" int gn_filter_upload_ibf(int device, const gn_ibf_desc* ibf, const uint32_t* bin2target, uint32_t n_targets,
                                    gn_filter** out)
{
    if (!out || !ibf)
        return gn_fail(GN_EINVAL, "gn_filter_upload_ibf: null argument");
    *out = nullptr;
    int rc = gn_set_device(device);
    if (rc)
        return rc;
    if (!bin2target)
    {
        // storage-only filter (the builder's): bits can be written, inserted and read back, reads cannot be classified
        // against it -- so the shape limits of the count kernels do not apply (a database may have more bins than one
        // classify-side filter takes; ganon-classify cuts such a file into column parts when it loads it)
        gn_filter* f = new (std::nothrow) gn_filter();
        if (!f)
            return gn_fail(GN_ENOMEM, "out of host memory");
        f->device       = device;
        f->storage_only = true;
        rc              = gn_upload_ibf_rows(ibf, &f->ibf, &f->device_bytes);
        if (rc)
        {
            delete f;
            return rc;
        }
        *out = f;
        return GN_OK;
    }
    GnCountGeometry geom{};
    const char*     why = "";
    if (!gn_count_geometry(ibf->bin_words, ibf->hash_funs, &geom, &why))
        return gn_fail(GN_ERANGE, "unsupported IBF shape: %s", why);

    gn_filter* f = new (std::nothrow) gn_filter();
    if (!f)
        return gn_fail(GN_ENOMEM, "out of host memory");
    f->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        f->n_cu = prop.multiProcessorCount;
    f->geom = geom;
    rc      = gn_upload_ibf_rows(ibf, &f->ibf, &f->device_bytes);
    if (rc)
    {
        delete f;
        return rc;
    }
    // bin -> target map as CSR indexed by target id (replaces Filter::map, GanonClassify.cpp:1021-1025)
    bool identity = (n_targets == ibf->bins);
    for (uint64_t b = 0; b < ibf->bins; ++b)
    {
        const uint32_t t = bin2target[b];
        if (t != 0xFFFFFFFFu && t >= n_targets)
        {
            gn_filter_free(f);
            return gn_fail(GN_EINVAL, "bin2target[%llu] = %u >= n_targets %u", (unsigned long long)b, t, n_targets);
        }
        if (t != b)
            identity = false;
    }
    f->n_targets = n_targets;
    f->identity  = identity;
    if (!identity)
    {
        std::vector<uint32_t> off(n_targets + 1, 0), bins(ibf->bins ? ibf->bins : 1, 0);
        for (uint64_t b = 0; b < ibf->bins; ++b)
            if (bin2target[b] != 0xFFFFFFFFu)
                off[bin2target[b] + 1]++;
        for (uint32_t t = 0; t < n_targets; ++t)
            off[t + 1] += off[t];
        std::vector<uint32_t> fill(off.begin(), off.end() - 1);
        for (uint64_t b = 0; b < ibf->bins; ++b)
            if (bin2target[b] != 0xFFFFFFFFu)
                bins[fill[bin2target[b]]++] = (uint32_t)b;
        f->csr_identity = off[n_targets] == ibf->bins;
        for (size_t x = 0; x < bins.size() && f->csr_identity; ++x)
            f->csr_identity = bins[x] == (uint32_t)x;
        f->run_ok = f->csr_identity && n_targets < (1u << 28);
        for (uint32_t t = 0; t < n_targets && f->run_ok; ++t)
            f->run_ok = off[t + 1] > off[t];
        f->uniform_nb = 0;
        if (f->csr_identity && n_targets)
        {
            const uint32_t nb = off[1] - off[0];
            bool           same = nb == 2 || nb == 4;
            for (uint32_t t = 0; t < n_targets && same; ++t)
                same = off[t + 1] - off[t] == nb;
            f->uniform_nb = same ? nb : 0u;
        }
        std::vector<uint32_t> lds_idx(bins.size());
        for (size_t x = 0; x < bins.size(); ++x)
            lds_idx[x] = gn_count_lds_index(geom, bins[x]);
        hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&f->d_tgt_off), off.size() * 4);
        hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&f->d_tgt_bins), bins.size() * 4);
        hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&f->d_tgt_lds), bins.size() * 4);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
        {
            gn_filter_free(f);
            return gn_fail(GN_ENOMEM, "target map allocation failed");
        }
        hipMemcpy(f->d_tgt_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(f->d_tgt_bins, bins.data(), bins.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(f->d_tgt_lds, lds_idx.data(), lds_idx.size() * 4, hipMemcpyHostToDevice);
        std::vector<uint4> rec(n_targets ? n_targets : 1);
        for (uint32_t t = 0; t < n_targets; ++t)
        {
            const uint32_t len = off[t + 1] - off[t];
            rec[t] = make_uint4(off[t], len, len >= 1 ? lds_idx[off[t]] : 0u, len >= 2 ? lds_idx[off[t] + 1] : 0u);
        }
        if (hipMalloc(reinterpret_cast<void**>(&f->d_tgt_rec), rec.size() * sizeof(uint4)) != hipSuccess)
        {
            gn_filter_free(f);
            return gn_fail(GN_ENOMEM, "target map allocation failed");
        }
        hipMemcpy(f->d_tgt_rec, rec.data(), rec.size() * sizeof(uint4), hipMemcpyHostToDevice);
        // candidate-driven select of the generic kernel: bin -> target, and bins-per-target of every bin (one byte;
        // 0 for bins of no target and of targets with more than GN_CAND_NBIG bins, which go to big_list); the bytes
        // of count dwords 2j and 2j+1 (gn_count_lds_index: u16 pairs at dword q*(Gp+1)+gl) share dword
        // j*(Gp+1)+gl of a half-sized table
        {
            std::vector<uint32_t> bin_tgt(ibf->bins ? ibf->bins : 1, 0xFFFFFFFFu);
            std::vector<uint32_t> nb2((size_t)geom.wpr * (geom.slice_dwords / 2), 0u);
            std::vector<uint32_t> big;
            const uint32_t        gp1 = (1u << geom.gp_log2) + 1u;
            for (uint32_t t = 0; t < n_targets; ++t)
                if (off[t + 1] - off[t] > GN_CAND_NBIG)
                    big.push_back(t);
            for (uint64_t b = 0; b < ibf->bins; ++b)
            {
                const uint32_t t = bin2target[b];
                if (t == 0xFFFFFFFFu)
                    continue;
                bin_tgt[b]         = t;
                const uint32_t len = off[t + 1] - off[t];
                if (len > GN_CAND_NBIG)
                    continue;
                const uint32_t a    = gn_count_lds_index(geom, (uint32_t)b);
                const uint32_t dw   = a >> 1, half = a & 1u; // count dword inside the read's area, u16 half
                const uint32_t sl   = dw / geom.slice_dwords, in_sl = dw - sl * geom.slice_dwords;
                const uint32_t q    = in_sl / gp1, g = in_sl - q * gp1;
                nb2[(size_t)sl * (geom.slice_dwords / 2) + (q >> 1) * gp1 + g] |= len << (8 * (2 * (q & 1u) + half));
            }
            f->n_big = (uint32_t)big.size();
            if (big.empty())
                big.push_back(0);
            if (hipMalloc(reinterpret_cast<void**>(&f->d_bin_tgt), bin_tgt.size() * 4) != hipSuccess
                || hipMalloc(reinterpret_cast<void**>(&f->d_bin_nb2), nb2.size() * 4) != hipSuccess
                || hipMalloc(reinterpret_cast<void**>(&f->d_big_list), big.size() * 4) != hipSuccess)
            {
                gn_filter_free(f);
                return gn_fail(GN_ENOMEM, "target map allocation failed");
            }
            hipMemcpy(f->d_bin_tgt, bin_tgt.data(), bin_tgt.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(f->d_bin_nb2, nb2.data(), nb2.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(f->d_big_list, big.data(), big.size() * 4, hipMemcpyHostToDevice);
            // split kernel (register counters + byte image): the same bytes in the layout of the byte counters --
            // register r = (d*4+j)*2+pp of lane l at [(slice*8*nd + r)*64 + l], byte y <-> bit 8y + 4pp + j of dword d
            const size_t split_lds = gn_split_lds_bytes(geom, ibf->hash_funs);
            if (geom.gp_log2 == 6 && split_lds <= 160u * 1024u)
            {
                const uint32_t        lw = geom.lw, nd = 2u * lw;
                std::vector<uint32_t> nbr((size_t)geom.wpr * 8u * nd * 64u, 0u);
                for (uint64_t b = 0; b < ibf->bins; ++b)
                {
                    const uint32_t t = bin2target[b];
                    if (t == 0xFFFFFFFFu)
                        continue;
                    const uint32_t len = off[t + 1] - off[t];
                    if (len > GN_CAND_NBIG)
                        continue;
                    const uint32_t word = (uint32_t)(b >> 6), sl = word / (64u * lw), wrel = word - sl * 64u * lw, lane = wrel / lw;
                    const uint32_t tp = (wrel - lane * lw) * 64u + (uint32_t)(b & 63u), d = tp >> 5, bit = tp & 31u;
                    const uint32_t r  = (d * 4u + (bit & 3u)) * 2u + ((bit >> 2) & 1u);
                    nbr[((size_t)sl * 8u * nd + r) * 64u + lane] |= len << (8u * (bit >> 3));
                }
                if (hipMalloc(reinterpret_cast<void**>(&f->d_sl_nbr), nbr.size() * 4) != hipSuccess)
                {
                    gn_filter_free(f);
                    return gn_fail(GN_ENOMEM, "target map allocation failed");
                }
                hipMemcpy(f->d_sl_nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice);
                f->split_bpc = (uint32_t)((160u * 1024u) / split_lds);
            }
        }
    }
    *out = f;
    return GN_OK;
}

int gn_hibf_build(gn_filter* f, uint32_t n_ibf, const gn_ibf_desc* ibfs, const int64_t* const* next_ibf_id,
                  const int64_t* const* bin2userbin, uint64_t n_user_bins); // gn_hibf.hip

extern "C" int gn_filter_upload_hibf(int device, uint32_t n_ibf, const gn_ibf_desc* ibfs, const int64_t* const* next_ibf_id,
                                     const int64_t* const* bin2userbin, uint64_t n_user_bins, gn_filter** out)
{
    if (!out || !ibfs || !next_ibf_id || !bin2userbin || n_ibf == 0)
        return gn_fail(GN_EINVAL, "gn_filter_upload_hibf: null/empty argument");
    *out = nullptr;
    int rc = gn_set_device(device);
    if (rc)
        return rc;
    gn_filter* f = new (std::nothrow) gn_filter();
    if (!f)
        return gn_fail(GN_ENOMEM, "out of host memory");
    f->device  = device;
    f->is_hibf = true;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        f->n_cu = prop.multiProcessorCount;
    f->ibfs.resize(n_ibf);
    for (uint32_t i = 0; i < n_ibf; ++i)
    {
        rc = gn_upload_ibf_rows(&ibfs[i], &f->ibfs[i], &f->device_bytes, gn_hibf_row_stride_words(ibfs[i].bin_words) != ibfs[i].bin_words);
        if (rc)
        {
            gn_filter_free(f);
            return rc;
        }
    }
    rc = gn_hibf_build(f, n_ibf, ibfs, next_ibf_id, bin2userbin, n_user_bins);
    if (rc)
    {
        gn_filter_free(f);
        return rc;
    }
    *out = f;
    return GN_OK;
}

extern "C" int gn_filter_free(gn_filter* f)
{
    if (!f)
        return GN_OK;
    hipSetDevice(f->device);
    if (f->ibf.d_rows)
        hipFree(f->ibf.d_rows);
    if (f->d_tgt_off)
        hipFree(f->d_tgt_off);
    if (f->d_tgt_bins)
        hipFree(f->d_tgt_bins);
    if (f->d_tgt_lds)
        hipFree(f->d_tgt_lds);
    if (f->d_tgt_rec)
        hipFree(f->d_tgt_rec);
    if (f->d_bin_tgt)
        hipFree(f->d_bin_tgt);
    if (f->d_bin_nb2)
        hipFree(f->d_bin_nb2);
    if (f->d_big_list)
        hipFree(f->d_big_list);
    if (f->d_sl_nbr)
        hipFree(f->d_sl_nbr);
    for (auto& i : f->ibfs)
        if (i.d_rows)
            hipFree(i.d_rows);
    for (void* p : f->hibf_allocs)
        hipFree(p);
    if (f->d_hibf)
        hipFree(f->d_hibf);
    if (f->d_emplace_stage)
        hipFree(f->d_emplace_stage);
    if (f->load_st)
        hipStreamDestroy(f->load_st);
    delete f;
    return GN_OK;
}

extern "C" int gn_filter_info(const gn_filter* f, int* is_hibf, uint32_t* n_ibf, uint64_t* n_targets, uint64_t* device_bytes)
{
    if (!f)
        return gn_fail(GN_EINVAL, "null filter");
    if (is_hibf)
        *is_hibf = f->is_hibf ? 1 : 0;
    if (n_ibf)
        *n_ibf = f->is_hibf ? (uint32_t)f->ibfs.size() : 1u;
    if (n_targets)
        *n_targets = f->is_hibf ? f->n_user_bins : f->n_targets;
    if (device_bytes)
        *device_bytes = f->device_bytes;
    return GN_OK;
}

// IBF `ibf_idx` of a filter (0 for a flat one); nullptr + error message when out of range
static GnIbfHost* gn_filter_ibf(gn_filter* f, uint32_t ibf_idx)
{
    if (!f)
    {
        gn_fail(GN_EINVAL, "null filter");
        return nullptr;
    }
    if (f->is_hibf)
    {
        if (ibf_idx >= f->ibfs.size())
        {
            gn_fail(GN_EINVAL, "ibf index %u out of range (%zu IBFs)", ibf_idx, f->ibfs.size());
            return nullptr;
        }
        return &f->ibfs[ibf_idx];
    }
    if (ibf_idx != 0)
    {
        gn_fail(GN_EINVAL, "ibf index %u out of range (flat filter)", ibf_idx);
        return nullptr;
    }
    return &f->ibf;
}

extern "C" int gn_filter_emplace_ibf(gn_filter* f, uint32_t ibf_idx, const uint64_t* hashes, const uint32_t* bins, uint64_t n)
{
    GnIbfHost* ib = gn_filter_ibf(f, ibf_idx);
    if (!ib)
        return GN_EINVAL;
    if (n == 0)
        return GN_OK;
    if (!hashes || !bins)
        return gn_fail(GN_EINVAL, "null argument");
    for (uint64_t i = 0; i < n; ++i)
        if (bins[i] >= ib->B)
            return gn_fail(GN_EINVAL, "bin %u out of range", bins[i]);
    GN_HIP(hipSetDevice(f->device));
    uint64_t* dh = nullptr;
    uint32_t* db = nullptr;
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&dh), n * 8));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&db), n * 4);
    if (e != hipSuccess)
    {
        hipFree(dh);
        return gn_fail(GN_ENOMEM, "emplace staging allocation failed");
    }
    hipMemcpy(dh, hashes, n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, bins, n * 4, hipMemcpyHostToDevice);
    e = gn_launch_emplace(ib->d_rows, ib->S, (uint32_t)ib->Ws, ib->shift, ib->h, dh, db, n, nullptr); // (the kernel's W is the row stride)
    hipError_t e2 = hipDeviceSynchronize();
    hipFree(dh);
    hipFree(db);
    if (e != hipSuccess || e2 != hipSuccess)
        return gn_fail(GN_ENODEV, "emplace kernel failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GN_OK;
}

extern "C" int gn_filter_emplace(gn_filter* f, const uint64_t* hashes, const uint32_t* bins, uint64_t n)
{
    if (!f || f->is_hibf)
        return gn_fail(GN_EINVAL, "gn_filter_emplace needs a flat IBF filter");
    return gn_filter_emplace_ibf(f, 0, hashes, bins, n);
}

extern "C" int gn_filter_download_rows(const gn_filter* f, uint32_t ibf_idx, uint64_t row_begin, uint64_t n_rows, uint64_t* out)
{
    if (!f || !out)
        return gn_fail(GN_EINVAL, "null argument");
    const GnIbfHost* ib = nullptr;
    if (f->is_hibf)
    {
        if (ibf_idx >= f->ibfs.size())
            return gn_fail(GN_EINVAL, "ibf index out of range");
        ib = &f->ibfs[ibf_idx];
    }
    else
    {
        if (ibf_idx != 0)
            return gn_fail(GN_EINVAL, "ibf index out of range");
        ib = &f->ibf;
    }
    if (row_begin + n_rows > ib->S)
        return gn_fail(GN_EINVAL, "row range out of bounds");
    GN_HIP(hipSetDevice(f->device));
    if (ib->Ws == ib->W)
        GN_HIP(hipMemcpy(out, ib->d_rows + row_begin * ib->W, n_rows * ib->W * 8, hipMemcpyDeviceToHost));
    else
        GN_HIP(hipMemcpy2D(out, ib->W * 8, ib->d_rows + row_begin * ib->Ws, ib->Ws * 8, ib->W * 8, n_rows, hipMemcpyDeviceToHost));
    return GN_OK;
}

__global__ void gn_gather_rows_kernel(const uint64_t* __restrict__ rows, uint64_t W, uint64_t Ws, const uint64_t* __restrict__ idx,
                                      uint64_t* __restrict__ out)
{
    const uint64_t r = idx[blockIdx.x];
    for (uint64_t j = threadIdx.x; j < W; j += blockDim.x)
        out[(uint64_t)blockIdx.x * W + j] = rows[r * Ws + j];
}

extern "C" int gn_filter_download_row_list(const gn_filter* f, uint32_t ibf_idx, const uint64_t* row_idx, uint64_t n, uint64_t* out)
{
    const GnIbfHost* ib = gn_filter_ibf(const_cast<gn_filter*>(f), ibf_idx);
    if (!ib)
        return GN_EINVAL;
    if (n == 0)
        return GN_OK;
    if (!row_idx || !out)
        return gn_fail(GN_EINVAL, "null argument");
    for (uint64_t i = 0; i < n; ++i)
        if (row_idx[i] >= ib->S)
            return gn_fail(GN_EINVAL, "row %llu out of range", (unsigned long long)row_idx[i]);
    GN_HIP(hipSetDevice(f->device));
    const uint64_t per = std::max<uint64_t>(1, (256ull << 20) / (ib->W * 8)); // rows per 256 MiB staging round
    const uint64_t cap = std::min(per, n);
    uint64_t *     d_idx = nullptr, *d_out = nullptr;
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), cap * 8));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), cap * ib->W * 8);
    if (e != hipSuccess)
    {
        hipFree(d_idx);
        return gn_fail(GN_ENOMEM, "row gather staging allocation failed");
    }
    for (uint64_t o = 0; o < n && e == hipSuccess; o += cap)
    {
        const uint64_t m = std::min(cap, n - o);
        e                = hipMemcpy(d_idx, row_idx + o, m * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess)
            break;
        hipLaunchKernelGGL(gn_gather_rows_kernel, dim3((unsigned)m), dim3(64), 0, nullptr, ib->d_rows, ib->W, ib->Ws, d_idx, d_out);
        e = hipMemcpy(out + o * ib->W, d_out, m * ib->W * 8, hipMemcpyDeviceToHost);
    }
    hipFree(d_idx);
    hipFree(d_out);
    if (e != hipSuccess)
        return gn_fail(GN_ENODEV, "row gather failed: %s", hipGetErrorString(e));
    return GN_OK;
}

// ---- streaming load ------------------------------------------------------------------------------
// Page-locked host memory.  hipHostMalloc locks 4 KiB pages at ~5 GB/s; an anonymous mapping backed by transparent huge pages,
// touched and then registered (hipHostRegister), is ready at ~26 GB/s and copies to the device just as fast (57 GB/s) --
// scripts/pin_thp_probe.cpp, profiles/r05_pin_thp_probe.jsonl.  A run of a few seconds locks gigabytes (batch pool, staging, result
// sets), so this is most of its start-up.  Falls back to hipHostMalloc where the mapping or the registration is refused.
namespace
{
struct GnPinned
{
    void*  raw;
    size_t raw_bytes;
};
std::mutex                           g_pinned_mutex;
std::unordered_map<void*, GnPinned> g_pinned; // blocks that came from mmap + hipHostRegister
} // namespace

extern "C" int gn_pinned_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0)
        return gn_fail(GN_EINVAL, "gn_pinned_alloc: bad argument");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return gn_fail(GN_ENODEV, "no HIP device available (libganon_hip has no CPU fallback)");
    constexpr size_t huge = 2u << 20;
    if (bytes >= huge && !gn_sw().pinned_malloc)
    {
        const size_t len = (bytes + huge - 1) & ~(huge - 1);
        void*        raw = mmap(nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != MAP_FAILED)
        {
            void* p = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(raw) + huge - 1) & ~(uintptr_t)(huge - 1));
            (void)madvise(p, len, MADV_HUGEPAGE);
            for (size_t i = 0; i < len; i += 4096) // fault the pages in (one fault per huge page where they are granted)
                static_cast<volatile char*>(p)[i] = 0;
            if (hipHostRegister(p, len, hipHostRegisterPortable) == hipSuccess)
            {
                std::lock_guard<std::mutex> lk(g_pinned_mutex);
                g_pinned[p] = GnPinned{ raw, len + huge };
                *out        = p;
                return GN_OK;
            }
            (void)hipGetLastError();
            munmap(raw, len + huge);
        }
    }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess)
        return gn_fail(GN_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return GN_OK;
}

extern "C" int gn_pinned_free(void* p)
{
    if (!p)
        return GN_OK;
    GnPinned blk{ nullptr, 0 };
    {
        std::lock_guard<std::mutex> lk(g_pinned_mutex);
        auto                        it = g_pinned.find(p);
        if (it != g_pinned.end())
        {
            blk = it->second;
            g_pinned.erase(it);
        }
    }
    if (blk.raw)
    {
        (void)hipHostUnregister(p);
        munmap(blk.raw, blk.raw_bytes);
    }
    else
        hipHostFree(p);
    return GN_OK;
}

static int gn_filter_load_stream(gn_filter* f)
{
    GN_HIP(hipSetDevice(f->device));
    if (!f->load_st)
        GN_HIP(hipStreamCreateWithFlags(&f->load_st, hipStreamNonBlocking));
    return GN_OK;
}

extern "C" int gn_filter_write_rows(gn_filter* f, uint32_t ibf_idx, uint64_t row_begin, uint64_t n_rows, const uint64_t* src,
                                    uint64_t src_row_words, uint64_t word_lo)
{
    GnIbfHost* ib = gn_filter_ibf(f, ibf_idx);
    if (!ib)
        return GN_EINVAL;
    if (n_rows == 0)
        return GN_OK;
    if (!src)
        return gn_fail(GN_EINVAL, "null argument");
    if (row_begin + n_rows > ib->S || word_lo + ib->W > src_row_words)
        return gn_fail(GN_EINVAL, "gn_filter_write_rows: rows [%llu,+%llu) x words [%llu,+%llu) outside the filter (%llu rows) / "
                                  "the source rows (%llu words)",
                       (unsigned long long)row_begin, (unsigned long long)n_rows, (unsigned long long)word_lo,
                       (unsigned long long)ib->W, (unsigned long long)ib->S, (unsigned long long)src_row_words);
    int rc = gn_filter_load_stream(f);
    if (rc)
        return rc;
    uint64_t* dst = ib->d_rows + row_begin * ib->Ws;
    if (src_row_words == ib->W && ib->Ws == ib->W)
        GN_HIP(hipMemcpyAsync(dst, src, n_rows * ib->W * 8, hipMemcpyHostToDevice, f->load_st));
    else
        GN_HIP(hipMemcpy2DAsync(dst, ib->Ws * 8, src + word_lo, src_row_words * 8, ib->W * 8, n_rows, hipMemcpyHostToDevice,
                                f->load_st));
    return GN_OK;
}

extern "C" int gn_filter_write_sync(gn_filter* f)
{
    if (!f)
        return gn_fail(GN_EINVAL, "null filter");
    GN_HIP(hipSetDevice(f->device));
    if (f->load_st)
        GN_HIP(hipStreamSynchronize(f->load_st));
    return GN_OK;
}

static int gn_clear_padding(gn_filter* f, GnIbfHost* ib, hipStream_t st)
{
    if (ib->B & 63)
    {
        hipLaunchKernelGGL(gn_clear_padding_kernel, dim3((unsigned)((ib->S + 255) / 256)), dim3(256), 0, st, ib->d_rows, ib->S,
                           ib->W, ib->Ws, (1ull << (ib->B & 63)) - 1ull);
        GN_HIP(hipGetLastError());
    }
    return GN_OK;
}

extern "C" int gn_filter_finalize(gn_filter* f)
{
    if (!f)
        return gn_fail(GN_EINVAL, "null filter");
    int rc = gn_filter_load_stream(f);
    if (rc)
        return rc;
    if (f->is_hibf)
    {
        for (auto& ib : f->ibfs)
            if ((rc = gn_clear_padding(f, &ib, f->load_st)) != GN_OK)
                return rc;
    }
    else if ((rc = gn_clear_padding(f, &f->ibf, f->load_st)) != GN_OK)
        return rc;
    GN_HIP(hipStreamSynchronize(f->load_st));
    return GN_OK;
}

// ---- synthetic fill ------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t gn_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void gn_fill_random_kernel(uint64_t* __restrict__ rows, uint64_t S, uint64_t W, uint64_t Ws, uint64_t seed,
                                                             uint32_t and_words, uint64_t word_lo, uint64_t row_words_total,
                                                             uint64_t last_mask)
{
    const uint64_t total  = S * W;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t       keys[8];
    for (uint32_t a = 0; a < (and_words == GN_FILL_3_OF_8 ? 3u : and_words == GN_FILL_3_OF_16 ? 4u : and_words); ++a)
        keys[a] = gn_mix64(seed + a);
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride)
    {
        const uint64_t r = idx / W, j = idx - r * W;
        const uint64_t g = (r * row_words_total + word_lo + j) * 0x9E3779B97F4A7C15ULL;
        uint64_t       v = ~0ULL;
        if (and_words == GN_FILL_3_OF_8) // a & (b | c): density 3/8, i.e. p^3 = 0.053 for three hash functions
            v = gn_mix64(keys[0] + g) & (gn_mix64(keys[1] + g) | gn_mix64(keys[2] + g));
        else if (and_words == GN_FILL_3_OF_16) // a & b & (c | d): density 3/16, i.e. p^4 = 0.0012 for four hash functions
            v = gn_mix64(keys[0] + g) & gn_mix64(keys[1] + g) & (gn_mix64(keys[2] + g) | gn_mix64(keys[3] + g));
        else
            for (uint32_t a = 0; a < and_words; ++a)
                v &= gn_mix64(keys[a] + g);
        if (j == W - 1)
            v &= last_mask;
        rows[r * Ws + j] = v; // (Ws: the device's row stride; the words beyond W of a padded row stay zero)
    }
}

extern "C" int gn_filter_fill_random(gn_filter* f, uint32_t ibf_idx, uint64_t seed, uint32_t and_words, uint64_t word_lo,
                                     uint64_t row_words_total)
{
    GnIbfHost* ib = gn_filter_ibf(f, ibf_idx);
    if (!ib)
        return GN_EINVAL;
    if ((and_words < 1 || and_words > 8) && and_words != GN_FILL_3_OF_8 && and_words != GN_FILL_3_OF_16)
        return gn_fail(GN_EINVAL, "and_words must be 1..8 (or GN_FILL_3_OF_8 / GN_FILL_3_OF_16)");
    if (row_words_total == 0)
        row_words_total = ib->W;
    if (word_lo + ib->W > row_words_total)
        return gn_fail(GN_EINVAL, "column slice [%llu,+%llu) outside rows of %llu words", (unsigned long long)word_lo,
                       (unsigned long long)ib->W, (unsigned long long)row_words_total);
    GN_HIP(hipSetDevice(f->device));
    const uint64_t total  = ib->S * ib->W;
    const uint64_t want   = (total + 255) / 256;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(want, (uint64_t)f->n_cu * 32);
    const uint64_t mask   = (ib->B & 63) ? (1ull << (ib->B & 63)) - 1ull : ~0ull;
    hipLaunchKernelGGL(gn_fill_random_kernel, dim3(blocks), dim3(256), 0, nullptr, ib->d_rows, ib->S, ib->W, ib->Ws, seed, and_words,
                       word_lo, row_words_total, mask);
    GN_HIP(hipGetLastError());
    GN_HIP(hipDeviceSynchronize());
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------
// streams
// ------------------------------------------------------------------------------------------------
template <typename T>
static hipError_t gn_dmalloc(T** p, size_t n)
{
    return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}

extern "C" int gn_stream_destroy(gn_stream* s)
{
    if (!s)
        return GN_OK;
    hipSetDevice(s->device); // (not s->f->device: a caller may have freed the filter first)
    if (s->st)
        hipStreamSynchronize(s->st);
    void* ptrs[] = { s->d_bases,  s->d_off1,    s->d_off2,      s->d_slot_cnt,  s->d_slot_off, s->d_hashes, s->d_nh,
                     s->d_status, s->d_matches, s->d_sorted,    s->d_ctr,       s->d_seg_begin, s->d_seg_count,
                     s->d_seg_off, s->d_deferred, s->d_mdeferred, s->d_scan_tmp, s->d_work[0], s->d_work[1], s->d_hdefer, s->d_hdefer2, s->d_hctr, s->d_hsub, s->d_keys[0], s->d_keys[1], s->d_vals[0],
                     s->d_vals[1], s->d_sort_tmp };
    for (void* p : ptrs)
        if (p)
            hipFree(p);
    gn_postfilter_release(s);
    gn_build_release(s);
    gn_fastq_release(s);
    for (void* q : { (void*)s->d_long_list, (void*)s->d_long_count, (void*)s->d_long_scratch })
        if (q)
            hipFree(q);
    if (s->h_ctr)
        hipHostFree(s->h_ctr);
    if (s->h_hctr)
        hipHostFree(s->h_hctr);
    for (auto& e : s->ev)
        if (e)
            hipEventDestroy(e);
    for (auto& e : s->ev_chunk)
        if (e)
            hipEventDestroy(e);
    for (auto& e : s->ev_lvl)
        if (e)
            hipEventDestroy(e);
    if (s->ev_sync)
        hipEventDestroy(s->ev_sync);
    for (auto& e : s->ev_cmp)
        if (e)
            hipEventDestroy(e);
    if (s->ev_count0)
        hipEventDestroy(s->ev_count0);
    if (s->st2 && s->st2 != s->st)
    {
        hipStreamSynchronize(s->st2);
        hipStreamDestroy(s->st2);
    }
    if (s->st)
        hipStreamDestroy(s->st);
    delete s;
    return GN_OK;
}

extern "C" int gn_stream_create(gn_filter* f, uint32_t max_reads, uint64_t max_bases, uint64_t max_matches, gn_stream** out)
{
    if (!f || !out || max_reads == 0 || max_bases == 0)
        return gn_fail(GN_EINVAL, "gn_stream_create: bad argument");
    if (f->storage_only)
        return gn_fail(GN_EINVAL, "gn_stream_create: the filter was created without a bin map (storage only)");
    *out = nullptr;
    GN_HIP(hipSetDevice(f->device));
    gn_stream* s = new (std::nothrow) gn_stream();
    if (!s)
        return gn_fail(GN_ENOMEM, "out of host memory");
    s->f         = f;
    s->device    = f->device;
    s->max_reads = max_reads;
    s->max_bases = max_bases;
    s->match_cap = max_matches ? max_matches : (uint64_t)max_reads * 4;
    const uint32_t wpr  = f->is_hibf ? 1 : f->geom.wpr;
    const size_t   nseg = (size_t)max_reads * wpr;
    hipError_t     e    = hipSuccess;
    auto           ok   = [&](hipError_t x) {
        if (e == hipSuccess)
            e = x;
    };
    ok(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    // The minimiser kernels may run on a side stream, one chunk ahead of the count kernels ($GANON_HIP_CHUNK: the tests' way of running a
    // batch in read ranges).
    // With one chunk per batch -- the default: chunking gained nothing, see gn_stream_classify -- a second stream per batch context
    // only costs: the runtime maps all streams of a process onto a few hardware queues (4 by default), and streams that share a
    // queue wait for each other's copies and kernels.  One stream per context keeps a worker's batches independent of the others'.
    if (gn_sw().chunk)
        ok(hipStreamCreateWithFlags(&s->st2, hipStreamNonBlocking));
    else
        s->st2 = s->st;
    for (auto& ev : s->ev)
        ok(hipEventCreate(&ev));
    ok(hipEventCreate(&s->ev_sync));
    ok(hipEventCreate(&s->ev_count0));
    for (auto& ev : s->ev_chunk)
        ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ok(gn_dmalloc(&s->d_bases, max_bases + 64));
    ok(gn_dmalloc(&s->d_off1, (size_t)max_reads + 1));
    ok(gn_dmalloc(&s->d_off2, (size_t)max_reads + 1));
    ok(gn_dmalloc(&s->d_slot_cnt, (size_t)max_reads + 1));
    ok(gn_dmalloc(&s->d_slot_off, (size_t)max_reads + 1));
    ok(gn_dmalloc(&s->d_hashes, max_bases)); // #windows <= #bases
    ok(gn_dmalloc(&s->d_nh, max_reads));
    ok(gn_dmalloc(&s->d_status, (size_t)max_reads + 8)); // (+8: the HIBF level-0 kernel reads status bytes as aligned dwords)
    ok(gn_dmalloc(&s->d_matches, s->match_cap));
    ok(gn_dmalloc(&s->d_sorted, s->match_cap));
    ok(gn_dmalloc(&s->d_ctr, GN_NCTR));
    ok(gn_dmalloc(&s->d_seg_begin, nseg));
    ok(gn_dmalloc(&s->d_seg_count, nseg + 1));
    ok(gn_dmalloc(&s->d_seg_off, nseg + 1));
    ok(gn_dmalloc(&s->d_deferred, max_reads));
    ok(gn_dmalloc(&s->d_mdeferred, max_reads));
    size_t tmp1 = 0, tmp2 = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tmp1, s->d_slot_cnt, s->d_slot_off, (int)(max_reads + 1), s->st);
    gn_scan_counts(nullptr, tmp2, s->d_seg_count, s->d_seg_off, (int)(nseg + 1), s->st);
    s->scan_tmp_bytes = std::max(tmp1, tmp2) + 256;
    ok(hipMalloc(&s->d_scan_tmp, s->scan_tmp_bytes));
    if (f->is_hibf)
    {
        s->work_cap = max_reads * 4u > 1024u ? max_reads * 4u : 1024u;
        ok(gn_dmalloc(&s->d_work[0], s->work_cap));
        ok(gn_dmalloc(&s->d_work[1], s->work_cap));
        ok(gn_dmalloc(&s->d_hdefer, s->work_cap));
        ok(gn_dmalloc(&s->d_hdefer2, s->work_cap));
        ok(gn_dmalloc(&s->d_hctr, 37 * (GN_HIBF_MAXDEPTH + 1) + 2));
        ok(gn_dmalloc(&s->d_hsub, 384 * (GN_HIBF_MAXDEPTH + 1))); // per level: counts, bases, cursors of the 128 (class, n-bin) keys
        ok(hipHostMalloc(reinterpret_cast<void**>(&s->h_hctr), (5 * (GN_HIBF_MAXDEPTH + 1) + 2) * sizeof(unsigned long long), hipHostMallocDefault));
    }
    ok(hipHostMalloc(reinterpret_cast<void**>(&s->h_ctr), GN_NCTR * sizeof(unsigned long long), hipHostMallocDefault));
    if (e != hipSuccess)
    {
        gn_stream_destroy(s);
        return gn_fail(e == hipErrorOutOfMemory ? GN_ENOMEM : GN_ENODEV, "gn_stream_create: %s", hipGetErrorString(e));
    }
    s->v_hashes   = s->d_hashes;
    s->v_slot_off = s->d_slot_off;
    s->v_nh       = s->d_nh;
    s->v_status   = s->d_status;
    *out = s;
    return GN_OK;
}

extern "C" int gn_stream_upload_reads(gn_stream* s, const uint8_t* bases, uint64_t n_bases, const uint64_t* off1,
                                      const uint64_t* off2, uint32_t n_reads)
{
    if (!s || !off1 || (!bases && n_bases))
        return gn_fail(GN_EINVAL, "gn_stream_upload_reads: null argument");
    if (n_reads > s->max_reads || n_bases > s->max_bases)
        return gn_fail(GN_EINVAL, "batch (%u reads, %llu bases) exceeds the stream capacity (%u, %llu)", n_reads,
                       (unsigned long long)n_bases, s->max_reads, (unsigned long long)s->max_bases);
    if (n_reads && (off1[n_reads] > n_bases || (off2 && off2[n_reads] > n_bases)))
        return gn_fail(GN_EINVAL, "read offsets exceed n_bases");
    GN_HIP(hipSetDevice(s->f->device));
    GN_HIP(hipStreamSynchronize(s->st)); // previous batch must be done before its inputs are overwritten
    if (n_bases)
        GN_HIP(hipMemcpyAsync(s->d_bases, bases, n_bases, hipMemcpyHostToDevice, s->st));
    GN_HIP(hipMemcpyAsync(s->d_off1, off1, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, s->st));
    if (off2)
        GN_HIP(hipMemcpyAsync(s->d_off2, off2, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, s->st));
    s->v_hashes   = s->d_hashes; // (a batch of its own again)
    s->v_slot_off = s->d_slot_off;
    s->v_nh       = s->d_nh;
    s->v_status   = s->d_status;
    s->src        = nullptr;
    s->n_reads    = n_reads;
    s->n_bases    = n_bases;
    s->paired     = off2 != nullptr;
    s->have_reads = true;
    s->classified = false;
    s->ctr_copied = false;
    s->hashed     = false;
    s->build_distinct = ~0ull;
    return GN_OK;
}

// small helper kernels (batch bookkeeping) ---------------------------------------------------------
__global__ void gn_slot_count_kernel(const uint64_t* off1, const uint64_t* off2, uint32_t n_reads, uint32_t w, uint64_t* cnt)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads)
        return;
    uint64_t c = 0;
    if (r < n_reads)
    {
        const uint64_t l1 = off1[r + 1] - off1[r];
        if (l1 >= w)
        {
            c = l1 - w + 1;
            if (off2)
            {
                const uint64_t l2 = off2[r + 1] - off2[r];
                if (l2 >= w)
                    c += l2 - w + 1;
            }
        }
    }
    cnt[r] = c;
}

// Group the matches by read: read r's segments (one per column slice, written by the count kernels into wave-private
// chunks of d_matches) are copied behind each other to out[seg_off[r*wpr] ...), ascending target inside a read.
//   * reads with at most GN_GATHER_SMALL matches (the usual case: 0-2 per read) are handled one per lane, with an
//     insertion sort by target (the candidate-driven select emits a slice's few hits in no particular order);
//   * reads with more (low cutoffs: ~100 chance matches per read at --rel-cutoff 0.2 on a 4096-bin filter) are handled
//     by the whole wave one after the other: coalesced 12-byte copies, with an order check on the way; the count kernels
//     emit ascending targets, so the rank sort behind the check is a rarely taken fallback (O(c^2/64)).
#define GN_GATHER_SMALL 6u
__global__ void gn_gather_kernel(const gn_match* __restrict__ in, gn_match* __restrict__ out, const uint64_t* __restrict__ seg_begin,
                                 const uint32_t* __restrict__ seg_count, const uint64_t* __restrict__ seg_off, uint64_t n_reads,
                                 uint32_t wpr, const unsigned long long* __restrict__ cursor, uint64_t cap)
{
    if (*cursor > cap) // overflowed batch: nothing was written, gn_finish() grows the buffers and re-runs
        return;
    const uint64_t r     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane  = threadIdx.x & 63u;
    const bool     valid = r < n_reads;
    uint64_t       o     = 0;
    uint32_t       c     = 0;
    if (valid)
    {
        o = seg_off[r * wpr];
        c = (uint32_t)(seg_off[(r + 1) * wpr] - o);
    }
    if (valid && c != 0 && c <= GN_GATHER_SMALL)
    {
        uint32_t k_total = 0;
        for (uint32_t sl = 0; sl < wpr; ++sl)
        {
            const uint32_t cs = seg_count[r * wpr + sl];
            const uint64_t b  = seg_begin[r * wpr + sl];
            for (uint32_t j = 0; j < cs; ++j)
            {
                const gn_match m = in[b + j];
                uint32_t       k = k_total;
                while (k > 0 && out[o + k - 1].target > m.target)
                {
                    out[o + k] = out[o + k - 1];
                    --k;
                }
                out[o + k] = m;
                ++k_total;
            }
        }
    }
    uint64_t heavy = __ballot(valid && c > GN_GATHER_SMALL);
    while (heavy)
    {
        const uint32_t L = (uint32_t)__builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const uint64_t rr = r - lane + L;
        const uint64_t oo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), (int)L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, (int)L);
        uint32_t done = 0, prev_last = 0;
        bool     have_prev = false, disorder = false;
        for (uint32_t sl = 0; sl < wpr; ++sl)
        {
            const uint32_t cs = seg_count[rr * wpr + sl];
            const uint64_t b  = seg_begin[rr * wpr + sl];
            for (uint32_t j0 = 0; j0 < cs; j0 += 64)
            {
                const uint32_t j   = j0 + lane;
                const bool     act = j < cs;
                gn_match       m{};
                if (act)
                    m = in[b + j];
                const uint32_t t  = act ? m.target : 0xFFFFFFFFu;
                uint32_t       tp = (uint32_t)__shfl_up((int)t, 1);
                if (lane == 0)
                    tp = have_prev ? prev_last : 0u;
                disorder = disorder || __ballot(act && tp > t) != 0;
                if (act)
                    out[oo + done + j] = m;
                const uint32_t nact = cs - j0 < 64u ? cs - j0 : 64u;
                prev_last = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)(nact - 1));
                have_prev = true;
            }
            done += cs;
        }
        if (disorder)
        {
            // rank of every element among the read's matches (a target occurs once per read), straight from `in`
            for (uint32_t sl = 0; sl < wpr; ++sl)
            {
                const uint32_t cs = seg_count[rr * wpr + sl];
                const uint64_t b  = seg_begin[rr * wpr + sl];
                for (uint32_t j0 = 0; j0 < cs; j0 += 64)
                {
                    const uint32_t j   = j0 + lane;
                    const bool     act = j < cs;
                    gn_match       m{};
                    if (act)
                        m = in[b + j];
                    uint32_t rank = 0;
                    for (uint32_t s2 = 0; s2 < wpr; ++s2)
                    {
                        const uint32_t c2 = seg_count[rr * wpr + s2];
                        const uint64_t b2 = seg_begin[rr * wpr + s2];
                        for (uint32_t e = 0; e < c2; ++e)
                            rank += in[b2 + e].target < m.target ? 1u : 0u;
                    }
                    if (act)
                        out[oo + rank] = m;
                }
            }
        }
    }
}

// Segmented results (gn_run_group): the reads of `list` (what the fast kernel deferred to the generic one) may hold their matches in any
// order; one wave per read checks and, where needed, ranks them by target through the scratch buffer (a target occurs once per read).
__global__ void gn_seg_order_kernel(const uint32_t* __restrict__ list, const unsigned long long* __restrict__ n_list, gn_match* m, gn_match* scratch,
                                    const uint64_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_count,
                                    const uint64_t* __restrict__ seg_off, const unsigned long long* __restrict__ cursor, uint64_t cap)
{
    if (*cursor > cap)
        return;
    const uint32_t lane   = threadIdx.x & 63u;
    const uint64_t waves  = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t n      = *n_list;
    for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += waves)
    {
        const uint32_t r = list[i];
        const uint32_t c = seg_count[r];
        if (c < 2)
            continue;
        gn_match* seg = m + seg_begin[r];
        bool      bad = false;
        for (uint32_t j = lane; j + 1 < c; j += 64)
            bad = bad || seg[j].target > seg[j + 1].target;
        if (!__ballot(bad))
            continue;
        gn_match* tmp = scratch + seg_off[r];
        for (uint32_t j = lane; j < c; j += 64)
        {
            const gn_match x    = seg[j];
            uint32_t       rank = 0;
            for (uint32_t e = 0; e < c; ++e)
                rank += seg[e].target < x.target ? 1u : 0u;
            tmp[rank] = x;
        }
        __threadfence();
        for (uint32_t j = lane; j < c; j += 64)
            seg[j] = tmp[j];
    }
}

int gn_hibf_classify(gn_stream* s, gn_filter* f, hipStream_t st); // gn_hibf.hip

// $GANON_HIP_ABLATE=fake_count (host-ceiling measurement, include/ganon_hip.h): the result of a count + select that was never run -- every
// second read that has minimisers gets one match (target read % n_targets, count = its minimisers), so that the host's post stage has
// the usual share of classified reads to write
__global__ void gn_fake_count_kernel(gn_match* matches, uint64_t* seg_begin, uint32_t* seg_count, const uint8_t* status, const uint32_t* n_hashes,
                                     uint32_t lo, uint32_t hi, uint32_t wpr, uint32_t n_targets, unsigned long long* cursor)
{
    const uint32_t r = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= hi)
        return;
    const bool hit = (r & 1u) && status[r] == GN_READ_OK && n_hashes[r] != 0;
    for (uint32_t sl = 0; sl < wpr; ++sl)
    {
        seg_begin[(size_t)r * wpr + sl] = r >> 1;
        seg_count[(size_t)r * wpr + sl] = hit && sl == 0 ? 1u : 0u;
    }
    if (hit)
    {
        gn_match m;
        m.read   = r;
        m.target = r % n_targets;
        m.count  = n_hashes[r];
        matches[r >> 1] = m;
    }
    if (r + 1 == hi)
        atomicMax(cursor, (unsigned long long)((hi + 1) >> 1));
}

// count + select over reads [lo, hi) on the stream's main HIP stream (the match cursor is NOT reset here)
static int gn_run_count_range(gn_stream* s, uint32_t lo, uint32_t hi)
{
    gn_filter* f = s->f;
    GnCountParams p{};
    p.rows       = f->ibf.d_rows;
    p.S          = f->ibf.S;
    p.W          = (uint32_t)f->ibf.W;
    p.B          = (uint32_t)f->ibf.B;
    p.shift      = f->ibf.shift;
    p.tgt_off    = f->identity ? nullptr : f->d_tgt_off;
    p.tgt_bins   = f->d_tgt_bins;
    p.tgt_lds    = f->d_tgt_lds;
    p.tgt_rec    = f->d_tgt_rec;
    p.bin_tgt    = f->d_bin_tgt;
    p.bin_nb2    = gn_sw().cand_select ? nullptr : f->d_bin_nb2;
    p.nbtab_off  = (uint32_t)f->geom.nbtab_off;
    p.candcnt_off = (uint32_t)f->geom.candcnt_off;
    p.big_list   = f->d_big_list;
    p.n_big      = f->n_big;
    p.tgt_ids    = nullptr;
    p.n_targets  = f->n_targets;
    p.hashes     = s->v_hashes;
    p.slot_off   = s->v_slot_off;
    p.n_hashes   = s->v_nh;
    p.status     = s->v_status;
    p.n_reads    = hi;
    p.read_begin = lo;
    p.rel_cutoff = s->rel_cutoff;
    if (gn_sw().fake_count)
    {
        if (lo == 0)
            s->pf_predrop = false; // (every match is "written": a pre-pass judges them like any others)
        if ((uint64_t)(hi + 1) / 2 <= s->match_cap && hi > lo)
            hipLaunchKernelGGL(gn_fake_count_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, s->st, s->d_matches, s->d_seg_begin, s->d_seg_count,
                               s->v_status, s->v_nh, lo, hi, (uint32_t)f->geom.wpr, f->n_targets ? f->n_targets : 1u, s->d_ctr);
        GN_HIP(hipGetLastError());
        return GN_OK;
    }
    p.wpr        = f->geom.wpr;
    p.gp_log2    = f->geom.gp_log2;
    p.slice_dwords = f->geom.slice_dwords;
    p.matches    = s->d_matches;
    p.match_cap  = s->match_cap;
    p.cursor     = s->d_ctr;
    p.seg_begin  = s->d_seg_begin;
    p.seg_count  = s->d_seg_count;
    p.dense      = nullptr;
    p.max_blocks = (uint32_t)f->n_cu * 16u;
    // persistent grid = a whole number of resident rounds: 8-byte-lane variant holds 4 blocks per CU, 16-byte one 3
    p.max_blocks_fast = (uint32_t)f->n_cu * (f->geom.lw == 1 ? 8u : 6u);
    p.nt_loads = gn_sw().emit_probe; // (0 in the product; bits 6 / 7: the emission probe of the fast kernel's low-cutoff epilogue)
    p.early_exit = gn_sw().early_exit ? 0u : 1u; // (bench.py's every-row measurement and the parity tests of the exit)
    p.skip_ctr   = s->d_ctr + 7;
    if (lo == 0) // (a re-run after a match-buffer regrow starts the tally again)
        GN_HIP(hipMemsetAsync(s->d_ctr + 7, 0, sizeof(unsigned long long), s->st));
    p.sl_nbr = f->d_sl_nbr;
    p.csr_identity = f->csr_identity && !gn_sw().csr_identity ? 1u : 0u;
    p.uniform_nb   = p.csr_identity && !gn_sw().uniform_select ? f->uniform_nb : 0u;
    p.run_select   = p.csr_identity && f->run_ok && !p.uniform_nb && !gn_sw().run_select ? 1u : 0u;
    p.max_first    = gn_sw().max_first ? 0u : 1u;
    p.const_nb     = p.uniform_nb && !gn_sw().const_nb ? p.uniform_nb : (p.run_select ? 4u : 0u);
    const bool split = !f->identity && f->d_sl_nbr != nullptr && !gn_sw().split_kernel;
    const bool fast = f->identity;
    // with a filter_matches pre-pass on the stream the fast and the split-bin kernel do not write matches the --rel-filter rule is bound to
    // drop (not for a merging level: there the minimum follows the entries that got in, which only the merge knows)
    const bool predrop = (fast || split) && s->pf_on && !s->pf_merge && s->d_pf_segmin && s->pf_rel_filter >= 0.0 && s->pf_rel_filter < 1.0 &&
                         (uint64_t)hi * f->geom.wpr <= s->pf_segmin_cap && !gn_sw().predrop;
    if (lo == 0)
        s->pf_predrop = predrop;
    if (predrop)
    {
        if (lo == 0)
            GN_HIP(hipMemsetAsync(s->d_pf_pre, 0, sizeof(unsigned long long), s->st));
        GN_HIP(hipMemsetAsync(s->d_pf_segmin + (size_t)lo * f->geom.wpr, 0xFF, (size_t)(hi - lo) * f->geom.wpr * 4, s->st));
        p.pre_mode = s->pf_joint ? 2u : 1u;
        p.pre_rel  = s->pf_rel_filter;
        p.seg_min  = s->d_pf_segmin;
        p.pre_ctr  = s->d_pf_pre;
    }
    if (split)
    {
        // split-bin maps: register counters + byte image for reads with <= 127 minimisers, the rest lands in d_deferred
        GN_HIP(hipMemsetAsync(s->d_ctr + 4, 0, sizeof(unsigned long long), s->st));
        p.work_list_out  = s->d_deferred;
        p.work_count_out = s->d_ctr + 4;
        const uint32_t keep = p.max_blocks;
        p.max_blocks        = (uint32_t)f->n_cu * (f->split_bpc ? f->split_bpc : 1u) * 2u;
        GN_HIP(gn_launch_count_split(p, f->geom, f->ibf.h, s->st));
        p.max_blocks = keep;
        p.work_list  = s->d_deferred;
        p.work_count = s->d_ctr + 4;
    }
    if (fast)
    {
        // register-resident fast path for reads with <= 127 minimisers; the rest lands in d_deferred
        GN_HIP(hipMemsetAsync(s->d_ctr + 4, 0, sizeof(unsigned long long), s->st));
        p.work_list_out  = s->d_deferred;
        p.work_count_out = s->d_ctr + 4;
        if (!gn_sw().on_demand)
        {
            GN_HIP(hipMemsetAsync(s->d_ctr + 72, 0, sizeof(unsigned long long), s->st));
            p.grab = s->d_ctr + 72;
        }
        GN_HIP(gn_launch_count_fast(p, f->geom, f->ibf.h, s->st));
        p.grab       = nullptr; // (the generic kernel that takes the deferred reads hands out its rounds as before)
        p.work_list  = s->d_deferred;
        p.work_count = s->d_ctr + 4;
    }
    if ((fast || split) && s->prev_count_deferred != ~0ull && lo == 0 && hi == s->n_reads && !gn_sw().deferred_grids)
        // the generic kernel only takes what the fast / split kernel deferred: a grid for twice the last batch's list (one block per read)
        p.max_blocks = (uint32_t)std::min<uint64_t>(p.max_blocks, std::max<uint64_t>((uint64_t)f->n_cu, s->prev_count_deferred * 2));
    GN_HIP(gn_launch_count(p, f->geom, f->ibf.h, s->st));
    if (s->long_reads) // reads the kernels above skipped as GN_READ_BIG: 32-bit counters, one workgroup each
    {
        GN_HIP(hipMemsetAsync(s->d_long_count, 0, sizeof(unsigned long long), s->st));
        p.work_list = nullptr;
        GN_HIP(gn_launch_count_long(p, f->ibf.h, s->d_long_list, s->d_long_count, s->d_long_scratch, GN_LONG_BLOCKS, s->st));
    }
    return GN_OK;
}

// whole batch in one go (HIBF, and the re-run after a match-buffer overflow)
static int gn_run_count(gn_stream* s)
{
    GN_HIP(hipMemsetAsync(s->d_ctr, 0, sizeof(unsigned long long), s->st)); // cursor
    if (s->f->is_hibf)
        return gn_hibf_classify(s, s->f, s->st);
    return gn_run_count_range(s, 0, s->n_reads);
}

static int gn_run_group(gn_stream* s)
{
    // group matches by read on the device: exclusive scan of the segment sizes + gather
    s->pf_out    = nullptr; // (set again when a pre-pass has run over this batch)
    s->segmented = false;
    s->compacted = true;
    s->cmp_timed = false;
    if (s->f->is_hibf)
        return GN_OK; // HIBF matches are grouped by gn_hibf_classify
    const size_t nseg = (size_t)s->n_reads * s->f->geom.wpr;
    GN_HIP(hipMemsetAsync(s->d_seg_count + nseg, 0, 4, s->st));
    size_t tmp = s->scan_tmp_bytes;
    GN_HIP(gn_scan_counts(s->d_scan_tmp, tmp, s->d_seg_count, s->d_seg_off, (int)(nseg + 1), s->st));
    // One unit per read: a read's matches are one segment of d_matches already (the fast kernel's and the generic kernel's select emit
    // ascending targets) -- nothing to move.  Several column slices per read: their segments are put behind each other now.
    // (identity maps only: the fast kernel writes a read's matches by ascending target; the reads it defers to the generic kernel
    // -- more than 127 minimisers -- come out of a candidate-driven select in no particular order and are put right below)
    // (... and one count range per batch: the list of deferred reads is per range -- the `chunk=N` test switch keeps the copy per batch)
    s->segmented = s->f->identity && s->f->geom.wpr == 1 && !s->long_reads && !gn_sw().seg_result && gn_sw().chunk == 0;
    s->compacted = !s->segmented;
    if (nseg && s->segmented)
        hipLaunchKernelGGL(gn_seg_order_kernel, dim3(64), dim3(256), 0, s->st, s->d_deferred, s->d_ctr + 4, s->d_matches, s->d_sorted, s->d_seg_begin,
                           s->d_seg_count, s->d_seg_off, s->d_ctr, s->match_cap);
    if (nseg && !s->segmented)
        hipLaunchKernelGGL(gn_gather_kernel, dim3((unsigned)((s->n_reads + 255) / 256)), dim3(256), 0, s->st, s->d_matches,
                           s->d_sorted, s->d_seg_begin, s->d_seg_count, s->d_seg_off, (uint64_t)s->n_reads,
                           (uint32_t)s->f->geom.wpr, s->d_ctr, s->match_cap);
    GN_HIP(hipGetLastError());
    // exact number of matches = scan total (the cursor counts allocated space including chunk holes)
    GN_HIP(hipMemcpyAsync(s->d_ctr + 6, s->d_seg_off + nseg, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->st));
    return GN_OK;
}

// The contiguous (CSR) copy of a segmented result, for the consumers that want one: gn_fetch_batch(matches), gn_stream_device_matches,
// the merge of a partitioned filter, a joint pre-pass.  Queued on the stream, once per batch; callers come after gn_finish (no overflow).
int gn_result_compact(gn_stream* s)
{
    if (!s->segmented || s->compacted || s->f->is_hibf || s->pf_out)
        return GN_OK;
    GN_HIP(hipSetDevice(s->f->device));
    if (!s->ev_cmp[0])
        for (auto& e : s->ev_cmp)
            GN_HIP(hipEventCreate(&e));
    GN_HIP(hipEventRecord(s->ev_cmp[0], s->st));
    if (s->n_reads)
        hipLaunchKernelGGL(gn_gather_kernel, dim3((unsigned)((s->n_reads + 255) / 256)), dim3(256), 0, s->st, s->d_matches, s->d_sorted,
                           s->d_seg_begin, s->d_seg_count, s->d_seg_off, (uint64_t)s->n_reads, 1u, s->d_ctr, s->match_cap);
    GN_HIP(hipGetLastError());
    GN_HIP(hipEventRecord(s->ev_cmp[1], s->st));
    s->compacted = true;
    s->cmp_timed = true;
    return GN_OK;
}

const gn_match* gn_result_matches(gn_stream* s)
{
    return s->pf_out ? s->pf_out : s->d_sorted;
}

static int gn_check_shape(gn_stream* s, uint32_t k, uint32_t w)
{
    if (!s->have_reads)
        return gn_fail(GN_EINVAL, "no reads uploaded on this stream");
    if (k < 1 || k > 32 || w < k)
        return gn_fail(GN_EINVAL, "need 1 <= k <= 32 and w >= k (k=%u w=%u)", k, w);
    if (w - k + 1 > 448 || w > 4096)
        return gn_fail(GN_ERANGE, "window of %u k-mers exceeds the LDS sliding window (max 448)", w - k + 1);
    return GN_OK;
}

// counters + hash slots (#windows per read, exclusive scan) on stream `st`
static int gn_prepare_batch(gn_stream* s, uint32_t w, hipStream_t st)
{
    GN_HIP(hipMemsetAsync(s->d_ctr, 0, GN_NCTR * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(gn_slot_count_kernel, dim3((s->n_reads + 1 + 255) / 256), dim3(256), 0, st, s->d_off1,
                       s->paired ? s->d_off2 : nullptr, s->n_reads, w, s->d_slot_cnt);
    size_t tmp = s->scan_tmp_bytes;
    GN_HIP(hipcub::DeviceScan::ExclusiveSum(s->d_scan_tmp, tmp, s->d_slot_cnt, s->d_slot_off, (int)(s->n_reads + 1), st));
    return GN_OK;
}

// minimiser kernels over reads [lo, hi) on stream `st`
static int gn_run_minimisers_range(gn_stream* s, uint32_t lo, uint32_t hi, hipStream_t st)
{
    gn_filter* f = s->f;
    const uint32_t k = s->k, w = s->w;
    GnMinimiserParams mp{};
    mp.bases        = s->d_bases;
    mp.off1         = s->d_off1;
    mp.off2         = s->paired ? s->d_off2 : nullptr;
    mp.slot_off     = s->d_slot_off;
    mp.n_reads      = hi;
    mp.read_begin   = lo;
    mp.k            = k;
    mp.w            = w;
    mp.hashes       = s->d_hashes;
    mp.n_hashes     = s->d_nh;
    mp.status       = s->d_status;
    mp.total_hashes = s->d_ctr + 8; // 64 shards
    mp.work_hint     = ~0u;
    if (w - k + 1 <= 65)
    {
        // short reads: lane-per-read kernel; longer ones are deferred to the wave-per-read kernel below
        GN_HIP(hipMemsetAsync(s->d_ctr + 5, 0, sizeof(unsigned long long), st));
        mp.lpr_max_len = 640;
        mp.defer_list  = s->d_mdeferred;
        mp.defer_count = s->d_ctr + 5;
        GN_HIP(gn_launch_minimiser_lpr(mp, st));
        mp.work_list  = s->d_mdeferred;
        mp.work_count = s->d_ctr + 5;
        if (s->prev_min_deferred != ~0ull && lo == 0 && hi == s->n_reads && !gn_sw().deferred_grids)
            mp.work_hint = (uint32_t)std::min<uint64_t>(s->prev_min_deferred * 2, 0x7FFFFFFFull); // (twice what the last batch deferred)
    }
    GN_HIP(gn_launch_minimiser(mp, f->n_cu, st));
    return GN_OK;
}

extern "C" int gn_stream_minimisers(gn_stream* s, uint32_t k, uint32_t w)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    int rc = gn_check_shape(s, k, w);
    if (rc)
        return rc;
    GN_HIP(hipSetDevice(s->f->device));
    s->k = k;
    s->w = w;
    GN_HIP(hipEventRecord(s->ev[0], s->st));
    rc = gn_prepare_batch(s, w, s->st);
    if (rc)
        return rc;
    rc = gn_run_minimisers_range(s, 0, s->n_reads, s->st);
    if (rc)
        return rc;
    GN_HIP(hipEventRecord(s->ev[1], s->st));
    s->hashed     = true;
    s->build_distinct = ~0ull;
    s->classified = false;
    return GN_OK;
}

// The minimiser kernels run on a side stream, the count kernels on the main stream.  With $GANON_HIP_CHUNK=<reads>
// the batch is cut into chunks and software-pipelined (minimiser of chunk c+1 || count of chunk c).  Measured on
// MI355X (10 M reads, 8 GiB filter): no gain -- 65.3 ms unchunked vs 65.7-66.7 ms pipelined, the persistent
// HBM-bound count kernel slows down by as much as the hashing it overlaps -- so the default is ONE chunk.
extern "C" int gn_stream_classify(gn_stream* s, uint32_t k, uint32_t w, double rel_cutoff)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (!(rel_cutoff >= 0.0 && rel_cutoff <= 1.0))
        return gn_fail(GN_EINVAL, "rel_cutoff must be within [0,1]");
    int rc = gn_check_shape(s, k, w);
    if (rc)
        return rc;
    gn_filter* f = s->f;
    GN_HIP(hipSetDevice(f->device));
    s->k          = k;
    s->w          = w;
    s->rel_cutoff = rel_cutoff;
    const uint32_t n = s->n_reads;

    uint32_t chunk = gn_sw().chunk;
    if (chunk == 0 || f->is_hibf)
        chunk = n ? n : 1;
    const uint32_t n_chunks = n ? (n + chunk - 1) / chunk : 1;
    if (n_chunks > GN_MAX_CHUNKS)
        chunk = (n + GN_MAX_CHUNKS - 1) / GN_MAX_CHUNKS;
    const uint32_t nc = n ? (n + chunk - 1) / chunk : 1;

    // side stream starts after everything queued on the main stream (uploads, previous batch)
    GN_HIP(hipEventRecord(s->ev_sync, s->st));
    GN_HIP(hipStreamWaitEvent(s->st2, s->ev_sync, 0));
    GN_HIP(hipEventRecord(s->ev[0], s->st2));
    rc = gn_prepare_batch(s, w, s->st2);
    if (rc)
        return rc;
    for (uint32_t c = 0; c < nc; ++c)
    {
        const uint32_t lo = c * chunk, hi = std::min<uint64_t>((uint64_t)lo + chunk, n);
        rc = gn_run_minimisers_range(s, lo, hi, s->st2);
        if (rc)
            return rc;
        GN_HIP(hipEventRecord(s->ev_chunk[c], s->st2));
    }
    GN_HIP(hipEventRecord(s->ev[1], s->st2));
    s->hashed = true;
    s->build_distinct = ~0ull;

    for (uint32_t c = 0; c < nc; ++c)
    {
        const uint32_t lo = c * chunk, hi = std::min<uint64_t>((uint64_t)lo + chunk, n);
        GN_HIP(hipStreamWaitEvent(s->st, s->ev_chunk[c], 0));
        if (c == 0)
            GN_HIP(hipEventRecord(s->ev_count0, s->st)); // first count kernel can start here
        if (f->is_hibf)
            rc = gn_run_count(s);
        else
            rc = gn_run_count_range(s, lo, hi);
        if (rc)
            return rc;
    }
    GN_HIP(hipEventRecord(s->ev[2], s->st));
    rc = gn_run_group(s);
    if (rc)
        return rc;
    rc = gn_run_postfilter(s);
    if (rc)
        return rc;
    GN_HIP(hipMemcpyAsync(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipEventRecord(s->ev[3], s->st));
    s->ctr_copied = true;
    s->n_chunks   = nc;
    s->classified = true;
    s->pf_joint_done = false;
    return GN_OK;
}

extern "C" int gn_submit_batch(gn_stream* s, const uint8_t* bases, uint64_t n_bases, const uint64_t* off1, const uint64_t* off2,
                               uint32_t n_reads, uint32_t k, uint32_t w, double rel_cutoff)
{
    int rc = gn_stream_upload_reads(s, bases, n_bases, off1, off2, n_reads);
    if (rc)
        return rc;
    return gn_stream_classify(s, k, w, rel_cutoff);
}

// The batch resident in `source` (uploaded and hashed there: gn_submit_batch / gn_stream_classify / gn_stream_minimisers) counted
// against THIS stream's filter as well: several filters of a hierarchy level, or the column parts of a wide filter, on one
// device see the same reads with the same (k, w) -- one upload and one pass of the minimiser kernels serve them all.
extern "C" int gn_stream_classify_shared(gn_stream* s, gn_stream* source, double rel_cutoff)
{
    if (!s || !source || s == source)
        return gn_fail(GN_EINVAL, "gn_stream_classify_shared: two different streams are needed");
    if (!(rel_cutoff >= 0.0 && rel_cutoff <= 1.0))
        return gn_fail(GN_EINVAL, "rel_cutoff must be within [0,1]");
    if (s->device != source->device)
        return gn_fail(GN_EINVAL, "gn_stream_classify_shared: the streams are on different devices");
    if (!source->hashed || source->src)
        return gn_fail(GN_EINVAL, "gn_stream_classify_shared: the source stream holds no hashed batch of its own");
    if (source->n_reads > s->max_reads)
        return gn_fail(GN_EINVAL, "batch (%u reads) exceeds the stream capacity (%u)", source->n_reads, s->max_reads);
    gn_filter* f = s->f;
    GN_HIP(hipSetDevice(f->device));
    GN_HIP(hipStreamSynchronize(s->st)); // this stream's previous batch is done with its buffers
    s->src        = source;
    s->v_hashes   = source->d_hashes;
    s->v_slot_off = source->d_slot_off;
    s->v_nh       = source->d_nh;
    s->v_status   = source->d_status;
    s->n_reads    = source->n_reads;
    s->n_bases    = source->n_bases;
    s->paired     = source->paired;
    s->have_reads = true;
    s->hashed     = false; // (no hashes of its own: gn_stream_fetch_hashes belongs to the source)
    s->k          = source->k;
    s->w          = source->w;
    s->rel_cutoff = rel_cutoff;
    s->build_distinct = ~0ull;
    // behind the source's minimiser kernels (its side stream), or behind its main stream when it hashed there
    GN_HIP(hipEventRecord(s->ev_sync, source->st2));
    GN_HIP(hipStreamWaitEvent(s->st, s->ev_sync, 0));
    GN_HIP(hipEventRecord(s->ev_sync, source->st));
    GN_HIP(hipStreamWaitEvent(s->st, s->ev_sync, 0));
    GN_HIP(hipMemsetAsync(s->d_ctr, 0, GN_NCTR * sizeof(unsigned long long), s->st));
    GN_HIP(hipEventRecord(s->ev[0], s->st));
    GN_HIP(hipEventRecord(s->ev[1], s->st));
    GN_HIP(hipEventRecord(s->ev_count0, s->st));
    int rc = f->is_hibf ? gn_run_count(s) : gn_run_count_range(s, 0, s->n_reads);
    if (rc)
        return rc;
    GN_HIP(hipEventRecord(s->ev[2], s->st));
    rc = gn_run_group(s);
    if (rc)
        return rc;
    rc = gn_run_postfilter(s);
    if (rc)
        return rc;
    GN_HIP(hipMemcpyAsync(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipEventRecord(s->ev[3], s->st));
    s->ctr_copied    = true;
    s->n_chunks      = 1;
    s->classified    = true;
    s->pf_joint_done = false;
    return GN_OK;
}

// Waits for the batch; if the device match buffer overflowed, grows it and re-runs count+group.
static int gn_finish(gn_stream* s)
{
    if (!s->classified)
        return gn_fail(GN_EINVAL, "no classified batch on this stream");
    GN_HIP(hipSetDevice(s->f->device));
    for (int attempt = 0; attempt < 4; ++attempt)
    {
        GN_HIP(hipStreamSynchronize(s->st));
        // the counters were copied to pinned memory behind the batch's last kernel; only a joint pre-pass, which runs after that
        // copy was queued, makes a second read-back necessary
        const bool stale = !s->ctr_copied || (s->pf_joint && s->pf_joint_done);
        if (stale)
            GN_HIP(hipMemcpy(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const uint64_t need = s->h_ctr[0];
        if (gn_sw().debug)
            fprintf(stderr, "[gn_finish] attempt %d need %llu cap %llu nh %llu\n", attempt, (unsigned long long)need,
                    (unsigned long long)s->match_cap, (unsigned long long)s->h_ctr[1]);
        if (need <= s->match_cap)
        {
            s->n_matches = s->h_ctr[6]; // exact (the cursor `need` counts allocated space including chunk holes)
            if (s->pf_on && (!s->pf_joint || s->pf_joint_done)) // the batch's result is what the device-side filter_matches pre-pass left
            {
                if (stale)
                    GN_HIP(hipMemcpy(s->h_pf_ctr, s->d_pf_ctr, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                s->n_matches = s->h_pf_ctr[2];
            }
            if (!s->f->is_hibf)
            {
                s->prev_count_deferred = s->h_ctr[4];
                if (!s->src)
                    s->prev_min_deferred = s->h_ctr[5];
            }
            return GN_OK;
        }
        const uint64_t ncap = need + need / 8 + 1024, ocap = s->match_cap;
        hipFree(s->d_matches);
        hipFree(s->d_sorted);
        s->d_matches = s->d_sorted = nullptr;
        if (gn_dmalloc(&s->d_matches, ncap) != hipSuccess || gn_dmalloc(&s->d_sorted, ncap) != hipSuccess)
        {
            // no room for this batch's matches: the stream goes back to the buffers it had (so that a smaller batch can follow)
            (void)hipGetLastError();
            if (s->d_matches)
                hipFree(s->d_matches);
            if (s->d_sorted)
                hipFree(s->d_sorted);
            s->d_matches = s->d_sorted = nullptr;
            s->match_cap  = 0;
            s->classified = false;
            if (gn_dmalloc(&s->d_matches, ocap) == hipSuccess && gn_dmalloc(&s->d_sorted, ocap) == hipSuccess)
                s->match_cap = ocap;
            return gn_fail(GN_ENOMEM, "the batch has %llu matches: 2 x %.1f GB of match records do not fit into device memory next to the filter -- "
                                      "submit fewer reads per batch (or raise the cutoff)",
                           (unsigned long long)need, (double)ncap * sizeof(gn_match) / 1e9);
        }
        s->match_cap = ncap;
        int rc       = gn_run_count(s);
        if (rc)
            return rc;
        rc = gn_run_group(s);
        if (rc)
            return rc;
        rc = gn_run_postfilter(s);
        if (rc)
            return rc;
        GN_HIP(hipMemcpyAsync(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->st));
        s->ctr_copied = true;
    }
    return gn_fail(GN_ENODEV, "match buffer kept overflowing");
}

int gn_finish_batch(gn_stream* s) // (for gn_postfilter.hip)
{
    return gn_finish(s);
}

extern "C" int gn_stream_sync(gn_stream* s)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (s->classified)
        return gn_finish(s);
    GN_HIP(hipSetDevice(s->f->device));
    GN_HIP(hipStreamSynchronize(s->st));
    return GN_OK;
}

__global__ void gn_pick_offsets_kernel(const uint64_t* seg_off, uint32_t wpr, uint32_t n_reads, uint64_t* match_off)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads)
        match_off[r] = seg_off[(size_t)r * wpr];
}

extern "C" int gn_fetch_batch(gn_stream* s, uint32_t* n_hashes, uint8_t* status, uint64_t* match_off, gn_match* matches,
                              uint64_t cap, uint64_t* n_matches)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    int rc = gn_finish(s);
    if (rc)
        return rc;
    if (n_matches)
        *n_matches = s->n_matches;
    const uint32_t n = s->n_reads;
    if (n_hashes && n)
        GN_HIP(hipMemcpyAsync(n_hashes, s->v_nh, (size_t)n * 4, hipMemcpyDeviceToHost, s->st));
    if (status && n)
        GN_HIP(hipMemcpyAsync(status, s->v_status, (size_t)n, hipMemcpyDeviceToHost, s->st));
    if (match_off)
    {
        const uint32_t wpr = s->f->is_hibf ? 1 : s->f->geom.wpr;
        if (s->pf_on)
            GN_HIP(hipMemcpyAsync(match_off, s->d_slot_cnt, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s->st));
        else if (wpr == 1)
            GN_HIP(hipMemcpyAsync(match_off, s->d_seg_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s->st));
        else
        {
            // d_slot_cnt is free after the minimiser kernel: reuse it for the strided pick
            hipLaunchKernelGGL(gn_pick_offsets_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, s->st, s->d_seg_off, wpr, n,
                               s->d_slot_cnt);
            GN_HIP(hipMemcpyAsync(match_off, s->d_slot_cnt, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s->st));
        }
    }
    if (matches)
    {
        if (cap < s->n_matches)
        {
            hipStreamSynchronize(s->st);
            return gn_fail(GN_EOVERFLOW, "match buffer too small: need %llu, have %llu", (unsigned long long)s->n_matches,
                           (unsigned long long)cap);
        }
        if ((rc = gn_result_compact(s)) != GN_OK)
            return rc;
        if (s->n_matches)
            GN_HIP(hipMemcpyAsync(matches, gn_result_matches(s), s->n_matches * sizeof(gn_match), hipMemcpyDeviceToHost, s->st));
    }
    GN_HIP(hipStreamSynchronize(s->st));
    if (status && s->long_reads) // the long kernel classified them (the device keeps GN_READ_BIG: a re-run after a
        for (uint32_t r = 0; r < n; ++r) // match-buffer regrow has to find them again)
            if (status[r] == GN_READ_BIG)
                status[r] = GN_READ_OK;
    return GN_OK;
}

extern "C" int gn_stream_set_long_reads(gn_stream* s, int on)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (!on)
    {
        s->long_reads = false;
        return GN_OK;
    }
    GN_HIP(hipSetDevice(s->device));
    if (s->f->is_hibf) // (the level kernels take the flag: 32-bit sums, GN_READ_BIG reads through the LDS-counter kernel)
    {
        s->long_reads = true;
        return GN_OK;
    }
    if (!s->d_long_list)
    {
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_long_list), ((size_t)s->max_reads + 1) * 4));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_long_count), sizeof(unsigned long long)));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_long_scratch), (size_t)GN_LONG_BLOCKS * ((size_t)s->f->ibf.B + 64) * 4));
    }
    s->long_reads = true;
    return GN_OK;
}

extern "C" int gn_fetch_postfilter(gn_stream* s, uint32_t* max_count, uint64_t* dropped_rel_filter, uint64_t* dropped_fpr_query)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (!s->pf_on)
        return gn_fail(GN_EINVAL, "no post-filter set on this stream");
    int rc = gn_finish(s);
    if (rc)
        return rc;
    if (max_count && s->n_reads)
        GN_HIP(hipMemcpy(max_count, s->d_pf_max, (size_t)s->n_reads * 4, hipMemcpyDeviceToHost));
    if (dropped_rel_filter)
        *dropped_rel_filter = s->h_pf_ctr[0];
    if (dropped_fpr_query)
        *dropped_fpr_query = s->h_pf_ctr[1];
    return GN_OK;
}

extern "C" int gn_stream_device_matches(gn_stream* s, const gn_match** d_matches, uint64_t* n_matches)
{
    if (!s || !d_matches || !n_matches)
        return gn_fail(GN_EINVAL, "null argument");
    int rc = gn_finish(s);
    if (rc)
        return rc;
    const bool was_compact = s->compacted || s->pf_out;
    if ((rc = gn_result_compact(s)) != GN_OK)
        return rc;
    if (!was_compact) // (the caller reads the buffer on a stream of its own: the copy queued just now has to be through)
        GN_HIP(hipStreamSynchronize(s->st));
    *d_matches = gn_result_matches(s);
    *n_matches = s->n_matches;
    return GN_OK;
}

extern "C" int gn_stream_fetch_hashes(gn_stream* s, uint64_t* hash_off, uint64_t* hashes, uint64_t cap, uint64_t* n_total)
{
    if (!s || !hash_off)
        return gn_fail(GN_EINVAL, "null argument");
    if (!s->hashed)
        return gn_fail(GN_EINVAL, "no minimisers computed on this stream");
    GN_HIP(hipSetDevice(s->f->device));
    GN_HIP(hipStreamSynchronize(s->st));
    const uint32_t        n = s->n_reads;
    std::vector<uint64_t> slot(n + 1);
    std::vector<uint32_t> nh(n);
    GN_HIP(hipMemcpy(slot.data(), s->v_slot_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
    if (n)
        GN_HIP(hipMemcpy(nh.data(), s->v_nh, (size_t)n * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t r = 0; r < n; ++r)
    {
        hash_off[r] = total;
        total += nh[r];
    }
    hash_off[n] = total;
    if (n_total)
        *n_total = total;
    if (!hashes)
        return GN_OK;
    if (cap < total)
        return gn_fail(GN_EOVERFLOW, "hash buffer too small: need %llu", (unsigned long long)total);
    std::vector<uint64_t> all(slot[n] ? slot[n] : 1);
    if (slot[n])
        GN_HIP(hipMemcpy(all.data(), s->v_hashes, slot[n] * 8, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n; ++r)
        if (nh[r])
            memcpy(hashes + hash_off[r], all.data() + slot[r], (size_t)nh[r] * 8);
    return GN_OK;
}

int gn_hibf_dense(gn_stream* s, uint32_t rb, uint32_t re, uint16_t* counts); // gn_hibf.hip

extern "C" int gn_stream_dense_counts(gn_stream* s, uint32_t read_begin, uint32_t read_end, uint16_t* counts)
{
    if (!s || !counts)
        return gn_fail(GN_EINVAL, "null argument");
    int rc = gn_finish(s);
    if (rc)
        return rc;
    if (read_begin > read_end || read_end > s->n_reads)
        return gn_fail(GN_EINVAL, "read range out of bounds");
    if (read_begin == read_end)
        return GN_OK;
    gn_filter* f = s->f;
    if (f->is_hibf)
        return gn_hibf_dense(s, read_begin, read_end, counts);
    const size_t nel = (size_t)(read_end - read_begin) * f->ibf.B;
    uint16_t*    dd  = nullptr;
    GN_HIP(gn_dmalloc(&dd, nel));
    if (hipMemsetAsync(dd, 0, nel * 2, s->st) != hipSuccess) // (skipped reads write nothing)
    {
        (void)hipFree(dd);
        return gn_fail(GN_ENODEV, "gn_stream_dense_counts: clearing the count matrix failed");
    }
    // re-run the count kernel with the dense tap on (matches of this run are discarded)
    unsigned long long saved[GN_NCTR];
    memcpy(saved, s->h_ctr, sizeof(saved));
    GnCountParams p{};
    p.rows = f->ibf.d_rows; p.S = f->ibf.S; p.W = (uint32_t)f->ibf.W; p.B = (uint32_t)f->ibf.B; p.shift = f->ibf.shift;
    p.tgt_off = f->identity ? nullptr : f->d_tgt_off; p.tgt_bins = f->d_tgt_bins; p.tgt_lds = f->d_tgt_lds; p.tgt_rec = f->d_tgt_rec; p.n_targets = f->n_targets;
    p.hashes = s->v_hashes; p.slot_off = s->v_slot_off; p.n_hashes = s->v_nh; p.status = s->v_status;
    p.n_reads = s->n_reads; p.rel_cutoff = s->rel_cutoff; p.wpr = f->geom.wpr; p.gp_log2 = f->geom.gp_log2;
    p.slice_dwords = f->geom.slice_dwords;
    p.matches = s->d_matches; p.match_cap = 0; // no writes
    unsigned long long* dctr = nullptr;
    hipError_t e = gn_dmalloc(&dctr, 1);
    if (e != hipSuccess) { hipFree(dd); return gn_fail(GN_ENOMEM, "alloc failed"); }
    hipMemsetAsync(dctr, 0, 8, s->st);
    uint64_t* segb = nullptr; uint32_t* segc = nullptr;
    gn_dmalloc(&segb, (size_t)s->n_reads * f->geom.wpr);
    gn_dmalloc(&segc, (size_t)s->n_reads * f->geom.wpr);
    p.cursor = dctr; p.seg_begin = segb; p.seg_count = segc;
    p.dense = dd; p.dense_begin = read_begin; p.dense_end = read_end;
    p.max_blocks = (uint32_t)f->n_cu * 16u;
    e = gn_launch_count(p, f->geom, f->ibf.h, s->st);
    hipError_t e2 = hipStreamSynchronize(s->st);
    if (e == hipSuccess && e2 == hipSuccess)
        e = hipMemcpy(counts, dd, nel * 2, hipMemcpyDeviceToHost);
    hipFree(dd); hipFree(dctr); hipFree(segb); hipFree(segc);
    memcpy(s->h_ctr, saved, sizeof(saved));
    if (e != hipSuccess || e2 != hipSuccess)
        return gn_fail(GN_ENODEV, "dense count tap failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return GN_OK;
}

extern "C" int gn_stream_timings(gn_stream* s, gn_timings* t)
{
    if (!s || !t)
        return gn_fail(GN_EINVAL, "null argument");
    int rc = gn_finish(s);
    if (rc)
        return rc;
    gn_timings tm{};
    hipEventElapsedTime(&tm.ms_minimiser, s->ev[0], s->ev[1]);
    hipEventElapsedTime(&tm.ms_count, s->ev_count0, s->ev[2]); // first count kernel start -> last count kernel end
    hipEventElapsedTime(&tm.ms_total, s->ev[0], s->ev[3]);
    tm.n_hashes = 0;
    for (int i = 8; i < 8 + 64; ++i) // 64 shards (the minimiser kernels of the batch ran on the stream the reads were uploaded to)
        tm.n_hashes += (s->src ? s->src : s)->h_ctr[i];
    tm.n_matches = s->n_matches;
    tm.n_count_launches = s->n_chunks;
    if (s->f->is_hibf)
        tm.algo_bytes = s->h_ctr[2];
    else
        tm.algo_bytes = tm.n_hashes * (uint64_t)s->f->ibf.h * s->f->ibf.W * 8ull;
    tm.fetched_bytes = tm.algo_bytes - (s->f->is_hibf ? 0ull : (uint64_t)s->h_ctr[7]);
    if (s->cmp_timed && hipEventSynchronize(s->ev_cmp[1]) == hipSuccess)
        hipEventElapsedTime(&tm.ms_compact, s->ev_cmp[0], s->ev_cmp[1]);
    *t = tm;
    return GN_OK;
}
