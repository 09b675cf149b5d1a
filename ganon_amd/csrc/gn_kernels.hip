// gn_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the read-classification hot path.
//
//   gn_minimiser_kernel   canonical (k,w)-minimiser hashes per read: one wavefront per read, 64 windows
//                         per tile, k-mer values staged in LDS (sliding window), rightmost-min scan per
//                         lane, emission resolved with ballots + a scalar expiry chain.
//                         == seqan3::views::minimiser_hash at /root/reference/src/ganon-classify/GanonClassify.cpp:647-650,693-700
//   gn_ibf_count_kernel   IBF bulk_count + select_matches: per hash, h row addresses -> coalesced 16 B/lane
//                         row loads (a 512 B row = 32 lanes x dwordx4), AND-reduced in registers, counted
//                         with bit-sliced SWAR nibble/byte counters, flushed to LDS, summed per target,
//                         capped, compared with the per-read cutoff and compacted to sparse matches.
//                         == counting_agent::bulk_count (GanonClassify.cpp:514) + :516-540
//   gn_emplace_kernel     IBF emplace (src/ganon-build/GanonBuild.cpp:694) as an atomic OR scatter.
//
// Pure integer / bit-vector work: HBM-bound random row gathers, no MFMA anywhere.
#include "gn_internal.h"
#include <cstdlib>

#define GN_WAVE 64
// candidate-driven select of the generic kernel: reads with more candidate bins per wave than this keep the scan over
// every target (GN_CAND_NBIG is in gn_internal.h)
#define GN_CAND_LIMIT 128u
// fast kernel, early-exit instances: with at most this many bins left in the race the remaining hashes only fetch the
// words that hold those bins
#ifndef GN_NARROW_MAX
#define GN_NARROW_MAX 4u
#endif
#define GN_MATCH_CHUNK 256u      // first wave-private chunk of the match buffer ...
#ifndef GN_MATCH_CHUNK_MAX
#define GN_MATCH_CHUNK_MAX 8192u
#endif
// ... doubling with every further request of the wave (see the fast kernel's epilogue)
#define GN_STAGE_CAP 128u // per-wave LDS staging of (target, count) hits in the generic select pass

// ------------------------------------------------------------------------------------------------
// dna4 rank table (seqan3::dna4 char_to_rank): ACGT(U) exact, IUPAC collapse, else A
// ------------------------------------------------------------------------------------------------
struct GnRankLut
{
    uint8_t t[256];
    constexpr GnRankLut() : t{}
    {
        for (int i = 0; i < 256; ++i)
            t[i] = 0;
        t['C'] = t['c'] = 1;
        t['G'] = t['g'] = 2;
        t['T'] = t['t'] = t['U'] = t['u'] = 3;
        t['Y'] = t['y'] = 1;
        t['S'] = t['s'] = 1;
        t['K'] = t['k'] = 2;
        t['B'] = t['b'] = 1;
    }
};
__constant__ GnRankLut GN_RANK_LUT = GnRankLut();

// seqan3::interleaved_bloom_filter hash seeds
__constant__ uint64_t GN_IBF_SEEDS[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST;   // include/ganon_ibf_hash.h

// LDS written by some lanes of a wave and read by other lanes of the SAME wave: the LDS unit executes a
// wave's DS instructions in issue order, so only compiler reordering has to be prevented.
__device__ __forceinline__ void gn_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t gn_readlane64(uint64_t v, uint32_t lane_uniform)
{
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, (int)lane_uniform);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)lane_uniform);
    return ((uint64_t)hi << 32) | lo;
}

// hash_and_fit without the final "* technical_bins": row index of hash function i
__device__ __forceinline__ uint32_t gn_ibf_row(uint64_t v, uint32_t i, uint32_t shift, uint64_t S)
{
    uint64_t x = v * GN_IBF_SEEDS[i];
    x ^= x >> shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint32_t)__umul64hi(x, S);
}

// ================================================================================================
// minimiser kernel
// ================================================================================================
// One mate sequence.  rk: LDS bytes [64 + w - 1], vv: LDS u64 [64 + K - 1].  Returns the number of emitted
// minimisers (uniform).  Restates SURVEY App. A.1 / App. D in parallel form:
//   W_j, R_j    = value / absolute position of the RIGHTMOST minimum of window j
//   seed(j)     = j == 0 || v[j+K-1] < W_{j-1}          (a strictly smaller value enters)
//   emission e  -> remembered position R_e; the next emission is min(next seed, R_e + 1 (expiry))
// which is exactly the state machine of seqan3's minimiser view (validated against the oracle).
// bit b of x -> bit 2b (Morton spread of a 32-bit word into 64 bits)
__device__ __forceinline__ uint32_t gn_spread16(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__device__ __forceinline__ uint64_t gn_spread32(uint32_t x)
{
    return ((uint64_t)gn_spread16(x >> 16) << 32) | gn_spread16(x & 0xFFFFu);
}
// bits [i, i+32) of the 128-bit string nxt:cur (cur, nxt wave-uniform; i = lane-dependent, 0..63)
__device__ __forceinline__ uint32_t gn_window32(uint64_t cur, uint64_t nxt, uint32_t i)
{
    const uint32_t lo = i < 32 ? (uint32_t)cur : (uint32_t)(cur >> 32);
    const uint32_t hi = i < 32 ? (uint32_t)(cur >> 32) : (uint32_t)nxt;
    return __builtin_amdgcn_alignbit(hi, lo, i & 31u);
}

// Fast minimiser path for windows of at most 65 k-mers (w - k <= 64; ganon's defaults are 12 and 16).
//   * bases are read coalesced, 64 per step; their dna4 ranks become two wave-uniform 64-bit bit planes (ballots)
//   * lane i of a chunk builds its k-mer from the planes: forward hash = Morton interleave of the bit-reversed
//     k-bit windows, reverse-complement hash = complement of the Morton interleave of the plain windows
//     (rc digit at weight 4^m is 3 - rank[p+m]); v = min(f ^ seed, rc ^ seed) goes to a two-chunk LDS ring
//   * the sliding-window / emission logic is the same as in the generic version below
__device__ uint32_t gn_mate_minimisers_fast(const uint8_t* __restrict__ seq, uint32_t L, uint32_t k, uint32_t w, uint64_t seed,
                                            uint64_t* vv, uint64_t* __restrict__ out, int lane)
{
    const uint32_t K      = w - k + 1;
    const uint32_t M      = L - k + 1;
    const uint32_t nwin   = L - w + 1;
    const uint32_t nchunk = (M + GN_WAVE - 1) / GN_WAVE;
    const uint32_t kmask  = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const uint64_t dmask  = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);

    auto load_planes = [&](uint32_t c, uint64_t& lo, uint64_t& hi) {
        const uint32_t pos = c * GN_WAVE + lane;
        uint32_t       r   = 0;
        if (pos < L)
            r = GN_RANK_LUT.t[seq[pos]];
        lo = __ballot(r & 1u);
        hi = __ballot(r & 2u);
    };
    auto compute_chunk = [&](uint32_t c, uint64_t lo_c, uint64_t hi_c, uint64_t lo_n, uint64_t hi_n) {
        const uint32_t xl = gn_window32(lo_c, lo_n, lane) & kmask;
        const uint32_t xh = gn_window32(hi_c, hi_n, lane) & kmask;
        const uint64_t P  = (gn_spread32(xh) << 1) | gn_spread32(xl);                       // sum d_m 4^m
        const uint32_t rl = __brev(xl) >> (32 - k), rh = __brev(xh) >> (32 - k);
        const uint64_t f  = (gn_spread32(rh) << 1) | gn_spread32(rl);                       // sum d_j 4^(k-1-j)
        const uint64_t a = f ^ seed, b = ((~P) & dmask) ^ seed;
        vv[(c & 1u) * GN_WAVE + lane] = a < b ? a : b;
    };

    uint64_t lo0, hi0, lo1, hi1;
    load_planes(0, lo0, hi0);
    load_planes(1, lo1, hi1);
    compute_chunk(0, lo0, hi0, lo1, hi1);

    uint32_t nout   = 0;
    uint32_t expiry = 0xFFFFFFFFu;
    uint64_t carryW = 0;
    uint32_t tile   = 0;
    for (uint32_t T0 = 0; T0 < nwin; T0 += GN_WAVE, ++tile)
    {
        // planes tile+1 are in (lo1, hi1); chunk tile+1 needs planes tile+1 and tile+2
        uint64_t lo2, hi2;
        load_planes(tile + 2, lo2, hi2);
        if (tile + 1 < nchunk)
            compute_chunk(tile + 1, lo1, hi1, lo2, hi2);
        lo1 = lo2;
        hi1 = hi2;
        gn_wave_lds_sync();

        const uint32_t j     = T0 + lane;
        const bool     valid = j < nwin;
        uint64_t       m     = ~0ULL, last = ~0ULL;
        uint32_t       pos   = 0;
        if (valid)
        {
            m    = vv[j & 127u];
            last = m;
            for (uint32_t i = 1; i < K; ++i)
            {
                const uint64_t x = vv[(j + i) & 127u];
                if (x <= m)
                {
                    m   = x;
                    pos = i;
                }
                last = x;
            }
        }
        const uint32_t R     = j + pos;
        uint64_t       prevW = __shfl_up((unsigned long long)m, 1);
        if (lane == 0)
            prevW = carryW;
        const bool enters = valid && j > 0 && last < prevW;
        uint64_t   seeds  = __ballot(enters);
        if (T0 == 0)
            seeds |= 1ULL;
        const uint32_t tile_n = min((uint32_t)GN_WAVE, nwin - T0);

        uint64_t EM = 0;
        uint32_t p  = 0;
        while (true)
        {
            uint32_t s = GN_WAVE;
            if (p < GN_WAVE)
            {
                const uint64_t rem = seeds >> p;
                if (rem)
                    s = p + (uint32_t)__builtin_ctzll(rem);
            }
            const uint32_t e2 = (expiry - T0 < (uint32_t)GN_WAVE) ? expiry - T0 : (uint32_t)GN_WAVE;
            uint32_t       e  = s < e2 ? s : e2;
            e                 = __builtin_amdgcn_readfirstlane(e);
            if (e >= tile_n)
                break;
            EM |= 1ULL << e;
            expiry = (uint32_t)__builtin_amdgcn_readlane((int)R, (int)e) + 1u;
            p      = e + 1;
        }
        if ((EM >> lane) & 1ULL)
            out[nout + __popcll(EM & ((1ULL << lane) - 1ULL))] = m;
        nout += __popcll(EM);
        carryW = gn_readlane64(m, GN_WAVE - 1);
        gn_wave_lds_sync(); // the next tile overwrites the older ring slot
    }
    return nout;
}

// Generic version (any window up to 448 k-mers): ranks staged as bytes in LDS, k-step rolling hash per lane.
__device__ uint32_t gn_mate_minimisers(const uint8_t* __restrict__ seq, uint32_t L, uint32_t k, uint32_t w, uint64_t seed,
                                       uint8_t* rk, uint64_t* vv, uint64_t* __restrict__ out, int lane)
{
    const uint32_t K    = w - k + 1;
    const uint32_t M    = L - k + 1;
    const uint32_t nwin = L - w + 1;
    uint32_t       nout = 0;
    uint32_t       expiry = 0xFFFFFFFFu; // absolute window index at which the remembered minimiser leaves
    uint64_t       carryW = 0;

    for (uint32_t T0 = 0; T0 < nwin; T0 += GN_WAVE)
    {
        // stage dna4 ranks of bases [T0, T0 + 64 + w - 1)
        const uint32_t nb = min(GN_WAVE + w - 1, L - T0);
        for (uint32_t i = lane; i < nb; i += GN_WAVE)
            rk[i] = GN_RANK_LUT.t[seq[T0 + i]];
        gn_wave_lds_sync();

        // canonical k-mer values v[T0 .. T0 + 64 + K - 1)
        const uint32_t nv = min(GN_WAVE + K - 1, M - T0);
        for (uint32_t i = lane; i < nv; i += GN_WAVE)
        {
            uint64_t f = 0, r = 0;
            for (uint32_t j = 0; j < k; ++j)
            {
                const uint64_t b = rk[i + j];
                f                = (f << 2) | b;
                r                = (r >> 2) | ((3ULL - b) << (2 * (k - 1)));
            }
            const uint64_t a = f ^ seed, c = r ^ seed;
            vv[i]            = a < c ? a : c;
        }
        gn_wave_lds_sync();

        // sliding window: rightmost minimum of v[j .. j+K-1]
        const uint32_t j     = T0 + lane;
        const bool     valid = j < nwin;
        uint64_t       m     = ~0ULL;
        uint32_t       pos   = 0;
        if (valid)
        {
            m = vv[lane];
            for (uint32_t i = 1; i < K; ++i)
            {
                const uint64_t x = vv[lane + i];
                if (x <= m)
                {
                    m   = x;
                    pos = i;
                }
            }
        }
        const uint32_t R = j + pos;
        uint64_t prevW = __shfl_up((unsigned long long)m, 1);
        if (lane == 0)
            prevW = carryW;
        const bool enters = valid && j > 0 && vv[lane + K - 1] < prevW;
        uint64_t   seeds  = __ballot(enters);
        if (T0 == 0)
            seeds |= 1ULL;
        const uint32_t tile_n = min((uint32_t)GN_WAVE, nwin - T0);

        // expiry chain (wave-uniform scalar loop; one iteration per emitted minimiser)
        uint64_t EM = 0;
        uint32_t p  = 0;
        while (true)
        {
            uint32_t s = GN_WAVE;
            if (p < GN_WAVE)
            {
                const uint64_t rem = seeds >> p;
                if (rem)
                    s = p + (uint32_t)__builtin_ctzll(rem);
            }
            const uint32_t e2 = (expiry - T0 < (uint32_t)GN_WAVE) ? expiry - T0 : (uint32_t)GN_WAVE; // expiry >= T0 + p
            uint32_t       e  = s < e2 ? s : e2;
            e                 = __builtin_amdgcn_readfirstlane(e);
            if (e >= tile_n)
                break;
            EM |= 1ULL << e;
            expiry = (uint32_t)__builtin_amdgcn_readlane((int)R, (int)e) + 1u;
            p      = e + 1;
        }

        if ((EM >> lane) & 1ULL)
            out[nout + __popcll(EM & ((1ULL << lane) - 1ULL))] = m;
        nout += __popcll(EM);
        carryW = gn_readlane64(m, GN_WAVE - 1);
        gn_wave_lds_sync(); // rk / vv are overwritten by the next tile
    }
    return nout;
}

__global__ __launch_bounds__(256) void gn_minimiser_kernel(GnMinimiserParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gn_smem[];
    const int      lane      = threadIdx.x & (GN_WAVE - 1);
    const int      wave      = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform -> SGPRs
    const uint32_t K         = p.w - p.k + 1;
    const bool     fastp     = K <= 65; // two-chunk LDS ring + bit-plane k-mers
    const uint32_t vv_bytes  = fastp ? 128u * 8u : (GN_WAVE + K) * 8;
    const uint32_t rk_bytes  = fastp ? 0u : ((GN_WAVE + p.w + 15) & ~15u);
    uint8_t*       base      = gn_smem + (size_t)wave * (vv_bytes + rk_bytes);
    uint64_t*      vv        = reinterpret_cast<uint64_t*>(base);
    uint8_t*       rk        = base + vv_bytes;
    const uint64_t seed      = 0x8F3F73B5CF1C9ADEULL >> (64u - 2u * p.k); // adjust_seed.hpp:33-37
    const uint32_t waves_all = gridDim.x * (blockDim.x >> 6);
    unsigned long long my_total = 0;

    const uint32_t n_work = p.work_list ? (uint32_t)*p.work_count : p.n_reads - p.read_begin;
    for (uint32_t widx = blockIdx.x * (blockDim.x >> 6) + wave; widx < n_work; widx += waves_all)
    {
        const uint32_t r    = p.work_list ? p.work_list[widx] : p.read_begin + widx;
        const uint64_t b1   = p.off1[r];
        const uint64_t len1 = p.off1[r + 1] - b1;
        uint64_t       b2 = 0, len2 = 0;
        if (p.off2)
        {
            b2   = p.off2[r];
            len2 = p.off2[r + 1] - b2;
        }
        uint32_t n  = 0;
        uint8_t  st = GN_READ_OK;
        if (len1 < p.w) // GanonClassify.cpp:690,743-747 -- regardless of the mate
        {
            st = GN_READ_SMALL;
        }
        else
        {
            uint64_t* out = p.hashes + p.slot_off[r];
            n = fastp ? gn_mate_minimisers_fast(p.bases + b1, (uint32_t)len1, p.k, p.w, seed, vv, out, lane)
                      : gn_mate_minimisers(p.bases + b1, (uint32_t)len1, p.k, p.w, seed, rk, vv, out, lane);
            if (len2 >= p.w) // :695-700 mate hashes appended
                n += fastp ? gn_mate_minimisers_fast(p.bases + b2, (uint32_t)len2, p.k, p.w, seed, vv, out + n, lane)
                           : gn_mate_minimisers(p.bases + b2, (uint32_t)len2, p.k, p.w, seed, rk, vv, out + n, lane);
            if (n > 65535u) // :674,706 TIntCount = uint16_t
                st = GN_READ_BIG;
            else
                my_total += n;
        }
        if (lane == 0)
        {
            p.n_hashes[r] = n;
            p.status[r]   = st;
        }
    }
    if (lane == 0 && my_total)
        atomicAdd(p.total_hashes + (blockIdx.x & 63u), my_total);
}

hipError_t gn_launch_minimiser(const GnMinimiserParams& p, int n_cu, hipStream_t st)
{
    if (p.n_reads <= p.read_begin)
        return hipSuccess;
    const uint32_t K        = p.w - p.k + 1;
    const bool     fastp    = K <= 65;
    const uint32_t per_wave = fastp ? 128u * 8u : (GN_WAVE + K) * 8 + ((GN_WAVE + p.w + 15) & ~15u);
    const size_t   lds      = (size_t)per_wave * 4;
    uint32_t       blocks   = (p.n_reads - p.read_begin + 3) / 4;
    uint32_t       cap      = (uint32_t)n_cu * 16;
    if (p.work_list && p.work_hint != ~0u) // a deferred list that is usually empty: a grid for about as many reads as the last batch left
        cap = std::min<uint32_t>(cap, std::max<uint32_t>((uint32_t)n_cu, p.work_hint / 2 + 1));
    if (blocks > cap)
        blocks = cap;
    hipLaunchKernelGGL(gn_minimiser_kernel, dim3(blocks), dim3(256), lds, st, p);
    return hipGetLastError();
}

// ================================================================================================
// IBF count + select kernel
// ================================================================================================
// Geometry.  A lane owns LW consecutive 64-bit words (LW*64 bins) of the row; Gp lanes (power of two)
// cover one wave's share of the row, so H = 64/Gp hashes are processed per wave iteration (for the
// 512-byte rows of a 4096-bin IBF: LW = 2 -> 16 B per lane, Gp = 32, H = 2).  Rows wider than 64*LW words
// are split over `wpr` cooperating waves (column slices).
//
// Counting.  The AND-ed 32-bit mask words are added into bit-sliced SWAR counters: 4 nibble registers
// per mask dword (bits j, j+4, ...), flushed every 15 iterations (and at the end) into 16-bit LDS
// counters with ds_add_u32 on u16 pairs.  LDS layout of a slice: dword(q, gl) at q*(Gp+1)+gl where gl is
// the lane in its group and q = 16*d + t/2 indexes the lane's u16 pairs (d mask dword, t bit) --
// bank-conflict free for the flush (fixed q, consecutive gl) and for the select scan (fixed gl,
// consecutive q; stride Gp+1 is odd).
bool gn_count_geometry(uint64_t W, uint32_t hash_funs, GnCountGeometry* g, const char** why)
{
    if (hash_funs < 1 || hash_funs > 5)
    {
        *why = "hash_funs must be 1..5";
        return false;
    }
    if (W == 0)
    {
        *why = "empty filter";
        return false;
    }
    // 16-byte lanes where rows have an even number of words -- except W == 64, where 8-byte lanes make a row exactly
    // one wave: no hash groups to add up, fewer registers (one more wave per SIMD) and a shuffle-free early-exit
    // check (measured on the 4096-bin headline shape: same speed without early exit, 8 % faster with it)
    g->lw                 = (W % 2 == 0 && W != 64) ? 2u : 1u;
    const uint64_t per_wv = 64ull * g->lw; // words per wave slice
    // waves per read = column slices of the row, exactly (rounding up to a power of two left 3 of 8 waves -- and
    // their LDS -- idle on a 640-word row)
    const uint64_t wpr    = (W + per_wv - 1) / per_wv;
    const uint32_t wpr2   = (uint32_t)wpr;
    if (wpr > 16)
    {
        *why = "rows wider than 16 wave slices (bins > 131072 for even bin_words) are not supported by the flat "
               "count kernel yet; partition the filter by bin range";
        return false;
    }
    g->wpr           = wpr2;
    uint64_t lanes   = (W < per_wv ? W : per_wv) / g->lw; // lanes needed in a wave (exact: W even if lw==2)
    if ((W < per_wv ? W : per_wv) % g->lw)
        ++lanes;
    uint32_t gp_log2 = 0;
    while ((1u << gp_log2) < lanes)
        ++gp_log2;
    g->gp_log2       = gp_log2;
    g->rpb           = wpr2 >= 4 ? 1u : 4u / wpr2; // reads per block: about four waves, whole reads only
    g->block         = g->rpb * wpr2 * 64;
    g->slice_dwords  = 32 * g->lw * ((1u << gp_log2) + 1);
    const size_t cnt = (size_t)g->rpb * wpr2 * g->slice_dwords * 4;
    const size_t tab = (size_t)(g->block / 64) * 64 * 8 * 4; // row table: 64 hashes x (up to 8 padded) u32
    const size_t stg = (size_t)(g->block / 64) * 2 * GN_STAGE_CAP * 4; // match staging
    g->lds_bytes     = cnt + tab + stg;
    g->candcnt_off   = g->lds_bytes / 4; // one dword per wave: candidate bins found by the select's prefilter
    g->lds_bytes += (size_t)(g->block / 64) * 4;
    g->nbtab_off     = 0; // (the bins-per-target bytes live in registers)
    if (g->lds_bytes > 160 * 1024)
    {
        *why = "per-read count vector does not fit LDS";
        return false;
    }
    return true;
}

uint32_t gn_count_lds_index(const GnCountGeometry& g, uint32_t b)
{
    // mirrors bin_count() in gn_ibf_count_kernel
    const uint32_t LW   = g.lw;
    const uint32_t Gp   = 1u << g.gp_log2;
    const uint32_t word = b >> 6;
    const uint32_t sl   = word / (64 * LW);
    const uint32_t wl   = word - sl * 64 * LW;
    const uint32_t gg   = wl / LW;
    const uint32_t tp   = (wl - gg * LW) * 64 + (b & 63);
    const uint32_t d = tp >> 5, t = tp & 31;
    const uint32_t q = 16 * d + (t >> 1), half = t & 1;
    return (uint32_t)(((size_t)sl * g.slice_dwords + q * (Gp + 1) + gg) * 2 + half);
}

template <int HF, int LW>
struct GnRowRegs
{
    uint32_t m[HF][2 * LW];
};

template <int HF, int LW, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT > 512 ? 4 : (LW == 2 || MAXT > 256 ? 2 : 3)))) void gn_ibf_count_kernel(GnCountParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gn_lds[];
    constexpr int ND  = 2 * LW;      // mask dwords per lane
    constexpr int HFP = HF <= 4 ? 4 : 8; // padded row-table stride (u32)

    const int      lane   = threadIdx.x & (GN_WAVE - 1);
    const int      wave   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform -> SGPRs
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wpr    = p.wpr;
    const uint32_t rpb    = nwaves / wpr;
    const uint32_t rslot  = wave / wpr;
    const uint32_t slice  = wave % wpr;
    const uint32_t Gp     = 1u << p.gp_log2;
    const uint32_t H      = GN_WAVE >> p.gp_log2;
    const uint32_t gl     = lane & (Gp - 1);
    const uint32_t hsub   = lane >> p.gp_log2;

    uint32_t* cnt_read = gn_lds + (size_t)rslot * wpr * p.slice_dwords; // all slices of my read
    uint32_t* cnt      = cnt_read + (size_t)slice * p.slice_dwords;     // my slice
    uint32_t* rowtab   = gn_lds + (size_t)rpb * wpr * p.slice_dwords + (size_t)wave * 64 * 8;
    uint32_t* stage    = gn_lds + (size_t)rpb * wpr * p.slice_dwords + (size_t)nwaves * 64 * 8 + (size_t)wave * 2 * GN_STAGE_CAP;

    // candidate-driven select: a persistent wave always works on the same column slice, so the bins-per-target bytes
    // of the bins a lane counts are the same for every read -- they stay in 8*ND registers for the whole kernel
    const bool      cand_ok = p.bin_nb2 != nullptr && p.tgt_off != nullptr;
    const uint32_t* nbt_g   = p.bin_nb2 + (size_t)slice * (p.slice_dwords / 2);
    uint32_t*       candcnt = gn_lds + p.candcnt_off;
    uint32_t        nbreg[8 * ND];
#pragma unroll
    for (int j = 0; j < 8 * ND; ++j)
        nbreg[j] = cand_ok ? nbt_g[j * (Gp + 1) + gl] : 0u;

    // Work items: either every read of the batch (work_list == nullptr) or the reads the fast kernel deferred.
    // Blocks stride over rounds of rpb reads; the trip count is block-uniform, so __syncthreads() is safe.
    const uint32_t n_work = p.work_list ? (uint32_t)*p.work_count : p.n_reads - p.read_begin;
    unsigned long long chunk_base = 0; // wave-private slice of the match buffer
    uint32_t           chunk_left = 0, chunk_size = GN_MATCH_CHUNK;
    for (uint32_t round0 = blockIdx.x * rpb; round0 < n_work; round0 += gridDim.x * rpb)
    {
    const uint32_t widx = round0 + rslot;
    const uint32_t read = widx < n_work ? (p.work_list ? p.work_list[widx] : p.read_begin + widx) : 0xFFFFFFFFu;
    uint32_t       n    = 0;
    if (read < p.n_reads && p.status[read] == GN_READ_OK)
        n = p.n_hashes[read];

    // zero my slice
    for (uint32_t i = lane; i < p.slice_dwords; i += GN_WAVE)
        cnt[i] = 0;

    const uint32_t wi      = slice * 64 * LW + gl * LW; // first word of this lane in the row
    const bool     col_act = wi < p.W;
    const uint64_t* hs     = p.hashes + (n ? p.slot_off[read] : 0);

    uint32_t nib[ND][4];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            nib[d][j] = 0;
    uint32_t acc_n = 0; // iterations since the last flush (wave-uniform)

    // nibble counters -> 16-bit LDS counters.  nib[d][j] nibble i counts bit t = 4i + j of mask dword d; the u16
    // pair (t, t+1), t = 4i + 2jp, lives in LDS dword q = 16d + 2i + jp.  Every register index is a
    // compile-time constant (macro expansion) so nothing is demoted to scratch.
#define GN_FLUSH_PAIR(d, jp, src_b, src_a, y, i)                                                                       \
    atomicAdd(&cnt[(16 * (d) + 2 * (i) + (jp)) * (Gp + 1) + gl],                                                       \
              __builtin_amdgcn_perm(src_b, src_a, 0x0C000C00u | ((4u + (y)) << 16) | (uint32_t)(y)));
#define GN_FLUSH_DJ(d, jp)                                                                                             \
    {                                                                                                                  \
        const uint32_t a0_ = nib[d][2 * (jp)] & 0x0F0F0F0Fu, a1_ = (nib[d][2 * (jp)] >> 4) & 0x0F0F0F0Fu;              \
        const uint32_t b0_ = nib[d][2 * (jp) + 1] & 0x0F0F0F0Fu, b1_ = (nib[d][2 * (jp) + 1] >> 4) & 0x0F0F0F0Fu;      \
        GN_FLUSH_PAIR(d, jp, b0_, a0_, 0, 0)                                                                           \
        GN_FLUSH_PAIR(d, jp, b1_, a1_, 0, 1)                                                                           \
        GN_FLUSH_PAIR(d, jp, b0_, a0_, 1, 2)                                                                           \
        GN_FLUSH_PAIR(d, jp, b1_, a1_, 1, 3)                                                                           \
        GN_FLUSH_PAIR(d, jp, b0_, a0_, 2, 4)                                                                           \
        GN_FLUSH_PAIR(d, jp, b1_, a1_, 2, 5)                                                                           \
        GN_FLUSH_PAIR(d, jp, b0_, a0_, 3, 6)                                                                           \
        GN_FLUSH_PAIR(d, jp, b1_, a1_, 3, 7)                                                                           \
        nib[d][2 * (jp)]     = 0;                                                                                      \
        nib[d][2 * (jp) + 1] = 0;                                                                                      \
    }
    auto flush_nibbles = [&]() {
        if (col_act)
        {
            GN_FLUSH_DJ(0, 0)
            GN_FLUSH_DJ(0, 1)
            GN_FLUSH_DJ(1, 0)
            GN_FLUSH_DJ(1, 1)
            if constexpr (ND == 4)
            {
                GN_FLUSH_DJ(2, 0)
                GN_FLUSH_DJ(2, 1)
                GN_FLUSH_DJ(3, 0)
                GN_FLUSH_DJ(3, 1)
            }
        }
    };

    // ---- hashes in chunks of 64: row table in LDS, then H hashes per wave iteration ----
    for (uint32_t c0 = 0; c0 < n; c0 += 64)
    {
        const uint32_t mch = min(64u, n - c0);
        gn_wave_lds_sync();
        for (uint32_t idx = lane; idx < mch * HF; idx += GN_WAVE)
        {
            const uint32_t q = idx / HF, i = idx - q * HF;
            rowtab[q * HFP + i] = gn_ibf_row(hs[c0 + q], i, p.shift, p.S);
        }
        gn_wave_lds_sync();

        const uint32_t iters = (mch + H - 1) / H;

        // every lane issues its row loads unconditionally (see the fast kernel): lanes without a column or past the
        // chunk's last hash read a valid word and are masked when the rows are consumed
        const uint32_t wi_ld = col_act ? wi : 0u; // (word 0 exists in every row)
        auto issue = [&](uint32_t it, GnRowRegs<HF, LW>& R) {
            uint32_t q = it * H + hsub;
            q          = q < mch ? q : mch - 1;
            uint32_t row[HF];
#pragma unroll
            for (int i = 0; i < HF; ++i)
                row[i] = rowtab[q * HFP + i];
#pragma unroll
            for (int i = 0; i < HF; ++i)
            {
                const uint64_t* ptr = p.rows + ((uint64_t)row[i] * p.W + wi_ld);
                if constexpr (LW == 2)
                {
                    const uint4 v = *reinterpret_cast<const uint4*>(ptr);
                    R.m[i][0] = v.x;
                    R.m[i][1] = v.y;
                    R.m[i][2] = v.z;
                    R.m[i][3] = v.w;
                }
                else
                {
                    const uint2 v = *reinterpret_cast<const uint2*>(ptr);
                    R.m[i][0] = v.x;
                    R.m[i][1] = v.y;
                }
            }
        };
        auto consume = [&](const GnRowRegs<HF, LW>& R, uint32_t it) {
            const uint32_t on = (col_act && it * H + hsub < mch) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
                uint32_t a = R.m[0][d] & on;
#pragma unroll
                for (int i = 1; i < HF; ++i)
                    a &= R.m[i][d];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    nib[d][j] += (a >> j) & 0x11111111u;
            }
            if (++acc_n == 15)
            {
                flush_nibbles();
                acc_n = 0;
            }
        };

        // straight-line double buffer: while one set is consumed the other set's loads stay in flight
        GnRowRegs<HF, LW> A, Bq;
        uint32_t          it = 0;
        issue(0, A);
        for (; it + 2 < iters; it += 2)
        {
            issue(it + 1, Bq);
            consume(A, it);
            issue(it + 2, A);
            consume(Bq, it + 1);
        }
        const bool two = it + 1 < iters;
        if (two)
            issue(it + 1, Bq);
        consume(A, it);
        if (two)
            consume(Bq, it + 1);
    }
    if (acc_n)
        flush_nibbles();
    __syncthreads();

    // ---- count of bin b of my read from the sliced LDS layout ----
    auto bin_count = [&](uint32_t b) -> uint32_t {
        const uint32_t word = b >> 6;
        const uint32_t sl   = word / (64 * LW);
        const uint32_t wl   = word - sl * 64 * LW;
        const uint32_t g    = wl / LW;
        const uint32_t tp   = (wl - g * LW) * 64 + (b & 63); // bit inside the lane's LW*64 bits
        const uint32_t d = tp >> 5, t = tp & 31;
        const uint32_t q = 16 * d + (t >> 1), half = t & 1;
        const uint32_t v = cnt_read[(size_t)sl * p.slice_dwords + q * (Gp + 1) + g];
        return half ? (v >> 16) : (v & 0xFFFFu);
    };

    const uint32_t tir   = slice * GN_WAVE + lane; // thread in read
    const uint32_t lanes = wpr * GN_WAVE;

    if (p.dense && read >= p.dense_begin && read < p.dense_end && read < p.n_reads)
    {
        uint16_t* dst = p.dense + (size_t)(read - p.dense_begin) * p.B;
        for (uint32_t b = tir; b < p.B; b += lanes)
            dst[b] = (uint16_t)(n ? bin_count(b) : 0);
    }

    // ---- select_matches: sum target bins, cap at n, cutoff, compact (GanonClassify.cpp:516-540) ----
    // threshold_cutoff = max(1, ceil(n * rel_cutoff)) in IEEE double (:492-495,720-724)
    uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
    if (T == 0)
        T = 1;

    // Candidate-driven select (split-bin maps; T >= 2, n <= 255), step 1: prefilter.
    // A target reaches T only if one of its nb bins holds at least T/nb hits, i.e. count*nb >= T.  Every lane tests
    // the bins it counted (its own LDS column) against a bins-per-target byte table (0 for bins of no target and of
    // targets with more than GN_CAND_NBIG bins, which are scanned from a list instead), both as u16 pairs: v_perm
    // unpack, packed multiply, packed saturating subtract.  The candidates of all waves of the read are added up;
    // a read with too many of them (tiny T, dense hits) keeps the plain scan over every target.
    uint32_t cand[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
        cand[d] = 0;
    bool use_cand = cand_ok && read < p.n_reads && n != 0 && n <= 255 && T >= 2;
    if (cand_ok) // kernel-uniform: the barrier below is reached by every wave of the block
    {
        uint32_t mine = 0;
        if (use_cand && col_act && hsub == 0 && !(p.nt_loads & 32u)) // one lane per LDS column
        {
            typedef unsigned short gn_u16x2 __attribute__((ext_vector_type(2)));
            const gn_u16x2 tm1 = __builtin_bit_cast(gn_u16x2, (T - 1) * 0x00010001u);
            // count pair q = (16d + qq) sits at dword q*(Gp+1)+gl; the nb bytes of pairs 2j and 2j+1 share dword
            // j*(Gp+1)+gl of the table
            auto pair_ge = [&](uint32_t q, uint32_t nb4) -> uint32_t {
                const gn_u16x2 c2  = __builtin_bit_cast(gn_u16x2, cnt[q * (Gp + 1) + gl]);
                const uint32_t sel = (q & 1u) ? 0x0C030C02u : 0x0C010C00u; // two bytes -> two u16
                const gn_u16x2 nb2 = __builtin_bit_cast(gn_u16x2, __builtin_amdgcn_perm(0u, nb4, sel));
                return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(c2 * nb2, tm1)); // count*nb <= 255*255
            };
            uint32_t any_ge = 0; // first a cheap pass: almost every lane has no candidate at all
#pragma unroll
            for (int j = 0; j < 8 * ND; ++j)
            {
                any_ge |= pair_ge(2 * j, nbreg[j]) | pair_ge(2 * j + 1, nbreg[j]);
            }
            if (any_ge)
            {
#pragma unroll
                for (int d = 0; d < ND; ++d)
                {
                    // (taken by one lane of every read that has a true match: unrolled, everything from registers/LDS)
                    uint32_t m = 0;
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq)
                    {
                        const uint32_t ge = pair_ge(16 * d + qq, nbreg[(16 * d + qq) >> 1]);
                        m |= (((ge & 0xFFFFu) ? 1u : 0u) | ((ge >> 16) ? 2u : 0u)) << (2 * qq);
                    }
                    cand[d] = m;
                    mine += (uint32_t)__popc(m);
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            mine += __shfl_xor(mine, off);
        uint32_t c_read = mine;
        if (wpr > 1) // the waves of a read must agree on which select runs
        {
            if (lane == 0)
                candcnt[wave] = mine;
            __syncthreads();
            c_read = 0;
            for (uint32_t sl = 0; sl < wpr; ++sl)
                c_read += candcnt[rslot * wpr + sl];
        }
        use_cand = use_cand && (c_read <= GN_CAND_LIMIT * wpr || (p.nt_loads & 8u)); // (bit 3: ablation -- never fall back)
    }

    if (read < p.n_reads)
    {
        // targets of this wave: [t_lo, t_hi), chunk-major inside the wave so that LDS reads are conflict free
        const uint32_t per_wave = (p.n_targets + wpr - 1) / wpr;
        const uint32_t t_lo     = min(p.n_targets, slice * per_wave);
        const uint32_t t_hi     = min(p.n_targets, t_lo + per_wave);

        // Per-target record (host-built, one 16-byte load): {first CSR entry, #bins, LDS slot of bin 0, LDS slot of
        // bin 1}; an LDS slot = (dword of the bin's u16 pair inside the read's count area) * 2 + half, the same
        // mapping as bin_count().  Targets with more than two bins read the rest through tgt_lds.
        auto lds_count = [&](uint32_t a) -> uint32_t {
            const uint32_t v = cnt_read[a >> 1];
            return (a & 1u) ? (v >> 16) : (v & 0xFFFFu);
        };
        auto rec_count = [&](const uint4& rec) -> uint32_t {
            uint32_t s = 0;
            if (rec.y >= 1)
                s += lds_count(rec.z);
            if (rec.y >= 2)
                s += lds_count(rec.w);
            for (uint32_t x = 2; x < rec.y; ++x)
                s += lds_count(p.tgt_lds[rec.x + x]);
            return s > n ? n : s; // :525-526
        };
        auto target_count = [&](uint32_t t) -> uint32_t {
            if (p.tgt_off == nullptr)
            {
                const uint32_t s = bin_count(t);
                return s > n ? n : s;
            }
            return rec_count(p.tgt_rec[t]);
        };

        // Candidate-driven select, step 2: only the candidate bins go on to bin -> target -> record.  A target is
        // reported by the wave that owns its lowest candidate bin -- every wave evaluates that the same way, so
        // nothing is reported twice -- with the exact sum over all its bins (any slice: the count area of the read
        // is complete after the barrier above).  Replaces a scan over every target's 16-byte record per read (32 KB of
        // L2 traffic per read at 2048 targets, 262 KB at 16384).
        // one candidate per lane and trip; `direct` = second pass of a read with more than GN_STAGE_CAP hits
        auto emit_hits = [&](bool emit, uint32_t tgt, uint32_t cv, uint32_t& tot, bool direct, gn_match* out) {
            const uint64_t bm = __ballot(emit);
            if (emit)
            {
                const uint32_t o = tot + __popcll(bm & ((1ULL << lane) - 1ULL));
                if (direct)
                {
                    gn_match mt;
                    mt.read   = read;
                    mt.target = tgt;
                    mt.count  = cv;
                    out[o]    = mt;
                }
                else if (o < GN_STAGE_CAP)
                {
                    stage[2 * o]     = tgt;
                    stage[2 * o + 1] = cv;
                }
            }
            tot += (uint32_t)__popcll(bm);
        };
        auto cand_rounds = [&](bool direct, gn_match* out) -> uint32_t {
            uint32_t c[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d)
                c[d] = cand[d];
            uint32_t tot = 0;
            // targets with more than GN_CAND_NBIG bins: their bins are no candidates (a big target collects random
            // hits in most of its bins); this wave scans its share of the list of such targets instead
            {
                const uint32_t per = (p.n_big + wpr - 1) / wpr;
                const uint32_t lo = min(p.n_big, slice * per), hi = min(p.n_big, lo + per);
                for (uint32_t i0 = lo; i0 < hi; i0 += GN_WAVE)
                {
                    const uint32_t i = i0 + lane;
                    bool           emit = false;
                    uint32_t       tgt = 0, cv = 0;
                    if (i < hi)
                    {
                        const uint32_t t = p.big_list[i];
                        cv               = rec_count(p.tgt_rec[t]);
                        emit             = cv >= T;
                        tgt              = p.tgt_ids ? p.tgt_ids[t] : t;
                    }
                    emit_hits(emit, tgt, cv, tot, direct, out);
                }
            }
            for (;;)
            {
                bool     have = false;
                uint32_t tp   = 0;
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (!have && c[d])
                    {
                        have = true;
                        tp   = 32u * d + (uint32_t)__builtin_ctz(c[d]);
                        c[d] &= c[d] - 1;
                    }
                if (__ballot(have) == 0)
                    break;
                bool     emit = false;
                uint32_t tgt = 0, cv = 0;
                if (have)
                {
                    const uint32_t b   = wi * 64 + tp;
                    const uint32_t t   = p.bin_tgt[b];
                    const uint4    rec = p.tgt_rec[t];
                    const uint32_t nbc = rec.y; // <= GN_CAND_NBIG: bins of bigger targets have a zero table entry
                    bool           lowest = true;
                    for (uint32_t x = 0; x < rec.y; ++x) // bins of a target ascend in the CSR
                    {
                        if (p.tgt_bins[rec.x + x] >= b)
                            break;
                        const uint32_t cx = lds_count(x == 0 ? rec.z : (x == 1 ? rec.w : p.tgt_lds[rec.x + x]));
                        if (cx * nbc >= T)
                        {
                            lowest = false;
                            break;
                        }
                    }
                    if (lowest)
                    {
                        cv   = rec_count(rec);
                        emit = cv >= T;
                        tgt  = p.tgt_ids ? p.tgt_ids[t] : t;
                    }
                }
                emit_hits(emit, tgt, cv, tot, direct, out);
            }
            return tot;
        };

        // Single pass: hits are staged in a small per-wave LDS list and copied to one reserved segment at the end; a
        // read with more than GN_STAGE_CAP hits in this slice falls back to count-then-write (two passes).
        uint32_t total    = 0;
        bool     overflow = false;
        if (use_cand)
        {
            total    = (p.nt_loads & 16u) ? 0u : cand_rounds(false, nullptr);
            overflow = total > GN_STAGE_CAP;
        }
        else if (n && !(p.nt_loads & 4u)) // (bit 2: ablation -- skip the select scan)
            for (uint32_t t0 = t_lo; t0 < t_hi; t0 += 4 * GN_WAVE)
            {
                // four 64-target chunks per trip: their records are fetched together so that the (L2-resident) table
                // latency is paid once per four chunks
                uint4 rec[4];
                if (p.tgt_off != nullptr)
                {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                    {
                        const uint32_t t = t0 + u * GN_WAVE + lane;
                        rec[u]           = t < t_hi ? p.tgt_rec[t] : make_uint4(0, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t t   = t0 + u * GN_WAVE + lane;
                    uint32_t       c   = 0;
                    bool           hit = false;
                    if (t < t_hi)
                    {
                        if (p.tgt_off != nullptr)
                            c = rec_count(rec[u]);
                        else
                        {
                            c = bin_count(t);
                            c = c > n ? n : c;
                        }
                        hit = c >= T;
                    }
                    const uint64_t bm = __ballot(hit);
                    const uint32_t nb = (uint32_t)__popcll(bm);
                    if (!overflow && total + nb <= GN_STAGE_CAP)
                    {
                        if (hit)
                        {
                            const uint32_t o = total + __popcll(bm & ((1ULL << lane) - 1ULL));
                            stage[2 * o]     = p.tgt_ids ? p.tgt_ids[t] : t;
                            stage[2 * o + 1] = c;
                        }
                    }
                    else
                        overflow = true;
                    total += nb;
                }
            }
        unsigned long long base = 0;
        if (total)
        {
            // wave-private chunk of the match buffer (see the fast kernel): one global atomic per GN_MATCH_CHUNK
            if (total > chunk_left)
            {
                const uint32_t need = total > chunk_size ? total : chunk_size;
                chunk_size = chunk_size * 2u <= GN_MATCH_CHUNK_MAX ? chunk_size * 2u : GN_MATCH_CHUNK_MAX;
                unsigned long long nb = 0;
                if (lane == 0)
                    nb = atomicAdd(p.cursor, (unsigned long long)need);
                chunk_base = gn_readlane64(nb, 0);
                chunk_left = need;
            }
            base = chunk_base;
            chunk_base += total;
            chunk_left -= total;
            if (base + total <= p.match_cap)
            {
                if (!overflow)
                {
                    gn_wave_lds_sync();
                    for (uint32_t o = lane; o < total; o += GN_WAVE)
                    {
                        gn_match mt;
                        mt.read   = read;
                        mt.target = stage[2 * o];
                        mt.count  = stage[2 * o + 1];
                        p.matches[base + o] = mt;
                    }
                    gn_wave_lds_sync();
                }
                else if (use_cand)
                    (void)cand_rounds(true, p.matches + base);
                else
                {
                    uint32_t run = 0;
                    for (uint32_t t0 = t_lo; t0 < t_hi; t0 += GN_WAVE)
                    {
                        const uint32_t t   = t0 + lane;
                        uint32_t       c   = 0;
                        bool           hit = false;
                        if (t < t_hi)
                        {
                            c   = target_count(t);
                            hit = c >= T;
                        }
                        const uint64_t bm = __ballot(hit);
                        if (hit)
                        {
                            gn_match mt;
                            mt.read   = read;
                            mt.target = p.tgt_ids ? p.tgt_ids[t] : t;
                            mt.count  = c;
                            p.matches[base + run + __popcll(bm & ((1ULL << lane) - 1ULL))] = mt;
                        }
                        run += __popcll(bm);
                    }
                }
            }
        }
        if (lane == 0)
        {
            p.seg_begin[(size_t)read * wpr + slice] = base;
            p.seg_count[(size_t)read * wpr + slice] = total;
        }
    }
    __syncthreads(); // the next round zeroes the count slices
    } // rounds
}

// build-time tuning knobs of the fast kernel (defaults = what was measured best on MI355X)
#define GN_FAST_LIST 512u // matches of a unit the fast kernel's epilogue lists in LDS before it writes them out (2 KiB a wave)
#ifndef GN_FAST_WPE_LW1
#define GN_FAST_WPE_LW1 4 // waves per SIMD the register allocator must reach, 8-byte lanes
#endif
#ifndef GN_FAST_WPE_LW2
#define GN_FAST_WPE_LW2 2 // same, 16-byte lanes, five hash functions
#endif
#ifndef GN_FAST_WPE_LW2_H4
#define GN_FAST_WPE_LW2_H4 3 // same, 16-byte lanes, up to four hash functions: 168 registers and 2 .. 15 of them in scratch (the epilogue's), measured
#endif                       // +4 % at the headline thresholds, +5 .. 7 % at low cutoffs on 32 768 bins / 8 GiB, +0.3 % on 128 GiB (round 4)

// ================================================================================================
// fast count + select kernel: identity bin->target map, reads with at most 127 minimisers
// ================================================================================================
// Same row-gather main loop, but the per-bin counters never leave the registers: 4-bit SWAR counters take 15
// iterations and are then spilled into 8-bit SWAR registers (single 150 bp reads with H = 2 never spill in the loop).
// Epilogue: partial counts of the H hash groups added with lane-xor shuffles, SWAR
// compare of every byte against the read's cutoff T ((x + 0x80 - T) & 0x80), ballot; only lanes that own a
// bin >= T extract (bin, count) pairs.  Reads with more minimisers are appended to `deferred` and handled by
// gn_ibf_count_kernel.  One wave per (read, column slice); no LDS counters, no block barriers.
// EE = with the exact early exit (instantiated separately: its check costs registers, and the variant without it
// must keep the occupancy it had).
// WIDE = rows of more than 128 words (1 KiB) with 16-byte lanes: there the third wave a SIMD bought +0.3 % (128 GiB filter, 4 KiB rows) for
// 64 bytes of scratch per lane, so those rows get the two-wave build with nothing spilled; narrower rows keep three waves (+4 .. 7 %).
template <int HF, int LW, bool EE, bool WIDE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EE ? (LW == 1 ? GN_FAST_WPE_LW1 : (HF <= 4 && !WIDE ? GN_FAST_WPE_LW2_H4 : GN_FAST_WPE_LW2)) : 1))) void gn_ibf_count_fast_kernel(GnCountParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gn_lds[];
    constexpr int      ND   = 2 * LW;
    constexpr int      HFP  = HF <= 4 ? 4 : 8;
    // counts live in 8-bit SWAR registers (4-bit first level, spilled every 15 iterations): exact up to 127, which
    // also keeps the byte-wise compare below carry-free.  Reads with more minimisers go to the generic kernel.
    constexpr uint32_t NMAX = 127;

    const int      lane  = threadIdx.x & (GN_WAVE - 1);
    const int      wave  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform: unit, read, n, ... live in SGPRs
    const uint32_t wpr   = p.wpr;
    const uint32_t Gp    = 1u << p.gp_log2;
    const uint32_t H     = GN_WAVE >> p.gp_log2;
    const uint32_t gl    = lane & (Gp - 1);
    const uint32_t hsub  = lane >> p.gp_log2;
    uint32_t*      rowtab = gn_lds + (size_t)wave * 128 * HFP;
    uint32_t*      mlist  = gn_lds + (size_t)(blockDim.x >> 6) * 128 * HFP + (size_t)wave * GN_FAST_LIST; // the epilogue's match list (low cutoffs)

    // Persistent waves: unit = (read, column slice), handed out in chunks (below).  The metadata of the NEXT unit
    // (status, n, hash slot) is loaded at the top of the current one and its hashes right after the current
    // main loop, so the chain status -> n -> slot -> hashes -> rows of dependent HBM latencies that a fresh
    // wave would pay before its first row load is hidden behind the previous read's work.
    const uint64_t n_units = (uint64_t)(p.n_reads - p.read_begin) * wpr;
    const uint64_t stride  = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint64_t       unit    = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (unit >= n_units)
        return;
    // Which units a wave takes.  Wave i used to take units i, i + waves, ...: the same count for every wave.  But the waves of a launch do
    // not run at the same pace (the HIBF's packed kernel, which hands out its batches the same way: first wave through after 0.84 of the
    // launch, the XCDs with odd numbers 11 % behind the even ones -- gn_hibf.hip), and the launch ends with its slowest wave.  With 64 or
    // more units a wave, a wave starts with 16 units of its own and takes every later chunk from a cursor -- asked for one chunk ahead,
    // so that the answer is there when it is needed; 16 units a chunk, 8 in the last quarter of the launch, 4 in its last sixteenth
    // (a counter address takes ~90 atomics a microsecond).
    const bool     dyn      = p.grab != nullptr && n_units >= 64u * stride;
    const uint64_t dyn_from = stride * 16u; // the cursor counts from here
    uint64_t       chunk_end = unit + 1;
    if (dyn)
    {
        unit *= 16u;
        chunk_end = unit + 16u;
    }
    unsigned long long g_next = 0; // (lane 0: what the cursor answered)
    uint32_t           g_size = 16u;
    auto               ask    = [&](uint64_t from) {
        g_size = from < n_units - n_units / 4u ? 16u : (from < n_units - n_units / 16u ? 8u : 4u);
        if (lane == 0)
            g_next = atomicAdd(p.grab, (unsigned long long)g_size);
    };
    if (dyn)
        ask(unit);
    // hash q of a unit feeds row-table entries idx = lane and lane + 64 (idx = q*HF + i)
    const uint32_t q0 = (uint32_t)lane / HF, q1 = ((uint32_t)lane + GN_WAVE) / HF;
    auto load_meta = [&](uint64_t u, uint32_t& rd, uint32_t& nn, uint64_t& so) {
        rd = p.read_begin + (uint32_t)(u / wpr);
        nn = p.status[rd] == GN_READ_OK ? p.n_hashes[rd] : 0u;
        so = p.slot_off[rd];
    };
    auto load_hashes = [&](uint32_t nn, uint64_t so, uint64_t& h0, uint64_t& h1) {
        h0 = h1 = 0;
        if (nn >= 1 && nn <= NMAX)
        {
            if (q0 < nn)
                h0 = p.hashes[so + q0];
            if (q1 < nn)
                h1 = p.hashes[so + q1];
        }
    };
    unsigned long long chunk_base = 0; // wave-private slice of the match buffer
    uint32_t           chunk_left = 0, chunk_size = GN_MATCH_CHUNK;
    uint64_t           skipped_bytes = 0; // row bytes this lane's column group did not fetch thanks to early exits
    uint32_t           n_pre = 0;         // matches of this lane left unwritten for the filter_matches pre-pass (pre_mode)
    uint32_t read, n;
    uint64_t slot, hA, hB;
    load_meta(unit, read, n, slot);
    load_hashes(n, slot, hA, hB);

    for (;;)
    {
    const uint32_t slice = (uint32_t)(unit - (uint64_t)(read - p.read_begin) * wpr);
    // prefetch the next unit's metadata (consumed after the main loop)
    uint64_t unit_n = unit + 1;
    if (unit_n >= chunk_end)
    {
        if (dyn)
        {
            const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g_next);
            const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(g_next >> 32));
            unit_n              = dyn_from + (((uint64_t)hi32 << 32) | lo32); // (asked for a chunk ago)
            chunk_end           = unit_n + g_size;
            if (unit_n < n_units)
                ask(unit_n);
        }
        else
        {
            unit_n    = unit + stride;
            chunk_end = unit_n + 1;
        }
    }
    const bool more = unit_n < n_units;
    uint32_t       read_n = 0, n_n = 0;
    uint64_t       slot_n = 0, hA_n = 0, hB_n = 0;
    if (more)
        load_meta(unit_n, read_n, n_n, slot_n);

    if (n > NMAX || n == 0)
    {
        if (lane == 0)
        {
            if (n > NMAX && slice == 0)
                p.work_list_out[atomicAdd(p.work_count_out, 1ULL)] = read;
            p.seg_begin[(size_t)read * wpr + slice] = 0;
            p.seg_count[(size_t)read * wpr + slice] = 0;
        }
        if (!more)
            break;
        load_hashes(n_n, slot_n, hA_n, hB_n);
        unit = unit_n; read = read_n; n = n_n; slot = slot_n; hA = hA_n; hB = hB_n;
        continue;
    }
    const uint32_t  wi      = slice * 64 * LW + gl * LW;
    const bool      col_act = wi < p.W;

    gn_wave_lds_sync(); // previous unit's row-table reads are done
    if ((uint32_t)lane < n * HF)
        rowtab[q0 * HFP + ((uint32_t)lane - q0 * HF)] = gn_ibf_row(hA, (uint32_t)lane - q0 * HF, p.shift, p.S);
    if ((uint32_t)lane + GN_WAVE < n * HF)
        rowtab[q1 * HFP + ((uint32_t)lane + GN_WAVE - q1 * HF)] = gn_ibf_row(hB, (uint32_t)lane + GN_WAVE - q1 * HF, p.shift, p.S);
    for (uint32_t idx = (uint32_t)lane + 2 * GN_WAVE; idx < n * HF; idx += GN_WAVE) // reads with more than 128/HF minimisers
    {
        const uint32_t q = idx / HF, i = idx - q * HF;
        rowtab[q * HFP + i] = gn_ibf_row(p.hashes[slot + q], i, p.shift, p.S);
    }
    gn_wave_lds_sync();

    uint32_t nib[ND][4];
    uint32_t byt[ND][4][2];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            nib[d][j]    = 0;
            byt[d][j][0] = 0;
            byt[d][j][1] = 0;
        }
    uint32_t acc_n = 0; // iterations since the nibbles were last spilled into the byte counters (wave-uniform)
    auto spill_nibbles = [&]() {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                byt[d][j][0] += nib[d][j] & 0x0F0F0F0Fu;
                byt[d][j][1] += (nib[d][j] >> 4) & 0x0F0F0F0Fu;
                nib[d][j] = 0;
            }
    };

    const uint32_t iters = (n + H - 1) / H;
    uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff)); // GanonClassify.cpp:492-495,720-724
    if (T == 0)
        T = 1;
    // Early exit (exact): after `done` iterations the hashes [0, done*H) are counted; a bin gains at most one per
    // remaining hash, so when no bin has reached T - (n - done*H) the read cannot match in this column slice and its
    // remaining rows need not be fetched.  First check where that bound exceeds half of the hashes seen so far
    // (random hits stay far below it, true matches far above), then every second iteration; the iteration already
    // in flight at a check is wasted, so checks that could not save a later one are not made.
    uint32_t chk1 = 0xFFFFFFFFu, chk2 = 0xFFFFFFFFu;
    if (EE && p.early_exit && T >= 2)
    {
        uint32_t c = (2 * (n - T + 1) + H - 1) / H;
        if (c < 1)
            c = 1;
        if (c + 2 <= iters)
        {
            chk1 = p.early_exit == 3 ? c : (c > 1 ? c - 1 : c); // first check (one hash before the half-way bound) ...
            chk2 = p.early_exit >= 2 && p.early_exit != 3 ? iters - 2 : c + 2; // ... then every iteration up to chk2
        }
    }
    // Row loads are issued by every lane, unconditionally (lanes without a column or past the last hash read a valid
    // word and are masked when the rows are consumed): with the loads inside an exec-masked region the compiler
    // cannot know how many are outstanding behind the set it waits for and falls back to s_waitcnt vmcnt(0) -- which
    // turned the double buffer into "issue two iterations, wait for both, consume both".
    const uint32_t wi_ld = col_act ? wi : 0u; // a word that exists in every row
    auto issue = [&](uint32_t it, GnRowRegs<HF, LW>& R) {
        uint32_t q = it * H + hsub;
        q          = q < n ? q : n - 1;
        uint32_t row[HF];
#pragma unroll
        for (int i = 0; i < HF; ++i)
            row[i] = rowtab[q * HFP + i];
#pragma unroll
        for (int i = 0; i < HF; ++i)
        {
            const uint64_t* ptr = p.rows + ((uint64_t)row[i] * p.W + wi_ld);
            if constexpr (LW == 2)
            {
                typedef uint32_t gn_u32x4 __attribute__((ext_vector_type(4)));
                const gn_u32x4* p4 = reinterpret_cast<const gn_u32x4*>(ptr);
                const gn_u32x4  v  = p.nt_loads ? __builtin_nontemporal_load(p4) : *p4; // rows are read once: optional nt hint
                R.m[i][0] = v.x;
                R.m[i][1] = v.y;
                R.m[i][2] = v.z;
                R.m[i][3] = v.w;
            }
            else
            {
                const uint2 v = *reinterpret_cast<const uint2*>(ptr);
                R.m[i][0] = v.x;
                R.m[i][1] = v.y;
            }
        }
    };
    auto consume = [&](const GnRowRegs<HF, LW>& R, uint32_t it) {
        const uint32_t on = (col_act && it * H + hsub < n) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int d = 0; d < ND; ++d)
        {
            uint32_t a = R.m[0][d] & on;
#pragma unroll
            for (int i = 1; i < HF; ++i)
                a &= R.m[i][d];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nib[d][j] += (a >> j) & 0x11111111u;
        }
        if (++acc_n == 15)
        {
            spill_nibbles();
            acc_n = 0;
        }
    };
    // Survey at a check point (EE instances run with one hash group per wave, so `done` iterations = `done` hashes):
    // which bins of this slice can still reach T?  A bin gains at most one per remaining hash, so only bins with
    // count >= t = T - (n - done) are left in the race; their bits go to sm[] (bit = bin inside the lane's words)
    // and their number is returned.  Same SWAR compare as the epilogue; spills the nibbles first.
    uint32_t sm[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
        sm[d] = 0;
    // bins with count >= tval (1 <= tval <= 127; the byte counters must be complete) -> sm[], their number is returned
    auto survivors = [&](uint32_t tval) -> uint32_t {
        const uint32_t Kt = (0x80u - tval) * 0x01010101u;
        uint32_t any_t = 0;
#pragma unroll
        for (int d = 0; d < ND; ++d)
        {
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                {
                    const uint32_t g = (byt[d][j][pp] + Kt) & 0x80808080u; // byte y <-> bit 8y + 4pp + j of dword d
                    any_t |= g;
                    m |= (g >> 7) << (4 * pp + j);
                }
            sm[d] = m;
        }
        if (__ballot(any_t != 0) == 0)
            return 0u;
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < ND; ++d)
            c += (uint32_t)__popc(sm[d]);
        for (int off = 32; off > 0; off >>= 1)
            c += __shfl_xor(c, off);
        return c;
    };
    auto survey = [&](uint32_t done) -> uint32_t {
        if (acc_n)
        {
            spill_nibbles();
            acc_n = 0;
        }
        return survivors(T - (n - done)); // >= 1 at every check point
    };
    bool     dead    = false; // no bin can reach T any more: nothing to report
    bool     narrow  = false; // a handful of bins can: the remaining hashes only look at those (below)
    uint32_t fetched = iters; // iterations whose full rows were requested
    {
        // Two row-register sets (A, Bq), two iterations per trip, straight-line: while set X is consumed the other
        // set's four loads stay in flight (s_waitcnt vmcnt(HF)).  Invariant at the top of a trip and after the loop:
        // A holds iteration `it`, in flight.  The survey sits after the second half (check points are even when a
        // wave is one hash group, and only those instances check).
        GnRowRegs<HF, LW> A, Bq;
        uint32_t          it      = 0;
        bool              drained = false; // a check on an odd iteration consumed everything that was in flight
        issue(0, A);
        for (; it + 2 < iters; it += 2)
        {
            if (EE && narrow)
                break;
            issue(it + 1, Bq);
            consume(A, it);
            if constexpr (EE)
            {
                const uint32_t done = it + 1; // odd check points: Bq (iteration `done`) is in flight
                if (done >= chk1 && done <= chk2)
                {
                    const uint32_t left = survey(done);
                    if (left == 0)
                    {
                        dead    = true;
                        fetched = done + 1;
                        break;
                    }
                    // At the very first check a stray bin that happens to be ahead is cheaper to wait out for one
                    // more iteration than to follow for the rest of the read: narrow there only when every survivor
                    // has been hit by (nearly) every hash so far, as a true match has.
                    if (left <= GN_NARROW_MAX && (done > chk1 || (done > 2 && survivors(done - 1) == left)))
                    {
                        consume(Bq, it + 1);
                        narrow  = true;
                        fetched = it + 2;
                        drained = true;
                        break;
                    }
                }
            }
            issue(it + 2, A);
            consume(Bq, it + 1);
            if constexpr (EE)
            {
                const uint32_t done = it + 2;
                if (done >= chk1 && done <= chk2)
                {
                    const uint32_t left = survey(done);
                    if (left == 0)
                    {
                        dead    = true;
                        fetched = done + 1; // the iteration in flight
                        break;
                    }
                    if (left <= GN_NARROW_MAX)
                        narrow = true; // A (iteration `done`) is consumed below, then the narrow pass takes over
                }
            }
        }
        if (!dead && !drained)
        {
            const bool two = !(EE && narrow) && it + 1 < iters; // the last one or two iterations
            if (two)
                issue(it + 1, Bq);
            consume(A, it);
            if (two)
                consume(Bq, it + 1);
            if (EE && narrow)
                fetched = it + 1;
        }
    }
    // Narrow mode (exact): the bins that lost the race keep their stale counts (< t <= T, never reported); for each
    // of the few that are left, the remaining hashes fetch only the 8-byte word that holds the bin -- lane l looks at
    // (hash l / HF, row l % HF), a ballot collects the bits, the HF rows of a hash are AND-ed with shifts of the
    // ballot -- and the hits are added to the owner lane's byte counter before the normal epilogue runs.
    uint32_t narrow_loads = 0;
    if (EE && narrow)
    {
        constexpr uint32_t QPC = GN_WAVE / HF; // hashes per pass
        uint64_t           grp = 0;            // bit q*HF for q < QPC
#pragma unroll
        for (uint32_t q = 0; q < QPC; ++q)
            grp |= 1ULL << (q * HF);
        const uint32_t l_q = (uint32_t)lane / HF, l_i = (uint32_t)lane - l_q * HF;
        uint32_t       any_sm = 0;
#pragma unroll
        for (int d = 0; d < ND; ++d)
            any_sm |= sm[d];
        uint64_t lm = __ballot(any_sm != 0);
        while (lm)
        {
            const int L = __builtin_ctzll(lm);
            lm &= lm - 1;
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
                uint32_t mb = (uint32_t)__builtin_amdgcn_readlane((int)sm[d], L);
                while (mb)
                {
                    const uint32_t bit = (uint32_t)__builtin_ctz(mb);
                    mb &= mb - 1;
                    const uint32_t tp   = 32u * d + bit;
                    const uint32_t word = slice * 64 * LW + (uint32_t)L * LW + (tp >> 6);
                    const uint32_t sh   = tp & 63u;
                    uint32_t       add  = 0;
                    for (uint32_t q0 = fetched; q0 < n; q0 += QPC)
                    {
                        const uint32_t q  = q0 + l_q;
                        bool           on = false;
                        if (l_q < QPC && q < n)
                        {
                            const uint32_t row = rowtab[q * HFP + l_i];
                            on = (p.rows[(uint64_t)row * p.W + word] >> sh) & 1ULL;
                        }
                        const uint64_t bm  = __ballot(on);
                        uint64_t       all = bm;
#pragma unroll
                        for (int i = 1; i < HF; ++i)
                            all &= bm >> i;
                        add += (uint32_t)__popcll(all & grp);
                    }
                    narrow_loads += (n - fetched) * HF;
                    // into byte (bit >> 3) of byt[d][bit & 3][(bit >> 2) & 1] of lane L
                    const uint32_t inc = add << (8u * (bit >> 3));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            if ((bit & 3u) == (uint32_t)j && ((bit >> 2) & 1u) == (uint32_t)pp && lane == L)
                                byt[d][j][pp] += inc;
                }
            }
        }
    }
    if ((dead || narrow) && slice * 64 * LW < p.W)
    {
        const uint32_t got  = fetched < n ? fetched : n;
        const uint32_t wcol = p.W - slice * 64 * LW < 64u * LW ? p.W - slice * 64 * LW : 64u * LW; // words of this slice
        const uint64_t full = (uint64_t)(n - got) * HF * wcol * 8ull;
        const uint64_t part = (uint64_t)narrow_loads * 128ull; // a narrow load fills a 128-byte L2 line (FETCH_SIZE agrees)
        skipped_bytes += full > part ? full - part : 0ull;
    }

    if (more) // next unit's hashes: the loads fly while this unit's epilogue runs
        load_hashes(n_n, slot_n, hA_n, hB_n);

    // ---- epilogue: bytes, cross-group sum, SWAR threshold ----
    uint32_t Kc  = (0x80u - T) * 0x01010101u; // 1 <= T <= n <= 127 and counts <= 127: no carry between bytes
    uint32_t any = 0;
    if (!dead)
    {
        if (acc_n)
            spill_nibbles();
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                {
                    uint32_t x = byt[d][j][pp];
                    for (uint32_t off = Gp; off < GN_WAVE; off <<= 1) // partial counts of the H hash groups
                        x += __shfl_xor(x, (int)off);
                    byt[d][j][pp] = x;
                    any |= (x + Kc) & 0x80808080u;
                }
    }
    const bool owner = hsub == 0 && col_act; // one lane per column chunk reports
    uint64_t   hm    = dead ? 0ull : __ballot(owner && any != 0);
    if (p.pre_mode && __popcll(hm) > 6)
    {
        // A filter_matches pre-pass follows (gn_postfilter.hip) and many bins passed the cutoff (chance matches at a low
        // --rel-cutoff): with the largest count of THIS unit and a lower bound of the read's minimum, the --rel-filter threshold
        // of the read can only be higher than t2 below (gn_pf_threshold is non-decreasing in both), so bins under t2 are not
        // written at all.  What the pre-pass still needs of them is their number and their smallest count (the read's minimum
        // is taken over ALL bins that passed the cutoff).  Largest / smallest count by binary search with SWAR compares.
        auto any_ge = [&](uint32_t v) -> bool { // a bin with count >= v  (1 <= v <= 127)
            const uint32_t K = (0x80u - v) * 0x01010101u;
            uint32_t       a = 0;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        a |= byt[d][j][pp] + K;
            return __ballot(owner && (a & 0x80808080u) != 0) != 0;
        };
        uint32_t lo = T, hi = n;
        while (lo < hi)
        {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (any_ge(mid))
                lo = mid;
            else
                hi = mid - 1;
        }
        const uint32_t t2 = gn_pf_threshold(lo, p.pre_mode == 1 ? T : 0u, p.pre_rel);
        if (t2 > T && t2 <= lo)
        {
            const uint32_t K2 = (0x80u - t2) * 0x01010101u;
            auto in_range = [&](uint32_t v, bool count) -> uint32_t { // bins with T <= count <= v  (v < t2): any / how many (this lane)
                const uint32_t K1 = (0x80u - (v + 1u)) * 0x01010101u;
                uint32_t       a = 0;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                        {
                            const uint32_t e = ((byt[d][j][pp] + Kc) & ~(byt[d][j][pp] + K1)) & 0x80808080u;
                            a = count ? a + (uint32_t)__popc(e) : (a | e);
                        }
                return a;
            };
            const uint32_t below = owner ? in_range(t2 - 1u, true) : 0u;
            if (__ballot(below != 0))
            {
                uint32_t a = T, b = t2 - 1u; // the smallest count among them
                while (a < b)
                {
                    const uint32_t mid = (a + b) >> 1;
                    if (__ballot(owner && in_range(mid, false) != 0))
                        b = mid;
                    else
                        a = mid + 1u;
                }
                if (lane == 0)
                    p.seg_min[(size_t)read * wpr + slice] = a;
                n_pre += below;
                Kc  = K2;
                any = 0;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            any |= (byt[d][j][pp] + Kc) & 0x80808080u;
                hm = __ballot(owner && any != 0);
            }
        }
    }
    uint32_t       total = 0;
    unsigned long long base = 0;
    if (hm)
    {
        // matches of this lane (counts are capped at n by construction: a bin is hit at most once per hash) as one bit
        // per bin: byte y of byt[d][j][pp] is bin 8y + 4pp + j of dword d, so its flag bits 7/15/23/31 shift into place
        uint32_t hit[ND];
        uint32_t mine = 0;
#pragma unroll
        for (int d = 0; d < ND; ++d)
        {
            hit[d] = 0;
            if (owner && any)
            {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        hit[d] |= ((((byt[d][j][pp] + Kc) & 0x80808080u) >> 7) << (4 * pp + j));
            }
            mine += __popc(hit[d]);
        }
        // exclusive prefix over the lanes that have matches: a scalar loop over the ballot while they are few (high
        // cutoffs: one or two lanes), a shuffle scan otherwise (low cutoffs: chance matches in most lanes)
        uint32_t my_off = 0;
        if (__popcll(hm) <= 6)
        {
            uint64_t rem = hm;
            while (rem)
            {
                const uint32_t L = (uint32_t)__builtin_ctzll(rem);
                rem &= rem - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)L);
                if ((uint32_t)lane == L)
                    my_off = total;
                total += c;
            }
        }
        else
        {
            uint32_t x = mine;
#pragma unroll
            for (int off = 1; off < GN_WAVE; off <<= 1)
            {
                const uint32_t y = (uint32_t)__shfl_up((int)x, off);
                if (lane >= off)
                    x += y;
            }
            my_off = x - mine;
            total  = (uint32_t)__builtin_amdgcn_readlane((int)x, GN_WAVE - 1);
        }
        // padding bins (>= B) of the last word never count: real filters keep them zero; be exact anyway
        // Output space comes from a wave-private chunk: one returning atomic on the global cursor per chunk instead of
        // one per read (a single address sustains only ~90 atomics/us, which would cap the kernel at a few M reads/ms).
        // A wave's chunks double from GN_MATCH_CHUNK to GN_MATCH_CHUNK_MAX: at low cutoffs (~100 chance matches per
        // read) fixed 256-match chunks meant 4 M atomics per 10 M reads -- 20 ms of the kernel.  Unused chunk tails are
        // holes; the gather pass compacts.
        if (total > chunk_left)
        {
            const uint32_t need = total > chunk_size ? total : chunk_size;
                chunk_size = chunk_size * 2u <= GN_MATCH_CHUNK_MAX ? chunk_size * 2u : GN_MATCH_CHUNK_MAX;
            unsigned long long nb = 0;
            if (lane == 0)
                nb = atomicAdd(p.cursor, (unsigned long long)need);
            chunk_base = gn_readlane64(nb, 0);
            chunk_left = need;
        }
        base = chunk_base;
        chunk_base += total;
        chunk_left -= total;
        const bool fits = base + total <= p.match_cap;
        if (__popcll(hm) > 6)
        {
            // Many lanes report (low cutoffs: ~100 chance matches a read at --rel-cutoff 0.2 on 4096 bins).  Up to round 5 every lane
            // stored its own matches straight to the segment: a store instruction then carried 64 twelve-byte records at 64 unrelated
            // offsets, and the address unit of the CU -- which the row loads of the other waves share -- took them one line at a time
            // (78.7 M such instructions per 10 M reads: the kernel ran 11 ms over its every-row time, profiles/r06_cutoff_counters).
            // Now the lanes put their matches, in segment order, into a list in LDS (bin in the unit << 8 | count; the lane's count bytes
            // go through the wave's row table, idle during the epilogue, 32 bins at a time as before) and the wave writes the list out
            // with lane i holding record i: whole lines per store.  Lists longer than GN_FAST_LIST (an eighth of the bins and more:
            // cutoffs near zero) keep the direct stores.
            const uint8_t* tabb   = reinterpret_cast<const uint8_t*>(rowtab);
            const bool     listed = total <= GN_FAST_LIST;
            if (p.nt_loads & 128u) // (emit_probe: what the epilogue costs without listing or storing anything)
                goto emitted;
            gn_match*      out    = p.matches + base + my_off;
            uint32_t       k      = 0;
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
                gn_wave_lds_sync();
                if (owner && any)
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            rowtab[(uint32_t)lane * 8u + (uint32_t)(j * 2 + pp)] = byt[d][j][pp];
                }
                gn_wave_lds_sync();
                uint32_t mbits = (fits && owner && any) ? hit[d] : 0u;
                while (mbits)
                {
                    const uint32_t b = (uint32_t)__builtin_ctz(mbits);
                    mbits &= mbits - 1;
                    // bit b = 8y + 4pp + j of dword d is byte y of byt[d][j][pp]
                    const uint32_t cnt = tabb[(uint32_t)lane * 32u + (((b & 3u) << 1) + ((b >> 2) & 1u)) * 4u + (b >> 3)];
                    const uint32_t bin = (uint32_t)(gl * LW) * 64u + 32u * (uint32_t)d + b; // bin inside the unit's column slice
                    if (listed)
                        mlist[my_off + k] = (bin << 8) | cnt;
                    else
                    {
                        gn_match mt;
                        mt.read   = read;
                        mt.target = slice * 64u * LW * 64u + bin;
                        mt.count  = cnt;
                        out[k]    = mt;
                    }
                    ++k;
                }
            }
            gn_wave_lds_sync(); // the list is complete; the next unit refills the row table
            if (listed && fits && !(p.nt_loads & 64u))
            {
                gn_match* seg = p.matches + base;
                for (uint32_t m = (uint32_t)lane; m < total; m += GN_WAVE)
                {
                    const uint32_t e = mlist[m];
                    gn_match       mt;
                    mt.read   = read;
                    mt.target = slice * 64u * LW * 64u + (e >> 8);
                    mt.count  = e & 0xFFu;
                    seg[m] = mt;
                }
                gn_wave_lds_sync(); // (the next unit's list must not overtake these reads)
            }
        }
        else if (fits && owner && any)
        {
            // every match goes to its rank among the lane's hit bits, so a read's segment leaves in ascending target
            // order (the gather pass then only copies)
            gn_match* out = p.matches + base + my_off;
            uint32_t  before = 0; // hits in the lane's lower dwords
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                    {
                        uint32_t hbits = (byt[d][j][pp] + Kc) & 0x80808080u;
                        while (hbits)
                        {
                            const uint32_t y = (uint32_t)__builtin_ctz(hbits) >> 3;
                            hbits &= hbits - 1;
                            const uint32_t bitpos = 8 * y + 4 * pp + j; // bit t = 4*(2y+pp) + j of dword d
                            gn_match mt;
                            mt.read   = read;
                            mt.target = wi * 64 + 32 * d + bitpos;
                            mt.count  = (byt[d][j][pp] >> (8 * y)) & 0xFFu;
                            out[before + __popc(hit[d] & ((1u << bitpos) - 1u))] = mt;
                        }
                    }
                before += __popc(hit[d]);
            }
        }
    }
emitted:
    if (lane == 0)
    {
        p.seg_begin[(size_t)read * wpr + slice] = base;
        p.seg_count[(size_t)read * wpr + slice] = total;
    }
    if (!more)
        break;
    unit = unit_n; read = read_n; n = n_n; slot = slot_n; hA = hA_n; hB = hB_n;
    } // persistent loop
    if (lane == 0 && skipped_bytes) // wave-uniform amount, one atomic per wave
        atomicAdd(p.skip_ctr, (unsigned long long)skipped_bytes);
    if (p.pre_mode)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            n_pre += (uint32_t)__shfl_xor((int)n_pre, off);
        if (lane == 0 && n_pre)
            atomicAdd(p.pre_ctr, (unsigned long long)n_pre);
    }
}

template <int HF, int LW, int MAXT>
static hipError_t gn_launch_count_one(const GnCountParams& p, const GnCountGeometry& g, hipStream_t st)
{
    uint32_t blocks = (p.n_reads - p.read_begin + g.rpb - 1) / g.rpb;
    if (blocks > p.max_blocks)
        blocks = p.max_blocks;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gn_ibf_count_kernel<HF, LW, MAXT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
    hipLaunchKernelGGL((gn_ibf_count_kernel<HF, LW, MAXT>), dim3(blocks), dim3(g.block), g.lds_bytes, st, p);
    return hipGetLastError();
}

template <int HF, int LW>
static hipError_t gn_launch_fast_one(const GnCountParams& p, hipStream_t st)
{
    // the early-exit variant only where a wave is one hash group (no cross-lane sums in its check)
    const bool ee = p.early_exit && p.gp_log2 == 6;
    const uint64_t units  = (uint64_t)(p.n_reads - p.read_begin) * p.wpr;
    uint32_t       blocks = (uint32_t)((units + 3) / 4);
    if (blocks > p.max_blocks_fast)
        blocks = p.max_blocks_fast;
    const size_t   lds    = 4 * (128 * (HF <= 4 ? 4 : 8) + GN_FAST_LIST) * 4; // per wave: the row table + the epilogue's match list
    if (ee && LW == 2 && p.W > 128)
        hipLaunchKernelGGL((gn_ibf_count_fast_kernel<HF, LW, true, LW == 2>), dim3(blocks), dim3(256), lds, st, p);
    else if (ee)
        hipLaunchKernelGGL((gn_ibf_count_fast_kernel<HF, LW, true>), dim3(blocks), dim3(256), lds, st, p);
    else
        hipLaunchKernelGGL((gn_ibf_count_fast_kernel<HF, LW, false>), dim3(blocks), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t gn_launch_count_fast(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st)
{
    if (p.n_reads <= p.read_begin)
        return hipSuccess;
    const bool two = g.lw == 2;
    switch (hash_funs)
    {
        case 1: return two ? gn_launch_fast_one<1, 2>(p, st) : gn_launch_fast_one<1, 1>(p, st);
        case 2: return two ? gn_launch_fast_one<2, 2>(p, st) : gn_launch_fast_one<2, 1>(p, st);
        case 3: return two ? gn_launch_fast_one<3, 2>(p, st) : gn_launch_fast_one<3, 1>(p, st);
        case 4: return two ? gn_launch_fast_one<4, 2>(p, st) : gn_launch_fast_one<4, 1>(p, st);
        case 5: return two ? gn_launch_fast_one<5, 2>(p, st) : gn_launch_fast_one<5, 1>(p, st);
        default: return hipErrorInvalidValue;
    }
}

template <int HF>
static hipError_t gn_launch_count_hf(const GnCountParams& p, const GnCountGeometry& g, hipStream_t st)
{
    if (g.block <= 256)
        return g.lw == 2 ? gn_launch_count_one<HF, 2, 256>(p, g, st) : gn_launch_count_one<HF, 1, 256>(p, g, st);
    if (g.block <= 512) // 8 waves per read: two waves per SIMD, the whole register file is available (no spills)
        return g.lw == 2 ? gn_launch_count_one<HF, 2, 512>(p, g, st) : gn_launch_count_one<HF, 1, 512>(p, g, st);
    return g.lw == 2 ? gn_launch_count_one<HF, 2, 1024>(p, g, st) : gn_launch_count_one<HF, 1, 1024>(p, g, st);
}

hipError_t gn_launch_count(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st)
{
    if (p.n_reads <= p.read_begin)
        return hipSuccess;
    switch (hash_funs)
    {
        case 1: return gn_launch_count_hf<1>(p, g, st);
        case 2: return gn_launch_count_hf<2>(p, g, st);
        case 3: return gn_launch_count_hf<3>(p, g, st);
        case 4: return gn_launch_count_hf<4>(p, g, st);
        case 5: return gn_launch_count_hf<5>(p, g, st);
        default: return hipErrorInvalidValue;
    }
}

// ================================================================================================
// reads with more than 65 535 minimisers -- what the reference classifies only when built with -DLONGREADS
// (TIntCount = uint32_t, GanonClassify.cpp:45-49; hashes_limit :674).  Rare and long (a 65 536-minimiser read is ~0.5 Mbp),
// so: one workgroup per read, uint32 counters in a global scratch slab (L2-resident), every thread owns the row words
// j = tid, tid + 256, ... and with them their bins' counters (no atomics); then the select of :516-540 over contiguous
// target ranges per thread (ascending output), 64-bit sums capped at n.
// ================================================================================================
__global__ void gn_long_list_kernel(const uint8_t* __restrict__ status, uint32_t read_begin, uint32_t n_reads,
                                    uint32_t* __restrict__ list, unsigned long long* __restrict__ count)
{
    const uint32_t r = read_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads && status[r] == GN_READ_BIG)
        list[atomicAdd(count, 1ULL)] = r;
}

__global__ __launch_bounds__(256) void gn_ibf_count_long_kernel(GnCountParams p, uint32_t* __restrict__ scratch)
{
    __shared__ uint32_t           part[256];
    __shared__ unsigned long long base_sh;
    __shared__ uint32_t           total_sh;
    const uint32_t tid   = threadIdx.x;
    uint32_t*      cnt   = scratch + (size_t)blockIdx.x * ((size_t)p.B + 64);
    const uint32_t n_work = (uint32_t)*p.work_count;
    const uint32_t hf    = p.early_exit; // (the launcher passes the number of hash functions here)
    for (uint32_t widx = blockIdx.x; widx < n_work; widx += gridDim.x)
    {
        const uint32_t read = p.work_list[widx];
        const uint32_t n    = p.n_hashes[read];
        for (uint32_t i = tid; i < p.B; i += 256)
            cnt[i] = 0;
        __syncthreads();
        const uint64_t* hs = p.hashes + p.slot_off[read];
        for (uint32_t q = 0; q < n; ++q)
        {
            const uint64_t v = hs[q];
            uint32_t       row[5];
            for (uint32_t i = 0; i < hf; ++i)
                row[i] = gn_ibf_row(v, i, p.shift, p.S);
            for (uint32_t j = tid; j < p.W; j += 256)
            {
                uint64_t m = p.rows[(uint64_t)row[0] * p.W + j];
                for (uint32_t i = 1; i < hf; ++i)
                    m &= p.rows[(uint64_t)row[i] * p.W + j];
                while (m)
                {
                    const uint32_t b = j * 64u + (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    if (b < p.B)
                        cnt[b]++;
                }
            }
        }
        __syncthreads();
        uint64_t T = (uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff)); // GanonClassify.cpp:492-495,720-724
        if (T == 0)
            T = 1;
        // targets [lo, hi) of this thread: contiguous, so the read's matches come out in ascending target order
        const uint32_t per = (p.n_targets + 255u) / 256u;
        const uint32_t lo  = tid * per < p.n_targets ? tid * per : p.n_targets;
        const uint32_t hi  = lo + per < p.n_targets ? lo + per : p.n_targets;
        auto           summed = [&](uint32_t t) -> uint64_t {
            uint64_t c = 0;
            if (!p.tgt_off)
                c = cnt[t];
            else
                for (uint32_t e = p.tgt_off[t]; e < p.tgt_off[t + 1]; ++e)
                    c += cnt[p.tgt_bins[e]];
            return c > n ? (uint64_t)n : c; // :525-526
        };
        uint32_t mine = 0;
        for (uint32_t t = lo; t < hi; ++t)
            mine += summed(t) >= T ? 1u : 0u;
        part[tid] = mine;
        __syncthreads();
        if (tid == 0)
        {
            uint32_t run = 0;
            for (uint32_t i = 0; i < 256; ++i)
            {
                const uint32_t c = part[i];
                part[i]          = run;
                run += c;
            }
            base_sh = run ? atomicAdd(p.cursor, (unsigned long long)run) : 0ull;
            for (uint32_t sl = 0; sl < p.wpr; ++sl)
            {
                p.seg_begin[(size_t)read * p.wpr + sl] = sl == 0 ? base_sh : 0ull;
                p.seg_count[(size_t)read * p.wpr + sl] = sl == 0 ? run : 0u;
            }
            total_sh = run;
        }
        __syncthreads();
        const uint32_t           total = total_sh;
        const unsigned long long base  = base_sh;
        if (base + total <= p.match_cap)
        {
            uint32_t k = part[tid];
            for (uint32_t t = lo; t < hi; ++t)
            {
                const uint64_t c = summed(t);
                if (c >= T)
                {
                    gn_match mt;
                    mt.read   = read;
                    mt.target = p.tgt_ids ? p.tgt_ids[t] : t;
                    mt.count  = (uint32_t)c;
                    p.matches[base + k++] = mt;
                }
            }
        }
        __syncthreads();
    }
}

hipError_t gn_launch_count_long(const GnCountParams& p, uint32_t hash_funs, uint32_t* list, unsigned long long* count, uint32_t* scratch,
                                uint32_t blocks, hipStream_t st)
{
    const uint32_t n = p.n_reads - p.read_begin;
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(gn_long_list_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.status, p.read_begin, p.n_reads, list, count);
    GnCountParams q = p;
    q.work_list  = list;
    q.work_count = count;
    q.early_exit = hash_funs;
    hipLaunchKernelGGL(gn_ibf_count_long_kernel, dim3(blocks), dim3(256), 0, st, q, scratch);
    return hipGetLastError();
}

// ================================================================================================
// emplace (GPU side of the fixture/bench filter builder): set bit (row_i(v), bin) for i < h
// ================================================================================================
__global__ void gn_emplace_kernel(uint64_t* rows, uint64_t S, uint32_t W, uint32_t shift, uint32_t h,
                                  const uint64_t* __restrict__ hashes, const uint32_t* __restrict__ bins, uint64_t n)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * h)
        return;
    const uint64_t q   = idx / h;
    const uint32_t i   = (uint32_t)(idx - q * h);
    const uint32_t bin = bins[q];
    const uint32_t row = gn_ibf_row(hashes[q], i, shift, S);
    atomicOr(reinterpret_cast<unsigned long long*>(rows + ((uint64_t)row * W + (bin >> 6))), 1ULL << (bin & 63));
}

hipError_t gn_launch_emplace(uint64_t* rows, uint64_t S, uint32_t W, uint32_t shift, uint32_t h, const uint64_t* hashes,
                             const uint32_t* bins, uint64_t n, hipStream_t st)
{
    if (n == 0)
        return hipSuccess;
    const uint64_t total  = n * h;
    const uint32_t blocks = (uint32_t)((total + 255) / 256);
    hipLaunchKernelGGL(gn_emplace_kernel, dim3(blocks), dim3(256), 0, st, rows, S, W, shift, h, hashes, bins, n);
    return hipGetLastError();
}
