// gn_minimiser_lpr.hip -- lane-per-read minimiser kernel for short reads (Illumina-style batches).
//
// Same semantics as gn_minimiser_kernel (seqan3::views::minimiser_hash, call sites
// /root/reference/src/ganon-classify/GanonClassify.cpp:647-650,693-700) but organised the
// other way round: every LANE owns one read and rolls its canonical k-mer hash base by base, so a wavefront
// handles 64 reads with ~50 VALU instructions per base instead of ~700 per read-tile.  All lanes sit at the same
// position index, which makes the sliding-window bookkeeping wave-uniform:
//   * window minimum in O(1) per base with the van Herk / Gil-Werman block decomposition: k-mer values are cut
//     into blocks of K = w-k+1; a running prefix minimum (registers) covers the current block, a suffix-minimum
//     table of the previous block lives in an LDS sliding window laid out [slot][lane] (conflict free); the
//     table is rebuilt once per block by a K-step backward scan that every lane executes at the same time
//   * ties keep the RIGHTMOST position (less_equal forward, strict-less backward), as seqan3 does
//   * emission = the state machine of App. D in its seed/expiry form: emit window j when j == 0, when the
//     remembered minimiser leaves (j == expiry) or when a strictly smaller value enters; expiry = R_j + 1
// Reads longer than lpr_max_len (or windows wider than 65 k-mers) are appended to a deferred list that the
// wave-per-read kernel (gn_kernels.hip) processes.
#include "gn_internal.h"
#include <type_traits>

#define GN_WAVE 64

namespace
{
struct LprRankLut
{
    uint8_t t[256];
    constexpr LprRankLut() : t{}
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
} // namespace
__constant__ LprRankLut GN_LPR_RANK_LUT = LprRankLut();

__device__ __forceinline__ uint32_t gn_lpr_wave_max(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        const uint32_t o = __shfl_xor(v, off);
        v                = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long gn_lpr_wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

typedef uint32_t gn_u32x4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(64) void gn_minimiser_lpr_kernel(GnMinimiserParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gn_lpr_smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t k = p.k, w = p.w;
    const uint32_t K = w - k + 1;
    uint64_t*      bufV = reinterpret_cast<uint64_t*>(gn_lpr_smem);   // [2][K][64] values (then suffix minima)
    uint8_t*       bufP = gn_lpr_smem + (size_t)2 * K * GN_WAVE * 8;  // [2][K][64] suffix argmin, offset in block
    const uint64_t seed = 0x8F3F73B5CF1C9ADEULL >> (64u - 2u * k);    // adjust_seed.hpp:33-37
    const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);

    const uint32_t r   = p.read_begin + blockIdx.x * GN_WAVE + lane;
    const bool     inr = r < p.n_reads;
    uint64_t       b1 = 0, len1 = 0, b2 = 0, len2 = 0;
    if (inr)
    {
        b1   = p.off1[r];
        len1 = p.off1[r + 1] - b1;
        if (p.off2)
        {
            b2   = p.off2[r];
            len2 = p.off2[r + 1] - b2;
        }
    }
    const bool too_long = len1 > p.lpr_max_len || len2 > p.lpr_max_len;
    const bool mine     = inr && !too_long;
    if (inr && too_long)
        p.defer_list[atomicAdd(p.defer_count, 1ULL)] = r;

    uint8_t  st = GN_READ_OK;
    uint32_t n  = 0;
    if (mine && len1 < w) // GanonClassify.cpp:690,743-747
        st = GN_READ_SMALL;
    uint64_t* out = p.hashes + (mine ? p.slot_off[r] : 0);

    for (uint32_t seg = 0; seg < 2; ++seg)
    {
        const uint64_t L64  = seg ? len2 : len1;
        const bool     act  = mine && st == GN_READ_OK && L64 >= w; // :690 / :695
        const uint32_t Leff = act ? (uint32_t)L64 : 0u;
        const uint32_t Lmax = gn_lpr_wave_max(Leff);
        if (Lmax == 0)
            continue;
        // Bases are fetched as 4-byte-aligned dwords, 16 bytes per load and two loads ahead of their use, so the
        // HBM/L2 latency is not in the per-base dependency chain.  Base i is byte (o + i) of the dword stream.
        const uintptr_t  sa  = reinterpret_cast<uintptr_t>(p.bases + (seg ? b2 : b1));
        const uint32_t   o   = (uint32_t)(sa & 3u);
        const uint32_t*  dws = reinterpret_cast<const uint32_t*>(sa & ~(uintptr_t)3);
        const uint32_t   ndw = Leff ? (o + Leff + 3) / 4 : 0; // dwords this lane may touch
        // dwords q..q+3, loaded unconditionally (a load inside an exec-masked region makes the compiler wait with
        // vmcnt(0), i.e. for the prefetches too): past the lane's own range the index is clamped -- those bytes are never
        // looked at, and the base buffer carries 64 bytes of padding for the last read
        auto load4 = [&](uint32_t q) -> gn_u32x4u {
            const uint32_t qc = q < ndw ? q : (ndw ? ndw - 1 : 0u);
            return *reinterpret_cast<const gn_u32x4u*>(dws + qc);
        };
        gn_u32x4u cur4 = load4(0), nxt4 = load4(4);

        uint64_t f = 0, rc = 0;
        uint64_t pre_v = ~0ULL, Wprev = 0;
        uint32_t pre_p = 0, expiry = 0xFFFFFFFFu;
        uint32_t blk = 0, pin = 0; // collecting buffer, position inside the current block (both wave-uniform)

        auto position = [&](uint32_t i, uint32_t c) {
            const bool on = i < Leff;
            uint64_t   v  = ~0ULL;
            if (on)
            {
                // A C G T U (either case): rank = ((c>>1) ^ (c>>2)) & 3 without touching memory; anything else
                // (IUPAC codes, garbage) takes the table -- a rare, wave-level branch
                uint64_t       b      = ((c >> 1) ^ (c >> 2)) & 3u;
                const bool     simple = ((0x0030008Au >> (c & 31u)) & 1u) && (c & 0xC0u) == 0x40u; // letters 1,3,7,20,21
                if (!simple)
                    b = GN_LPR_RANK_LUT.t[c];
                f                = ((f << 2) | b) & mask;
                rc               = (rc >> 2) | ((3ULL - b) << (2 * (k - 1)));
                const uint64_t x = f ^ seed, y = rc ^ seed;
                v                = x < y ? x : y;
            }
            if (i + 1 < k)
                return;
            const uint32_t pk = i + 1 - k; // k-mer position (uniform)
            // collect the value, update the running prefix minimum of this block (rightmost on ties)
            bufV[((size_t)blk * K + pin) * GN_WAVE + lane] = v;
            if (pin == 0 || v <= pre_v)
            {
                pre_v = v;
                pre_p = pk;
            }
            if (pk + 1 >= K)
            {
                const uint32_t j = pk + 1 - K; // window index (uniform); valid for this lane iff `on`
                uint64_t       W = pre_v;
                uint32_t       R = pre_p;
                if (pin != K - 1)
                {
                    // window starts inside the previous block: suffix minimum from slot (j mod K) = pin + 1
                    const size_t   s  = ((size_t)(blk ^ 1u) * K + (pin + 1)) * GN_WAVE + lane;
                    const uint64_t sv = bufV[s];
                    if (sv < pre_v) // tie -> the prefix side (further right)
                    {
                        W = sv;
                        R = (pk - pin - K) + bufP[s];
                    }
                }
                if (on)
                {
                    const bool emit = j == 0 || j == expiry || v < Wprev;
                    if (emit)
                    {
                        out[n++] = W;
                        expiry   = R + 1u;
                    }
                    Wprev = W;
                }
            }
            if (pin == K - 1)
            {
                // block complete: turn its values into suffix minima (backward, strict less keeps the rightmost)
                uint64_t run_v = ~0ULL;
                uint32_t run_p = K - 1;
                for (int t = (int)K - 1; t >= 0; --t)
                {
                    const size_t   s = ((size_t)blk * K + (uint32_t)t) * GN_WAVE + lane;
                    const uint64_t x = bufV[s];
                    if (x < run_v || t == (int)K - 1)
                    {
                        run_v = x;
                        run_p = (uint32_t)t;
                    }
                    bufV[s] = run_v;
                    bufP[s] = (uint8_t)run_p;
                }
                blk ^= 1u;
                pin = 0;
            }
            else
                ++pin;
        };

        // 16 bases per trip: group g of 4 bases reads bytes o..o+6 of the 64-bit window {D[g], D[g+1]}
        for (uint32_t i0 = 0; i0 < Lmax; i0 += 16)
        {
            const gn_u32x4u nn4 = load4(i0 / 4 + 8); // consumed two trips from now
            uint32_t        d[8] = { cur4.x, cur4.y, cur4.z, cur4.w, nxt4.x, nxt4.y, nxt4.z, nxt4.w };
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const uint64_t win = ((uint64_t)d[u + 1] << 32) | d[u];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                {
                    const uint32_t i = i0 + 4 * u + t;
                    if (i < Lmax)
                        position(i, (uint32_t)(win >> (8u * (o + t))) & 0xFFu);
                }
            }
            cur4 = nxt4;
            nxt4 = nn4;
        }
    }
    unsigned long long mine_total = 0;
    if (mine)
    {
        if (n > 65535u) // :674,706
            st = GN_READ_BIG;
        else if (st == GN_READ_OK)
            mine_total = n;
        p.n_hashes[r] = n;
        p.status[r]   = st;
    }
    mine_total = gn_lpr_wave_sum(mine_total);
    if (lane == 0 && mine_total)
        atomicAdd(p.total_hashes + (blockIdx.x & 63u), mine_total);
}

// ================================================================================================
// Register-resident variant for compile-time window widths (KW = w-k+1 k-mers, 1..16) and k <= 24.
// ================================================================================================
// Same lane-per-read organisation and the same block decomposition, but
//   * a k-mer travels as ONE 64-bit key (value << 16 | 0xFFFF - position): the smaller key is the smaller value and, among
//     equal values, the one further right -- so "rightmost minimum" is a plain unsigned minimum, its position comes out of
//     the key, and no separate argmin is carried around (2k + 16 <= 64);
//   * the block of KW keys being collected and the suffix minima of the previous block live in REGISTERS: the block loop
//     is unrolled by KW, every index is a compile-time constant, the once-per-block backward scan is KW-1 register minima
//     -- the sliding window needs no LDS at all;
//   * bases are fetched one block ahead as aligned dwords and cut out with byte-align instructions;
//   * emitted minimisers are staged in LDS ([slot][lane], padded) and written out per read in one coalesced burst per
//     mate, instead of one scattered 8-byte store per emission (the scattered stores cost 3.6 x the written bytes in HBM
//     traffic, r01 WRITE_SIZE).
#define GN_LPRK_STAGE 24u // staged emissions per mate and lane; further ones of a mate go straight to memory

// min of two 64-bit keys.  FMIN: both are below 2^56 or the all-ones sentinel, so read as IEEE doubles they are non-negative
// finite numbers (denormals included: f64 denormals are never flushed on gfx9) or a quiet NaN, whose order is the order of
// the integers and for which v_min_f64 returns the other operand -- one VALU instruction instead of compare + two selects
// (the inline asm keeps the compiler from adding canonicalising operations around it).  The selected operand comes back
// bit for bit; NaN against NaN gives a NaN, which is still above every key.
template <bool FMIN>
__device__ __forceinline__ uint64_t gn_key_min(uint64_t a, uint64_t b)
{
    if constexpr (FMIN)
    {
        uint64_t r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    else
        return a < b ? a : b;
}

template <int KW, bool FMIN>
__global__ __launch_bounds__(64) void gn_minimiser_lprk_kernel(GnMinimiserParams p)
{
    __shared__ uint64_t stage[GN_LPRK_STAGE * (GN_WAVE + 1)];
    const uint32_t lane = threadIdx.x;
    // dna4 rank of every byte value (seqan3::dna4 char_to_rank; the same values as GN_LPR_RANK_LUT):
    // letters of either case index two 32-bit masks with c & 31 -- low rank bit set for C Y S B T U, high rank bit for
    // G K T U; every other byte is rank 0
    __shared__ uint8_t rank_lut[256];
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t)
    {
        const uint32_t c   = lane * 4 + t, idx = c & 31u;
        uint32_t       b32 = ((0x0238000Cu >> idx) & 1u) | (((0x00300880u >> idx) & 1u) << 1);
        rank_lut[c]        = (uint8_t)((c & 0xC0u) == 0x40u ? b32 : 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t k    = p.k;
    const uint64_t seed = 0x8F3F73B5CF1C9ADEULL >> (64u - 2u * k);    // adjust_seed.hpp:33-37
    const uint64_t mask = (1ULL << (2 * k)) - 1ULL;                   // k <= 24
    const uint32_t w    = k + KW - 1;

    const uint32_t r   = p.read_begin + blockIdx.x * GN_WAVE + lane;
    const bool     inr = r < p.n_reads;
    uint64_t       b1 = 0, len1 = 0, b2 = 0, len2 = 0;
    if (inr)
    {
        b1   = p.off1[r];
        len1 = p.off1[r + 1] - b1;
        if (p.off2)
        {
            b2   = p.off2[r];
            len2 = p.off2[r + 1] - b2;
        }
    }
    const bool too_long = len1 > p.lpr_max_len || len2 > p.lpr_max_len;
    const bool mine     = inr && !too_long;
    if (inr && too_long)
        p.defer_list[atomicAdd(p.defer_count, 1ULL)] = r;

    uint8_t  st = GN_READ_OK;
    uint32_t n  = 0;
    if (mine && len1 < w) // GanonClassify.cpp:690,743-747
        st = GN_READ_SMALL;
    const uint64_t out_base = mine ? p.slot_off[r] : 0;

    for (uint32_t seg = 0; seg < 2; ++seg)
    {
        const uint64_t L64  = seg ? len2 : len1;
        const bool     act  = mine && st == GN_READ_OK && L64 >= w; // :690 / :695
        const uint32_t Leff = act ? (uint32_t)L64 : 0u;
        const uint32_t Lmax = gn_lpr_wave_max(Leff);
        if (Lmax == 0)
            continue;
        const uint32_t Lmin = ~gn_lpr_wave_max(~Leff); // 0 as soon as one lane sits this mate out
        const uint32_t n_seg0 = n; // emissions of this mate are staged from here
        const uintptr_t sa  = reinterpret_cast<uintptr_t>(p.bases + (seg ? b2 : b1));
        const uint32_t  o   = (uint32_t)(sa & 3u);
        const uint32_t* dws = reinterpret_cast<const uint32_t*>(sa & ~(uintptr_t)3);
        const uint32_t  ndw = Leff ? (o + Leff + 3) / 4 : 0; // dwords this lane may touch
        auto dword_at = [&](uint32_t q) -> uint32_t { // clamped: bytes past the lane's range are never looked at
            const uint32_t qc = q < ndw ? q : (ndw ? ndw - 1 : 0u);
            return dws[qc];
        };

        // f = forward hash; rcx = reverse-complement hash XOR (4^k - 1): complementing turns the digit (3 - b) that enters
        // at the top into b itself, so both updates are shift + or, and the complement is folded into the seed once
        uint64_t       f = 0, rcx = mask;
        const uint64_t seed_rc = seed ^ mask;
        const uint32_t top     = 2 * (k - 1);
        auto roll = [&](uint32_t c) -> uint64_t { // append base c, return the canonical value of the k-mer ending here
            // dna4 rank of any byte: a 256-byte table in LDS (filled at kernel start, see there), one ds_read_u8 -- the
            // LDS port is idle in this VALU-bound loop, the ~9 VALU operations of the bit-mask form were not free
            // (2.82 -> 2.69 ms per 10 M reads)
            const uint64_t b = rank_lut[c];
            f                  = ((f << 2) | b) & mask;
            rcx                = (rcx >> 2) | (b << top);
            const uint64_t x = f ^ seed, y = rcx ^ seed_rc;
            return gn_key_min<FMIN>(x, y);
        };
        // warm-up: the first k-1 bases complete no k-mer (one aligned dword per four bases)
        for (uint32_t q = 0; q * 4 < o + k - 1; ++q)
        {
            const uint32_t dw = dword_at(q);
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
            {
                const uint32_t by = q * 4 + t;
                if (by >= o && by - o + 1 < k)  // (lane-dependent start offset o; lanes past their read: harmless)
                    (void)roll((dw >> (8u * t)) & 0xFFu);
            }
        }

        uint64_t V[KW], S[KW];
#pragma unroll
        for (int t = 0; t < KW; ++t)
            V[t] = S[t] = ~0ULL;
        uint64_t pre = ~0ULL, Wprev = 0;
        uint32_t expiry = 0xFFFFFFFFu;
        const uint32_t M = Lmax - k + 1; // k-mers of the longest read of the wave

        // bytes [o + k-1 + blk*KW, +KW) of the dword stream = 5 aligned dwords (KW <= 16), fetched one block ahead
        constexpr int NDW = (KW + 3) / 4 + 1;
        uint32_t      cur[NDW], nxt[NDW];
        auto fetch = [&](uint32_t blk, uint32_t* dst) {
            const uint32_t q = (o + k - 1 + blk * KW) >> 2;
#pragma unroll
            for (int d = 0; d < NDW; ++d)
                dst[d] = dword_at(q + d);
        };
        fetch(0, cur);
        for (uint32_t blk = 0; blk * KW < M; ++blk)
        {
            fetch(blk + 1, nxt);
            const uint32_t sh = (o + k - 1 + blk * KW) & 3u; // byte offset inside cur[0]
            uint32_t       by[NDW - 1];
#pragma unroll
            for (int d = 0; d + 1 < NDW; ++d)
                by[d] = __builtin_amdgcn_alignbyte(cur[d + 1], cur[d], sh);
            // a block that lies inside EVERY lane's read (the usual case: 64 reads of one length) needs no per-lane
            // predicate; the general form handles the blocks around the ends of shorter reads
            auto block = [&](auto all_on_tag) {
            constexpr bool ALL_ON = decltype(all_on_tag)::value;
#pragma unroll
            for (int pin = 0; pin < KW; ++pin)
            {
                const uint32_t pk = blk * KW + pin;     // k-mer position (wave-uniform)
                const uint32_t i  = pk + k - 1;         // base index
                const bool     on = ALL_ON || i < Leff; // (past the longest read of the wave nothing is on: the tail of the
                                                        //  last block runs empty instead of leaving the unrolled loop)
                const uint32_t c  = (by[pin >> 2] >> (8 * (pin & 3))) & 0xFFu;
                // (rolled by every lane: lanes past their read only waste the arithmetic, their key is discarded)
                const uint64_t kv  = (roll(c) << 16) | (uint64_t)(0xFFFFu - pk);
                const uint64_t key = on ? kv : ~0ULL;
                V[pin] = key;
                pre    = pin == 0 ? key : gn_key_min<FMIN>(key, pre);
                if (pk + 1 >= KW)
                {
                    const uint32_t j = pk + 1 - KW; // window index (uniform); valid for this lane iff `on`
                    uint64_t       W = pre;
                    if (pin != KW - 1)              // window starts inside the previous block: its suffix from slot pin+1
                        W = gn_key_min<FMIN>(S[(pin + 1) % KW], pre);
                    if (on)
                    {
                        // first window / the remembered minimiser leaves / a strictly smaller VALUE enters
                        const bool emit = j == 0 || j == expiry || (key | 0xFFFFull) < (Wprev & ~0xFFFFull);
                        if (emit)
                        {
                            const uint32_t e = n - n_seg0;
                            if (e < GN_LPRK_STAGE)
                                stage[e * (GN_WAVE + 1) + lane] = W >> 16;
                            else
                                p.hashes[out_base + n] = W >> 16;
                            ++n;
                            expiry = (0xFFFFu - (uint32_t)(W & 0xFFFFu)) + 1u;
                        }
                        Wprev = W;
                    }
                }
            }
            };
            if ((blk + 1) * KW + k - 1 <= Lmin)
                block(std::true_type{});
            else
                block(std::false_type{});
            // block complete: its keys become suffix minima (register to register)
            S[KW - 1] = V[KW - 1];
#pragma unroll
            for (int t = KW - 2; t >= 0; --t)
                S[t] = gn_key_min<FMIN>(V[t], S[t + 1]);
#pragma unroll
            for (int d = 0; d < NDW; ++d)
                cur[d] = nxt[d];
        }
        // flush this mate's staged emissions: read after read, one coalesced store of up to GN_LPRK_STAGE hashes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t cnt_mine = act ? (n - n_seg0 < GN_LPRK_STAGE ? n - n_seg0 : GN_LPRK_STAGE) : 0u;
        uint64_t       todo     = __ballot(cnt_mine != 0);
        while (todo)
        {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_mine, l);
            const uint64_t dst0 = out_base + n_seg0;
            const uint64_t dst  = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(dst0 >> 32), l) << 32) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dst0, l);
            if (lane < cnt)
                p.hashes[dst + lane] = stage[lane * (GN_WAVE + 1) + (uint32_t)l];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    unsigned long long mine_total = 0;
    if (mine)
    {
        if (n > 65535u) // :674,706
            st = GN_READ_BIG;
        else if (st == GN_READ_OK)
            mine_total = n;
        p.n_hashes[r] = n;
        p.status[r]   = st;
    }
    mine_total = gn_lpr_wave_sum(mine_total);
    if (lane == 0 && mine_total)
        atomicAdd(p.total_hashes + (blockIdx.x & 63u), mine_total);
}

template <int KW>
static void gn_launch_lprk(const GnMinimiserParams& p, hipStream_t st)
{
    const dim3 grid((p.n_reads - p.read_begin + GN_WAVE - 1) / GN_WAVE);
    // keys are (value << 16) | position with value < 4^k: below 2^56 -- the range gn_key_min's float form needs -- up to k = 20
    if (p.k <= 20)
        hipLaunchKernelGGL((gn_minimiser_lprk_kernel<KW, true>), grid, dim3(GN_WAVE), 0, st, p);
    else
        hipLaunchKernelGGL((gn_minimiser_lprk_kernel<KW, false>), grid, dim3(GN_WAVE), 0, st, p);
}

hipError_t gn_launch_minimiser_lpr(const GnMinimiserParams& p, hipStream_t st)
{
    if (p.n_reads <= p.read_begin)
        return hipSuccess;
    const uint32_t K   = p.w - p.k + 1;
    if (K <= 16 && p.k <= 24)
    {
        switch (K)
        {
            case 1: gn_launch_lprk<1>(p, st); break;
            case 2: gn_launch_lprk<2>(p, st); break;
            case 3: gn_launch_lprk<3>(p, st); break;
            case 4: gn_launch_lprk<4>(p, st); break;
            case 5: gn_launch_lprk<5>(p, st); break;
            case 6: gn_launch_lprk<6>(p, st); break;
            case 7: gn_launch_lprk<7>(p, st); break;
            case 8: gn_launch_lprk<8>(p, st); break;
            case 9: gn_launch_lprk<9>(p, st); break;
            case 10: gn_launch_lprk<10>(p, st); break;
            case 11: gn_launch_lprk<11>(p, st); break;
            case 12: gn_launch_lprk<12>(p, st); break;
            case 13: gn_launch_lprk<13>(p, st); break;
            case 14: gn_launch_lprk<14>(p, st); break;
            case 15: gn_launch_lprk<15>(p, st); break;
            default: gn_launch_lprk<16>(p, st); break;
        }
        return hipGetLastError();
    }
    const size_t   lds = (size_t)2 * K * GN_WAVE * 9;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gn_minimiser_lpr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(gn_minimiser_lpr_kernel, dim3((p.n_reads - p.read_begin + GN_WAVE - 1) / GN_WAVE), dim3(GN_WAVE), lds, st, p);
    return hipGetLastError();
}
