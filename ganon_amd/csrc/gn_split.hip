// gn_split.hip -- count + select for split-bin maps (a target owns several technical bins, GanonClassify.cpp:516-540),
// reads with at most 127 minimisers, rows of at least one full wave (one hash per wave iteration).
//
// The generic kernel (gn_kernels.hip) keeps the 16-bit counters of a whole read in LDS -- 2 bytes per bin, flushed
// from the nibble registers with LDS atomics and zeroed for every read -- which on wide rows leaves room for one block
// per CU.  Here the counters stay in registers exactly as in the fast kernel (4-bit SWAR, spilled into 8-bit SWAR),
// and LDS only ever holds a byte image of them, written when the read has a candidate at all:
//   1. prefilter in registers: a target with nb bins reaches T only through a bin with count*nb >= T; the bins-per-
//      target bytes of the bins a lane counts are the same for every read of a persistent wave (8*ND registers, in the
//      layout of the byte counters); targets with more than GN_CAND_NBIG bins have byte 0 and are scanned from a list
//   2. the waves of a read add up their candidates (block barrier); no candidate and no big target -> done
//   3. otherwise every wave of the read stores its byte image, and the candidate bins go on to
//      bin -> target -> CSR of bins: a target is reported by the wave that owns its lowest candidate bin, with the exact
//      sum over all its bins (any slice); reads with more candidates than the budget scan every target instead
// Reads with more minimisers are handed to the generic kernel through the work list.
#include "gn_internal.h"

#include <hip/hip_runtime.h>

#define GN_WAVE 64
#define GN_SPLIT_STAGE 128u
#define GN_SPLIT_LIMIT 128u // candidates per wave beyond which the read scans every target
#define GN_SPLIT_CHUNK 256u
#define GN_SPLIT_CHUNK_MAX 8192u

#define GN_SPLIT_RUN_SELECT(MAXT) ((MAXT) <= 512) // (the 1024-thread variants have 128 registers a lane: not there)

namespace
{

__device__ __forceinline__ void gn_sp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t gn_sp_readlane64(uint64_t v, uint32_t l)
{
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// IBF row of hash v for hash function i (seqan3 hash_and_fit) -- same as gn_kernels.hip
__device__ __forceinline__ uint32_t gn_sp_row(uint64_t v, uint32_t i, uint32_t shift, uint64_t S)
{
    constexpr uint64_t seeds[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST;   // include/ganon_ibf_hash.h
    uint64_t x = v * seeds[i];
    x ^= x >> shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint32_t)__umul64hi(x, S);
}

template <int HF, int LW>
struct GnSpRows
{
    uint32_t m[HF][2 * LW];
};

} // namespace

// RS: 0 = the general instantiation (bins-per-target registers, candidate select, scan); 1 = maps whose targets own runs of consecutive
// bins of mixed lengths (p.run_select: running-sum select); 2 = uniform maps (p.uniform_nb: packed select).  1 and 2 carry no
// bins-per-target registers -- the prefilter takes every bin for one of p.const_nb, a byte compare -- and not the selects they never run.
template <int HF, int LW, int MAXT, int RS>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT > 512 ? 4 : (LW == 2 || MAXT > 256 ? 2 : 3)))) void gn_ibf_count_split_kernel(GnCountParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gn_sp_lds[];
    constexpr int      ND   = 2 * LW;
    constexpr int      HFP  = HF <= 4 ? 4 : 8;
    constexpr uint32_t NMAX = 127;
    constexpr uint32_t IMG  = 8 * ND * 64; // dwords of one wave's byte image
    constexpr bool     RUNSEL = RS == 1; // 1: maps of mixed widths (run select), 2: uniform maps (packed select), both without nbreg

    const int      lane   = threadIdx.x & (GN_WAVE - 1);
    const int      wave   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wpr    = p.wpr;
    const uint32_t rpb    = nwaves / wpr;
    const uint32_t rslot  = wave / wpr;
    const uint32_t slice  = wave % wpr;

    uint32_t* img_read = gn_sp_lds + (size_t)rslot * wpr * IMG; // byte images of all slices of my read
    uint32_t* rowtab   = gn_sp_lds + (size_t)rpb * wpr * IMG + (size_t)wave * 128 * HFP;
    uint32_t* stage    = gn_sp_lds + (size_t)rpb * wpr * IMG + (size_t)nwaves * 128 * HFP + (size_t)wave * 2 * GN_SPLIT_STAGE;
    uint32_t* candcnt  = gn_sp_lds + (size_t)rpb * wpr * IMG + (size_t)nwaves * (128 * HFP + 2 * GN_SPLIT_STAGE);
    uint32_t* premax   = candcnt + nwaves; // largest count each wave reported (pre_mode)

    const uint32_t wi      = slice * 64 * LW + lane * LW; // first word of this lane in the row
    const bool     col_act = wi < p.W;
    const uint32_t wi_ld   = col_act ? wi : 0u;

    // bins-per-target bytes of my bins, in the layout of the byte counters (register r = (d*4+j)*2+pp, byte y <-> bit
    // 8y + 4pp + j of dword d)
    uint32_t nbreg[8 * ND];
#pragma unroll
    for (int r = 0; r < 8 * ND; ++r)
        nbreg[r] = RS ? 0u : p.sl_nbr[((size_t)slice * 8 * ND + r) * 64 + lane]; // (RS: every bin counts as one of four, see the prefilter)

    // Run select (p.run_select: every bin has a target, targets own consecutive bins in target order, none more than four, not all the
    // same number): which of my bins CONTINUE the target of the bin before (ns), and in one word (tbx) the first target that ends among my
    // bins, what it has before my first one (bits 28-30, below) and whether the bin after my last continues (bit 31).
    // Static for the persistent wave, like nbreg.
    uint32_t ns[ND], tbx = 0;
#pragma unroll
    for (int d = 0; d < ND; ++d)
        ns[d] = 0;
    if (RUNSEL && col_act)
    {
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        const uint32_t     b0   = wi * 64u;
        uint32_t           prev = b0 && b0 - 1 < p.B ? p.bin_tgt[b0 - 1] : NONE;
        uint32_t           cur  = b0 < p.B ? p.bin_tgt[b0] : NONE;
        uint32_t           first_end = NONE;
#pragma unroll
        for (int d = 0; d < ND; ++d)
        {
            uint32_t m = 0;
            for (uint32_t i = 0; i < 32u; ++i)
            {
                const uint32_t b   = b0 + 32u * (uint32_t)d + i;
                const uint32_t nxt = b + 1 < p.B ? p.bin_tgt[b + 1] : NONE;
                if (cur != NONE && cur == prev)
                    m |= 1u << i;
                if (cur != NONE && nxt != cur && first_end == NONE)
                    first_end = cur;
                prev = cur;
                cur  = nxt;
            }
            ns[d] = m;
        }
        // What the target that reaches into the lane brings along: up to three bins (bits 29-30), or "more than that" (bit 28: three
        // bins or more lie before mine and it is one of the list -- targets with more than GN_CAND_NBIG bins are not judged here).
        uint32_t back = 0, bigc = 0;
        if (ns[0] & 1u)
        {
            const uint32_t t = p.bin_tgt[b0];
            back             = b0 - p.tgt_off[t];
            bigc             = p.tgt_off[t + 1] - p.tgt_off[t] > GN_CAND_NBIG ? 1u : 0u;
            back             = bigc ? 0u : back;
        }
        tbx = (first_end == NONE ? 0u : first_end) | (bigc << 28) | (back << 29) | (cur != NONE && cur == prev ? 0x80000000u : 0u);
    }

    const uint32_t n_work = p.work_list ? (uint32_t)*p.work_count : p.n_reads - p.read_begin;
    unsigned long long chunk_base = 0;
    uint32_t           chunk_left = 0, chunk_size = GN_SPLIT_CHUNK; // (doubles per request up to GN_SPLIT_CHUNK_MAX, see gn_kernels.hip)
    unsigned long long n_pre = 0; // (lane 0) matches left unwritten for the filter_matches pre-pass, see below
    for (uint32_t round0 = blockIdx.x * rpb; round0 < n_work; round0 += gridDim.x * rpb)
    {
        const uint32_t widx = round0 + rslot;
        const uint32_t read = widx < n_work ? (p.work_list ? p.work_list[widx] : p.read_begin + widx) : 0xFFFFFFFFu;
        uint32_t       n    = 0;
        if (read < p.n_reads && p.status[read] == GN_READ_OK)
            n = p.n_hashes[read];
        if (n > NMAX) // the generic kernel takes it
        {
            if (slice == 0 && lane == 0)
                p.work_list_out[atomicAdd(p.work_count_out, 1ULL)] = read;
            n = 0;
        }
        const uint64_t* hs = p.hashes + (n ? p.slot_off[read] : 0);

        // ---- row table of the read, then one hash per iteration ----
        gn_sp_wave_sync();
        for (uint32_t idx = lane; idx < n * HF; idx += GN_WAVE)
        {
            const uint32_t q = idx / HF, i = idx - q * HF;
            rowtab[q * HFP + i] = gn_sp_row(hs[q], i, p.shift, p.S);
        }
        gn_sp_wave_sync();

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
        uint32_t acc_n = 0;
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
        auto issue = [&](uint32_t q, GnSpRows<HF, LW>& R) { // unconditional loads (see the fast kernel)
            q = q < n ? q : (n ? n - 1 : 0u);
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
        auto consume = [&](const GnSpRows<HF, LW>& R, uint32_t q) {
            const uint32_t on = (col_act && q < n) ? 0xFFFFFFFFu : 0u;
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
        if (n) // wave-uniform
        {
            GnSpRows<HF, LW> A, Bq;
            uint32_t         it = 0;
            issue(0, A);
            for (; it + 2 < n; it += 2)
            {
                issue(it + 1, Bq);
                consume(A, it);
                issue(it + 2, A);
                consume(Bq, it + 1);
            }
            const bool two = it + 1 < n;
            if (two)
                issue(it + 1, Bq);
            consume(A, it);
            if (two)
                consume(Bq, it + 1);
            if (acc_n)
                spill_nibbles();
        }

        // threshold_cutoff = max(1, ceil(n * rel_cutoff)) in IEEE double (:492-495,720-724)
        uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
        if (T == 0)
            T = 1;

        // ---- 1. prefilter in registers: bins with count * nb >= T ----
        uint32_t sm[ND];
        uint32_t mine = 0;
        {
            typedef unsigned short gn_u16x2 __attribute__((ext_vector_type(2)));
            const gn_u16x2 tm1 = __builtin_bit_cast(gn_u16x2, (T - 1) * 0x00010001u);
            const uint32_t tq  = (T + p.const_nb - 1u) >> (p.const_nb == 2u ? 1 : 2); // (RS: every bin is one of const_nb, 2 or 4)
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                    {
                        const uint32_t x  = byt[d][j][pp];
                        if constexpr (RS) // count * nb >= T  <=>  count >= ceil(T / nb), bytes compared in place (counts <= 127: no carry)
                        {
                            m |= (((x + (0x80u - tq) * 0x01010101u) & 0x80808080u) >> 7) << (4 * pp + j);
                            continue;
                        }
                        const uint32_t nb = nbreg[(d * 4 + j) * 2 + pp];
                        // bytes (0,1) and (2,3) as u16 pairs: count * nb <= 127 * 255
                        const gn_u16x2 lo = __builtin_bit_cast(gn_u16x2, __builtin_amdgcn_perm(0u, x, 0x0C010C00u))
                                            * __builtin_bit_cast(gn_u16x2, __builtin_amdgcn_perm(0u, nb, 0x0C010C00u));
                        const gn_u16x2 hi = __builtin_bit_cast(gn_u16x2, __builtin_amdgcn_perm(0u, x, 0x0C030C02u))
                                            * __builtin_bit_cast(gn_u16x2, __builtin_amdgcn_perm(0u, nb, 0x0C030C02u));
                        const uint32_t gl_ = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(lo, tm1));
                        const uint32_t gh_ = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(hi, tm1));
                        const uint32_t g   = ((gl_ & 0xFFFFu) ? 1u : 0u) | ((gl_ >> 16) ? 0x100u : 0u)
                                           | ((gh_ & 0xFFFFu) ? 0x10000u : 0u) | ((gh_ >> 16) ? 0x1000000u : 0u); // bit 8y
                        m |= g << (4 * pp + j);
                    }
                sm[d] = n ? m : 0u;
                mine += (uint32_t)__popc(sm[d]);
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            mine += __shfl_xor(mine, off);

        // ---- 2. the waves of a read agree: anything to do, and which select ----
        // (pre_mode, see step 3) the largest count of a single bin, by binary search with SWAR compares: a lower bound of the
        // read's largest target sum that is known before any target is summed
        uint32_t binmax = 0;
        if (p.pre_mode && n)
        {
            uint32_t lo = 0, hi = n;
            while (lo < hi)
            {
                const uint32_t mid = (lo + hi + 1) >> 1;
                const uint32_t K   = (0x80u - mid) * 0x01010101u; // counts <= n <= 127: no carry between bytes
                uint32_t       a   = 0;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            a |= byt[d][j][pp] + K;
                if (__ballot((a & 0x80808080u) != 0))
                    lo = mid;
                else
                    hi = mid - 1;
            }
            binmax = lo;
        }
        if (lane == 0)
        {
            candcnt[wave] = mine;
            premax[wave]  = binmax;
        }
        __syncthreads();
        uint32_t c_read = 0, Tpre = T;
        for (uint32_t sl = 0; sl < wpr; ++sl)
            c_read += candcnt[rslot * wpr + sl];
        if (p.pre_mode)
        {
            uint32_t mb = 0;
            for (uint32_t sl = 0; sl < wpr; ++sl)
                mb = premax[rslot * wpr + sl] > mb ? premax[rslot * wpr + sl] : mb;
            const uint32_t lb = p.pre_mode == 1 ? T : 0u;
            if (mb > lb)
            {
                const uint32_t t = gn_pf_threshold(mb, lb, p.pre_rel);
                Tpre = t > T ? t : T;
            }
        }
        const bool work     = n != 0 && (c_read != 0 || p.n_big != 0);
        const bool scan_all = c_read > GN_SPLIT_LIMIT * wpr;

        // ---- 3. byte image, then the select ----
        if (work)
        {
            // The image is in BIN ORDER: byte b of the read's image = count of bin b.  (Round 2 stored the registers as they
            // were -- register-major, lane-minor: neighbouring bins lay 256 or 512 bytes apart, i.e. in the SAME bank, and a
            // wave summing the bins of 64 neighbouring targets took 32-64 bank conflicts per read instruction; that, not
            // the arithmetic, was the 110 ms the select cost at low cutoffs.)  Byte y of register (d, j, pp) counts bin
            // 32d + 8y + 4pp + j of the lane: dword k of group d takes byte y = k/2 of the four registers (d, 0..3, k%2).
            uint32_t* nat = img_read + (size_t)wi * 16; // 16 dwords per 64-bin word
#pragma unroll
            for (int d = 0; d < ND; ++d)
            {
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                {
                    const uint32_t y = (uint32_t)k >> 1, pp = (uint32_t)k & 1u;
                    const uint32_t sel = 0x0C0C0000u | ((4u + y) << 8) | y; // {lo.byte y, hi.byte y, 0, 0}
                    const uint32_t t01 = __builtin_amdgcn_perm(byt[d][1][pp], byt[d][0][pp], sel);
                    const uint32_t t23 = __builtin_amdgcn_perm(byt[d][3][pp], byt[d][2][pp], sel);
                    o[k] = t01 | (t23 << 16);
                }
                *reinterpret_cast<uint4*>(nat + 8 * d)     = make_uint4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<uint4*>(nat + 8 * d + 4) = make_uint4(o[4], o[5], o[6], o[7]);
            }
        }
        __syncthreads();

        // Tsel: what a target's sum must reach to be reported (>= T; above it with a pre-pass to follow: targets in [T, Tsel) are
        // counted in drop1, their smallest sum kept in mnd, while `counting`); mxl / mne: largest / smallest sum reported
        uint32_t total = 0, mxl = 0, mne = 0xFFFFFFFFu, mnd = 0xFFFFFFFFu, drop1 = 0, Tsel = Tpre;
        bool     counting = true;
        unsigned long long base  = 0;
        {
            const uint8_t* img_bytes = reinterpret_cast<const uint8_t*>(img_read);
            const uint32_t img_dwords = wpr * IMG; // dwords of the read's image
            auto cnt_of = [&](uint32_t bin) -> uint32_t { return img_bytes[bin]; };
            auto target_sum = [&](const uint4& rec) -> uint32_t { // {first CSR entry, bins, ., .}
                uint32_t s = 0;
                for (uint32_t x = 0; x < rec.y; ++x)
                    s += cnt_of(p.tgt_bins[rec.x + x]);
                return s > n ? n : s; // :525-526
            };
            auto judge = [&](uint32_t cv) -> bool {
                if (cv >= Tsel)
                    return true;
                if (counting && cv >= T)
                {
                    ++drop1;
                    mnd = cv < mnd ? cv : mnd;
                }
                return false;
            };
            auto emit_hits = [&](bool emit, uint32_t tgt, uint32_t cv, uint32_t& tot, bool direct, gn_match* out) {
                const uint64_t bm = __ballot(emit);
                if (emit)
                {
                    mxl = cv > mxl ? cv : mxl;
                    mne = cv < mne ? cv : mne;
                    const uint32_t o = tot + __popcll(bm & ((1ULL << lane) - 1ULL));
                    if (direct)
                    {
                        gn_match mt;
                        mt.read   = read;
                        mt.target = tgt;
                        mt.count  = cv;
                        out[o]    = mt;
                    }
                    else if (o < GN_SPLIT_STAGE)
                    {
                        stage[2 * o]     = tgt;
                        stage[2 * o + 1] = cv;
                    }
                }
                tot += (uint32_t)__popcll(bm);
            };
            // `direct` = second pass of a (read, slice) with more hits than the staging list holds
            auto select = [&](bool direct, gn_match* out) -> uint32_t {
                uint32_t tot = 0;
                if (RS != 1 && scan_all && p.uniform_nb)
                {
                    // Every target owns the same power-of-two number of consecutive bins (2 or 4) in target order: a target is
                    // half a dword (or a dword) of the lane's own bin-ordered counters, so the lane judges its 64*LW/nb targets
                    // with packed 16-bit arithmetic -- sum, cap at n, the two compares, the tallies of what stays under the bar
                    // -- ~5 instructions a target instead of ~60 (the scan below), and only targets at or above the bar are
                    // looked at one by one.  Each wave takes the targets of its own column slice.
                    typedef unsigned short gn_u16x2 __attribute__((ext_vector_type(2)));
                    typedef short          gn_i16x2 __attribute__((ext_vector_type(2)));
                    const bool     two  = p.uniform_nb == 2;
                    const uint32_t tb   = (wi * 64u) >> (two ? 1 : 2); // first target of this lane
                    const gn_u16x2 nv   = __builtin_bit_cast(gn_u16x2, n * 0x00010001u);
                    const gn_u16x2 selv = __builtin_bit_cast(gn_u16x2, (Tsel > 0xFFFFu ? 0xFFFFu : Tsel) * 0x00010001u);
                    const gn_u16x2 tv   = __builtin_bit_cast(gn_u16x2, T * 0x00010001u);
                    gn_u16x2       mnd2 = __builtin_bit_cast(gn_u16x2, 0xFFFFFFFFu);
                    uint32_t       hit[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                    {
                        uint32_t hm = 0;
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                        {
                            const uint32_t y = (uint32_t)k >> 1, pp = (uint32_t)k & 1u;
                            const uint32_t sel = 0x0C0C0000u | ((4u + y) << 8) | y;
                            const uint32_t x   = __builtin_amdgcn_perm(byt[d][1][pp], byt[d][0][pp], sel)
                                               | (__builtin_amdgcn_perm(byt[d][3][pp], byt[d][2][pp], sel) << 16); // bins 4k .. 4k+3
                            uint32_t s2 = (x & 0x00FF00FFu) + ((x >> 8) & 0x00FF00FFu); // {bins 4k + 4k+1, bins 4k+2 + 4k+3}
                            if (!two)
                                s2 = (s2 & 0xFFFFu) + (s2 >> 16); // one target per dword (upper half: 0, never reaches T >= 1)
                            const gn_u16x2 sv  = __builtin_elementwise_min(__builtin_bit_cast(gn_u16x2, s2), nv); // :525-526
                            const gn_i16x2 pm  = sv >= selv;
                            const gn_i16x2 mid = counting ? (gn_i16x2)((sv >= tv) & ~pm) : (gn_i16x2)(0);
                            const uint32_t pmb = __builtin_bit_cast(uint32_t, pm), midb = __builtin_bit_cast(uint32_t, mid);
                            drop1 += (uint32_t)__popc(midb) >> 4;
                            mnd2 = __builtin_elementwise_min(mnd2, __builtin_bit_cast(gn_u16x2, (__builtin_bit_cast(uint32_t, sv) & midb) | ~midb));
                            hm |= ((pmb & 1u) | ((pmb >> 15) & 2u)) << (2 * k);
                        }
                        hit[d] = col_act ? hm : 0u;
                    }
                    {
                        const uint32_t m2 = __builtin_bit_cast(uint32_t, mnd2), lo16 = m2 & 0xFFFFu, hi16 = m2 >> 16;
                        const uint32_t mm = lo16 < hi16 ? lo16 : hi16;
                        if (mm != 0xFFFFu)
                            mnd = mm < mnd ? mm : mnd;
                    }
                    // emission in target order: lane after lane (a lane's targets are consecutive), each lane at the offset the
                    // wave's prefix sum of hit counts gives it -- the grouping pass (gn_gather_kernel) wants ascending targets
                    uint32_t lane_o = (uint32_t)lane; // (not kept across the reads: the shuffle addresses end in scratch otherwise)
                    asm volatile("" : "+v"(lane_o));
                    uint32_t mine_hits = 0;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        mine_hits += (uint32_t)__popc(hit[d]);
                    uint32_t inc = mine_hits;
#pragma unroll
                    for (int off = 1; off < GN_WAVE; off <<= 1)
                    {
                        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane_o - (uint32_t)off) & 63u) << 2), (int)inc);
                        inc += lane_o >= (uint32_t)off ? y : 0u;
                    }
                    uint32_t       o      = inc - mine_hits;
                    const uint32_t in_all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        while (hit[d])
                        {
                            const uint32_t tp = 16u * d + (uint32_t)__builtin_ctz(hit[d]); // half-dword index inside the lane
                            hit[d] &= hit[d] - 1;
                            const uint32_t b0 = wi * 64u + 2u * tp;
                            uint32_t       cv = cnt_of(b0) + cnt_of(b0 + 1);
                            if (!two)
                                cv += cnt_of(b0 + 2) + cnt_of(b0 + 3);
                            cv = cv > n ? n : cv;
                            const uint32_t tgt = tb + (two ? tp : tp >> 1);
                            mxl = cv > mxl ? cv : mxl;
                            mne = cv < mne ? cv : mne;
                            if (direct)
                            {
                                gn_match mt;
                                mt.read   = read;
                                mt.target = tgt;
                                mt.count  = cv;
                                out[o]    = mt;
                            }
                            else if (o < GN_SPLIT_STAGE)
                            {
                                stage[2 * o]     = tgt;
                                stage[2 * o + 1] = cv;
                            }
                            ++o;
                        }
                    tot += in_all;
                    return tot;
                }
                if (RUNSEL && scan_all)
                {
                    // Targets of one to four consecutive bins, widths mixed: a lane judges the targets that END among its own bins from
                    // its own registers -- a running sum over its bins in bin order that starts again where a target starts (ns), looked
                    // at where a target ends -- ~9 instructions a bin and no memory access, where the scan below spends ~60 a
                    // target on offsets, LDS gathers and shifts.  The smallest sum under the bar comes out of the same loop; only
                    // targets at or above the bar are looked at one by one afterwards (from the image, no global loads either).
                    // (the masks are static for the wave: without the barrier the compiler keeps all the per-bin masks derived from them
                    // in registers across the reads)
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        asm volatile("" : "+v"(ns[d]));
                    // (likewise what follows from the lane's number and its word: recomputed here, a few instructions, instead of being
                    // kept across the reads -- beside the counters such values end up in scratch and are read back before every use)
                    uint32_t tbx_o = tbx, lane_o = (uint32_t)lane, wi_o = wi;
                    asm volatile("" : "+v"(tbx_o), "+v"(lane_o), "+v"(wi_o));
                    // targets with more than GN_CAND_NBIG bins first: this wave's share of the list, summed from the image
                    {
                        const uint32_t per = (p.n_big + wpr - 1) / wpr;
                        const uint32_t lo = min(p.n_big, slice * per), hi = min(p.n_big, lo + per);
                        for (uint32_t i0 = lo; i0 < hi; i0 += GN_WAVE)
                        {
                            const uint32_t i    = i0 + lane;
                            bool           emit = false;
                            uint32_t       tgt = 0, cv = 0;
                            if (i < hi)
                            {
                                tgt  = p.big_list[i];
                                cv   = target_sum(p.tgt_rec[tgt]);
                                emit = judge(cv);
                            }
                            emit_hits(emit, tgt, cv, tot, direct, out);
                        }
                    }
                    const uint32_t tb = tbx_o & 0x0FFFFFFFu, back = (tbx_o >> 29) & 3u;
                    // "continues" of the three bins before mine (all three for a target of the list that reaches into the lane)
                    const uint32_t backbits = (tbx_o & 0x10000000u) ? 0xE0000000u : (back >= 2u ? 0x80000000u : 0u) | (back >= 3u ? 0x40000000u : 0u);
                    uint32_t       run      = 0;
                    for (uint32_t q = 1; q <= back; ++q) // the bins my first target has in the lane (or slice) before me
                        run += cnt_of(wi_o * 64u - q);
                    // min(sum, n) >= bar (:525-526)  <=>  bar <= n and sum >= bar; sums stay below 1024, so a bar of 2^16 never passes
                    const uint32_t selb = Tsel <= n ? Tsel : 0x10000u, tbar = T <= n ? T : 0x10000u;
                    uint32_t       hit[ND], low = 0xFFFFFFFFu, n_mid = 0;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                    {
                        // a bin ends its target where the next bin does not continue it (every bin below B has a target)
                        const uint32_t b0d   = wi_o * 64u + 32u * (uint32_t)d;
                        const uint32_t valid = !col_act || b0d >= p.B ? 0u : (p.B - b0d >= 32u ? 0xFFFFFFFFu : (1u << (p.B - b0d)) - 1u);
                        const uint32_t nextc = d + 1 < ND ? ns[d + 1 < ND ? d + 1 : 0] << 31 : tbx_o & 0x80000000u;
                        // ... and a bin that continues three bins in a row is the fifth of its target, or later: where such a bin ends a
                        // target, the target is one of the list and is not judged here
                        const uint32_t below = d ? ns[d > 0 ? d - 1 : 0] : backbits;
                        const uint32_t fifth = ns[d] & __builtin_amdgcn_alignbit(ns[d], below, 31) & __builtin_amdgcn_alignbit(ns[d], below, 30)
                                             & __builtin_amdgcn_alignbit(ns[d], below, 29);
                        const uint32_t en    = ~((ns[d] >> 1) | nextc) & valid & ~fifth;
                        const uint32_t nen   = ~en;
                        uint32_t       hm = 0, tm = 0; // sign bits shifted in from below: bin 0 ends up on top, 1 = under the bar
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                        {
                            const uint32_t y = (uint32_t)k >> 1, pp = (uint32_t)k & 1u;
                            const uint32_t sel = 0x0C0C0000u | ((4u + y) << 8) | y;
                            const uint32_t x   = __builtin_amdgcn_perm(byt[d][1][pp], byt[d][0][pp], sel)
                                               | (__builtin_amdgcn_perm(byt[d][3][pp], byt[d][2][pp], sel) << 16); // bins 4k .. 4k+3
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                            {
                                const int      pos = 4 * k + j;
                                const uint32_t c   = (x >> (8 * j)) & 0xFFu;
                                run                = (run & (uint32_t)((int32_t)(ns[d] << (31 - pos)) >> 31)) + c; // starts again where a target starts
                                const uint32_t xs  = run - selb, xt = run - tbar;                                    // sign set: under the bar
                                hm                 = __builtin_amdgcn_alignbit(hm, xs, 31);
                                tm                 = __builtin_amdgcn_alignbit(tm, xt, 31);
                                const uint32_t v   = xt | (uint32_t)((int32_t)(nen << (31 - pos)) >> 31); // all ones unless a target ends here
                                low                = v < low ? v : low;
                            }
                            __builtin_amdgcn_sched_barrier(0); // (one chain of dependent adds: nothing to gain from looking ahead, registers to lose)
                        }
                        hm = ~__builtin_bitreverse32(hm);
                        tm = ~__builtin_bitreverse32(tm);
                        hit[d] = hm & en;
                        n_mid += (uint32_t)__popc(tm & ~hm & en);
                    }
                    if (counting && n_mid) // (the smallest sum that reached T is one under the bar: they all lie below the hits)
                    {
                        const uint32_t cv = low + tbar > n ? n : low + tbar;
                        drop1 += n_mid;
                        mnd = cv < mnd ? cv : mnd;
                    }
                    uint32_t mine_hits = 0;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        mine_hits += (uint32_t)__popc(hit[d]);
                    uint32_t inc = mine_hits;
#pragma unroll
                    for (int off = 1; off < GN_WAVE; off <<= 1)
                    {
                        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane_o - (uint32_t)off) & 63u) << 2), (int)inc);
                        inc += lane_o >= (uint32_t)off ? y : 0u;
                    }
                    uint32_t       o      = tot + inc - mine_hits;
                    const uint32_t in_all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    uint32_t       ended  = 0; // targets that end in my groups before d
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                    {
                        const uint32_t b0d   = wi_o * 64u + 32u * (uint32_t)d;
                        const uint32_t valid = !col_act || b0d >= p.B ? 0u : (p.B - b0d >= 32u ? 0xFFFFFFFFu : (1u << (p.B - b0d)) - 1u);
                        const uint32_t nextc = d + 1 < ND ? ns[d + 1 < ND ? d + 1 : 0] << 31 : tbx_o & 0x80000000u;
                        const uint32_t en    = ~((ns[d] >> 1) | nextc) & valid; // (every target's end: the numbering counts them all)
                        const uint64_t flags = ((uint64_t)ns[d] << 32) | (d ? ns[d > 0 ? d - 1 : 0] : backbits);
                        while (hit[d])
                        {
                            const uint32_t pos = (uint32_t)__builtin_ctz(hit[d]);
                            hit[d] &= hit[d] - 1;
                            const uint32_t f3 = (uint32_t)(flags >> (30u + pos)) & 7u; // "continues" of the bins pos, pos-1, pos-2
                            const uint32_t nb = f3 == 7u ? 4u : (f3 >= 6u ? 3u : (f3 >= 4u ? 2u : 1u));
                            const uint32_t b  = b0d + pos;
                            uint32_t       cv = 0;
                            for (uint32_t q = 0; q < nb; ++q)
                                cv += cnt_of(b - q);
                            cv = cv > n ? n : cv;
                            const uint32_t tgt = tb + ended + (uint32_t)__popc(en & ((1u << pos) - 1u));
                            mxl = cv > mxl ? cv : mxl;
                            mne = cv < mne ? cv : mne;
                            if (direct)
                            {
                                gn_match mt;
                                mt.read   = read;
                                mt.target = tgt;
                                mt.count  = cv;
                                out[o]    = mt;
                            }
                            else if (o < GN_SPLIT_STAGE)
                            {
                                stage[2 * o]     = tgt;
                                stage[2 * o + 1] = cv;
                            }
                            ++o;
                        }
                        ended += (uint32_t)__popc(en);
                    }
                    tot += in_all;
                    return tot;
                }
                if (RS == 0 && scan_all)
                {
                    // too many candidates (tiny T, dense hits): every target, this wave takes its share
                    const uint32_t per = (p.n_targets + wpr - 1) / wpr;
                    const uint32_t lo = min(p.n_targets, slice * per), hi = min(p.n_targets, lo + per);
                    if (p.csr_identity)
                    {
                        // targets own consecutive bins in target order (what ganon-build writes): the bins of target t are
                        // [tgt_off[t], tgt_off[t+1]) -- two coalesced loads, no record, no bin list
                        uint32_t o0[4], o1[4];
                        auto     load_off = [&](uint32_t tb, uint32_t (&a)[4], uint32_t (&b)[4]) {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                            {
                                const uint32_t t = tb + 64u * (uint32_t)u + lane;
                                a[u] = t < hi ? p.tgt_off[t] : 0u;
                                b[u] = t < hi ? p.tgt_off[t + 1] : 0u;
                            }
                        };
                        load_off(lo, o0, o1);
                        for (uint32_t t0 = lo; t0 < hi; t0 += 4 * GN_WAVE)
                        {
                            uint32_t n0[4], n1[4], cv[4];
                            load_off(t0 + 4 * GN_WAVE, n0, n1); // the next trip's offsets fly while this trip's bins are summed
                            // a target of up to four bins is one pair of dwords of the bin-ordered image, shifted to its first
                            // bin and masked to its width: no loop whose trip count depends on the data, so the four targets of
                            // a lane are in flight together (targets with more bins: the loop below, rarely taken)
                            uint32_t x[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                            {
                                const uint32_t w0 = o0[u] >> 2;
                                const uint32_t lo_dw = img_read[w0], hi_dw = img_read[w0 + 1 < img_dwords ? w0 + 1 : w0];
                                x[u] = __builtin_amdgcn_alignbyte(hi_dw, lo_dw, o0[u] & 3u);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                            {
                                const uint32_t nb = o1[u] - o0[u];
                                const uint32_t m  = nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
                                const uint32_t y  = x[u] & m;
                                uint32_t       sum = (y & 0x00FF00FFu) + ((y >> 8) & 0x00FF00FFu);
                                sum = (sum & 0xFFFFu) + (sum >> 16);
                                for (uint32_t b = o0[u] + 4u; b < o1[u]; ++b)
                                    sum += cnt_of(b);
                                cv[u] = sum > n ? n : sum; // :525-526
                            }
                            // judged without a branch; the emission (a ballot, a prefix count and a store per group of 64 targets)
                            // only where some lane of the wave has a target at or above the bar -- at low cutoffs a fifth of the
                            // targets reach T, a few per cent reach the bar the pre-pass allows (this loop is VALU bound:
                            // 47.5 G vector instructions per 2 M reads before, SQ counters of scripts/split_lowcut.py)
                            bool pass[4], any = false;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                            {
                                const uint32_t t  = t0 + 64u * (uint32_t)u + lane;
                                const bool     in = t < hi;
                                pass[u]           = in && cv[u] >= Tsel;
                                const bool mid    = in && counting && cv[u] >= T && cv[u] < Tsel;
                                drop1 += mid ? 1u : 0u;
                                mnd = mid && cv[u] < mnd ? cv[u] : mnd;
                                any = any || pass[u];
                            }
                            if (__ballot(any))
                            {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (t0 + 64u * (uint32_t)u < hi) // (wave-uniform: emit_hits holds a ballot)
                                        emit_hits(pass[u], t0 + 64u * (uint32_t)u + lane, cv[u], tot, direct, out);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                            {
                                o0[u] = n0[u];
                                o1[u] = n1[u];
                            }
                        }
                        return tot;
                    }
                    // four targets per lane and trip: their records, then their first two bins, are loaded side by side (the chain
                    // record -> bin list -> byte image is all latency; one target at a time it was most of the kernel at low cutoffs)
                    for (uint32_t t0 = lo; t0 < hi; t0 += 4 * GN_WAVE)
                    {
                        uint4    rec[4];
                        uint32_t b0[4], b1[4], cv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                        {
                            const uint32_t t = t0 + 64u * (uint32_t)u + lane;
                            rec[u] = t < hi ? p.tgt_rec[t] : make_uint4(0u, 0u, 0u, 0u);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                        {
                            b0[u] = rec[u].y >= 1 ? p.tgt_bins[rec[u].x] : 0u;
                            b1[u] = rec[u].y >= 2 ? p.tgt_bins[rec[u].x + 1] : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                        {
                            uint32_t sum = 0;
                            if (rec[u].y >= 1)
                                sum = cnt_of(b0[u]);
                            if (rec[u].y >= 2)
                                sum += cnt_of(b1[u]);
                            for (uint32_t x = 2; x < rec[u].y; ++x)
                                sum += cnt_of(p.tgt_bins[rec[u].x + x]);
                            cv[u] = sum > n ? n : sum; // :525-526
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                        {
                            const uint32_t t = t0 + 64u * (uint32_t)u + lane;
                            if (t0 + 64u * (uint32_t)u < hi) // (wave-uniform: emit_hits holds a ballot)
                                emit_hits(t < hi && judge(cv[u]), t, cv[u], tot, direct, out);
                        }
                    }
                    return tot;
                }
                // targets with more than GN_CAND_NBIG bins: this wave's share of the list
                {
                    const uint32_t per = (p.n_big + wpr - 1) / wpr;
                    const uint32_t lo = min(p.n_big, slice * per), hi = min(p.n_big, lo + per);
                    for (uint32_t i0 = lo; i0 < hi; i0 += GN_WAVE)
                    {
                        const uint32_t i    = i0 + lane;
                        bool           emit = false;
                        uint32_t       tgt = 0, cv = 0;
                        if (i < hi)
                        {
                            tgt  = p.big_list[i];
                            cv   = target_sum(p.tgt_rec[tgt]);
                            emit = judge(cv);
                        }
                        emit_hits(emit, tgt, cv, tot, direct, out);
                    }
                }
                // candidate bins, one per lane and trip; a target is reported by its lowest candidate bin
                uint32_t c[ND];
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    c[d] = sm[d];
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
                        bool           lowest = !(RS == 1 && rec.y > GN_CAND_NBIG); // (RS: bins of the list's targets pass the prefilter too)
                        for (uint32_t x = 0; lowest && x < rec.y; ++x) // bins of a target ascend in the CSR
                        {
                            const uint32_t bx = p.tgt_bins[rec.x + x];
                            if (bx >= b)
                                break;
                            if (cnt_of(bx) * (RS ? p.const_nb : rec.y) >= T) // (the very rule of the prefilter: who is a candidate)
                            {
                                lowest = false;
                                break;
                            }
                        }
                        if (lowest)
                        {
                            cv   = target_sum(rec);
                            emit = judge(cv);
                            tgt  = t;
                        }
                    }
                    emit_hits(emit, tgt, cv, tot, direct, out);
                }
                return tot;
            };

            if (p.pre_mode && p.max_first && (RS == 1 || p.uniform_nb)) // (the same for every wave of the launch: the barrier below is met by all)
                {
                    // The read's true maximum before any target is judged -- where targets are runs of bins a lane sees in its own
                    // registers.  The bar it allows goes into the first select; without it a read with many targets between the two
                    // bars is scanned up to three times (bar of the largest bin, bar of the maximum, direct output).
                    // RS: the largest running sum over the bins (a partial sum never exceeds the sum of its target, which some lane
                    // sees whole), 4 instructions a bin.
                    uint32_t mx = 0;
                    if (RS != 1 && p.uniform_nb && work && scan_all && col_act)
                    {
                        // uniform two- or four-bin targets: the sums as u16 halves, as the packed select forms them
                        typedef unsigned short gn_u16x2 __attribute__((ext_vector_type(2)));
                        const bool two = p.uniform_nb == 2;
                        gn_u16x2   m2  = __builtin_bit_cast(gn_u16x2, 0u);
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                            {
                                const uint32_t y = (uint32_t)k >> 1, pp = (uint32_t)k & 1u;
                                const uint32_t sel = 0x0C0C0000u | ((4u + y) << 8) | y;
                                const uint32_t x   = __builtin_amdgcn_perm(byt[d][1][pp], byt[d][0][pp], sel)
                                                   | (__builtin_amdgcn_perm(byt[d][3][pp], byt[d][2][pp], sel) << 16); // bins 4k .. 4k+3
                                uint32_t s2 = (x & 0x00FF00FFu) + ((x >> 8) & 0x00FF00FFu);
                                if (!two)
                                    s2 = (s2 & 0xFFFFu) + (s2 >> 16);
                                m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(gn_u16x2, s2));
                            }
                        const uint32_t mm = __builtin_bit_cast(uint32_t, m2);
                        mx = (mm & 0xFFFFu) > (mm >> 16) ? (mm & 0xFFFFu) : (mm >> 16);
                    }
                    if (RUNSEL && work && scan_all && col_act)
                    {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
                            asm volatile("" : "+v"(ns[d]));
                        uint32_t tbx_o = tbx, wi_o = wi; // (not kept across the reads, cf. the select)
                        asm volatile("" : "+v"(tbx_o), "+v"(wi_o));
                        const uint32_t back = (tbx_o >> 29) & 3u;
                        uint32_t       run  = 0;
                        for (uint32_t q = 1; q <= back; ++q)
                            run += cnt_of(wi_o * 64u - q);
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                            {
                                const uint32_t y = (uint32_t)k >> 1, pp = (uint32_t)k & 1u;
                                const uint32_t sel = 0x0C0C0000u | ((4u + y) << 8) | y;
                                const uint32_t x   = __builtin_amdgcn_perm(byt[d][1][pp], byt[d][0][pp], sel)
                                                   | (__builtin_amdgcn_perm(byt[d][3][pp], byt[d][2][pp], sel) << 16); // bins 4k .. 4k+3
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                {
                                    const int pos = 4 * k + j;
                                    run = (run & (uint32_t)((int32_t)(ns[d] << (31 - pos)) >> 31)) + ((x >> (8 * j)) & 0xFFu);
                                    mx  = run > mx ? run : mx;
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                    for (int off = 32; off > 0; off >>= 1)
                    {
                        const uint32_t y = (uint32_t)__shfl_xor((int)mx, off);
                        mx = y > mx ? y : mx;
                    }
                    if (lane == 0)
                        candcnt[wave] = mx; // (the candidate counts were read before the image's barrier)
                    __syncthreads();
                    if (work && scan_all)
                    {
                        uint32_t mx_r = 0;
                        for (uint32_t sl = 0; sl < wpr; ++sl)
                            mx_r = candcnt[rslot * wpr + sl] > mx_r ? candcnt[rslot * wpr + sl] : mx_r;
                        mx_r = mx_r > n ? n : mx_r; // :525-526
                        const uint32_t t2 = mx_r ? gn_pf_threshold(mx_r, p.pre_mode == 1 ? T : 0u, p.pre_rel) : 0u;
                        if (t2 > Tsel && t2 <= mx_r)
                            Tsel = t2;
                    }
                }
            if (work)
                total = select(false, nullptr);
            if (p.pre_mode)
            {
                // A filter_matches pre-pass follows (gn_postfilter.hip).  The select above already left out targets under Tsel, the
                // bar that the largest single-bin count of the read allows (the --rel-filter threshold of the read cannot be
                // lower, see the fast kernel's epilogue in gn_kernels.hip).  Now the waves of the read share the largest sum any of
                // them reported -- the read's true maximum -- and what is under the bar t2 it allows leaves the staging list again.
                // Everything left out is counted and its smallest sum kept (the read's minimum is over all targets that reached T).
                for (int off = 32; off > 0; off >>= 1)
                {
                    const uint32_t y = (uint32_t)__shfl_xor((int)mxl, off);
                    mxl = y > mxl ? y : mxl;
                }
                if (lane == 0)
                    premax[wave] = mxl;
                __syncthreads();
                uint32_t mx_r = 0;
                for (uint32_t sl = 0; sl < wpr; ++sl)
                    mx_r = premax[rslot * wpr + sl] > mx_r ? premax[rslot * wpr + sl] : mx_r;
                counting = false;
                for (int off = 32; off > 0; off >>= 1)
                {
                    const uint32_t y = (uint32_t)__shfl_xor((int)mne, off), z = (uint32_t)__shfl_xor((int)mnd, off);
                    mne = y < mne ? y : mne;
                    mnd = z < mnd ? z : mnd;
                    drop1 += (uint32_t)__shfl_xor((int)drop1, off);
                }
                const uint32_t mn_all = mne < mnd ? mne : mnd; // smallest sum >= T of the unit, written or not
                uint32_t       dropped = drop1;
                const uint32_t t2 = mx_r ? gn_pf_threshold(mx_r, p.pre_mode == 1 ? T : 0u, p.pre_rel) : 0u;
                if (total && t2 > Tsel && t2 <= mx_r && mne < t2) // the true maximum raises the bar further
                {
                    uint32_t kept = 0;
                    if (total <= GN_SPLIT_STAGE)
                    {
                        gn_sp_wave_sync();
                        for (uint32_t o0 = 0; o0 < total; o0 += GN_WAVE)
                        {
                            const uint32_t o   = o0 + lane;
                            const bool     act = o < total;
                            uint32_t       tg = 0, cv = 0;
                            if (act)
                            {
                                tg = stage[2 * o];
                                cv = stage[2 * o + 1];
                            }
                            const bool     keep = act && cv >= t2;
                            const uint64_t bm   = __ballot(keep);
                            gn_sp_wave_sync(); // (every lane has read its entry: compaction in place)
                            if (keep)
                            {
                                const uint32_t q = kept + (uint32_t)__popcll(bm & ((1ULL << lane) - 1ULL));
                                stage[2 * q]     = tg;
                                stage[2 * q + 1] = cv;
                            }
                            kept += (uint32_t)__popcll(bm);
                        }
                    }
                    else // more hits than the list holds: counted again with the higher bar (and staged, if they fit now)
                    {
                        Tsel = t2;
                        kept = select(false, nullptr);
                    }
                    dropped += total - kept;
                    total = kept;
                }
                if (dropped && lane == 0)
                {
                    p.seg_min[(size_t)read * wpr + slice] = mn_all;
                    n_pre += dropped;
                }
            }
            if (total)
            {
                if (total > chunk_left)
                {
                    const uint32_t need = total > chunk_size ? total : chunk_size;
                    chunk_size = chunk_size < GN_SPLIT_CHUNK_MAX ? chunk_size * 2u : chunk_size;
                    unsigned long long nb = 0;
                    if (lane == 0)
                        nb = atomicAdd(p.cursor, (unsigned long long)need);
                    chunk_base = gn_sp_readlane64(nb, 0);
                    chunk_left = need;
                }
                base = chunk_base;
                chunk_base += total;
                chunk_left -= total;
                if (base + total <= p.match_cap)
                {
                    if (total <= GN_SPLIT_STAGE)
                    {
                        gn_sp_wave_sync();
                        for (uint32_t o = lane; o < total; o += GN_WAVE)
                        {
                            gn_match mt;
                            mt.read   = read;
                            mt.target = stage[2 * o];
                            mt.count  = stage[2 * o + 1];
                            p.matches[base + o] = mt;
                        }
                    }
                    else
                        (void)select(true, p.matches + base);
                }
            }
        }
        if (read < p.n_reads && lane == 0)
        {
            p.seg_begin[(size_t)read * wpr + slice] = base;
            p.seg_count[(size_t)read * wpr + slice] = total;
        }
        __syncthreads(); // the next round reuses the images, the hit lists and the candidate counters
    }
    if (p.pre_mode && lane == 0 && n_pre)
        atomicAdd(p.pre_ctr, n_pre);
}

size_t gn_split_lds_bytes(const GnCountGeometry& g, uint32_t hash_funs)
{
    const uint32_t nd = 2 * g.lw, hfp = hash_funs <= 4 ? 4 : 8, nwaves = g.block / 64;
    return ((size_t)g.rpb * g.wpr * 8 * nd * 64 + (size_t)nwaves * (128 * hfp + 2 * GN_SPLIT_STAGE) + 2 * nwaves) * 4;
}

template <int HF, int LW, int MAXT, int RS>
static hipError_t gn_launch_split_rs(const GnCountParams& p, const GnCountGeometry& g, hipStream_t st)
{
    uint32_t blocks = (p.n_reads - p.read_begin + g.rpb - 1) / g.rpb;
    if (blocks > p.max_blocks)
        blocks = p.max_blocks;
    const size_t lds = gn_split_lds_bytes(g, HF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gn_ibf_count_split_kernel<HF, LW, MAXT, RS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gn_ibf_count_split_kernel<HF, LW, MAXT, RS>), dim3(blocks), dim3(g.block), lds, st, p);
    return hipGetLastError();
}

template <int HF, int LW, int MAXT>
static hipError_t gn_launch_split_one(const GnCountParams& p, const GnCountGeometry& g, hipStream_t st)
{
    if constexpr (GN_SPLIT_RUN_SELECT(MAXT))
    {
        if (p.const_nb && p.uniform_nb)
            return gn_launch_split_rs<HF, LW, MAXT, 2>(p, g, st);
        if (p.const_nb)
            return gn_launch_split_rs<HF, LW, MAXT, 1>(p, g, st);
    }
    return gn_launch_split_rs<HF, LW, MAXT, 0>(p, g, st);
}

template <int HF>
static hipError_t gn_launch_split_hf(const GnCountParams& p, const GnCountGeometry& g, hipStream_t st)
{
    if (g.block <= 256)
        return g.lw == 2 ? gn_launch_split_one<HF, 2, 256>(p, g, st) : gn_launch_split_one<HF, 1, 256>(p, g, st);
    if (g.block <= 512)
        return g.lw == 2 ? gn_launch_split_one<HF, 2, 512>(p, g, st) : gn_launch_split_one<HF, 1, 512>(p, g, st);
    return g.lw == 2 ? gn_launch_split_one<HF, 2, 1024>(p, g, st) : gn_launch_split_one<HF, 1, 1024>(p, g, st);
}

hipError_t gn_launch_count_split(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st)
{
    if (p.n_reads <= p.read_begin)
        return hipSuccess;
    switch (hash_funs)
    {
        case 1: return gn_launch_split_hf<1>(p, g, st);
        case 2: return gn_launch_split_hf<2>(p, g, st);
        case 3: return gn_launch_split_hf<3>(p, g, st);
        case 4: return gn_launch_split_hf<4>(p, g, st);
        case 5: return gn_launch_split_hf<5>(p, g, st);
        default: return hipErrorInvalidValue;
    }
}
