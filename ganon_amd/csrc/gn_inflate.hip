// gn_inflate.hip -- a gzip file (RFC 1952 members of RFC 1951 DEFLATE data) inflated on the device.
//
// Where it sits: the reference reads `reads.fq.gz` through ONE zlib stream per file (seqan3::sequence_file_input over a gz
// stream, /root/reference/src/ganon-classify/GanonClassify.cpp:1220-1287; one decompression thread, :1433).  The host side of
// this repo already inflates such a file on several threads (host/pgzip.cpp: speculative block starts + 16-bit symbols with
// window markers, the scheme of pugz / rapidgzip); on the 16-core hosts of the GPU boxes that is what bounds a `.fq.gz` run
// (47 Mreads/s against 113-125 for plain FASTQ and 230 for the kernels alone).  Here the same scheme runs on the device, and
// the compressed bytes -- a quarter of the text -- are what crosses the link:
//
//   1. the compressed file is cut into CHUNKS of `chunk_bytes` (32 KiB).  One wave per chunk (gi_chunk_kernel):
//      * SEARCH for the chunk's start: a non-final dynamic-Huffman block header, or a gzip member header, at or behind the chunk's nominal
//        position.  Three screens, each over many positions at once: 256 bit positions a round take a 17-bit test (BFINAL = 0, BTYPE = 2,
//        HLIT / HDIST in range; on byte boundaries also 1f 8b 08); the survivors, 64 at a time, the Kraft sum of the code-length code;
//        what is left, 16+ at a time, a table-free decode of the whole header per lane (code lengths decode, literal/length code
//        complete with an end-of-block code, distance code legal).  What passes is tried in position order: its first block is decoded
//        in strict mode (printable text only: the inputs are FASTQ/FASTA), and the first position that passes is the chunk's start;
//      * DECODE from there to the first block (or member) boundary at or behind the next chunk's nominal position at which something
//        begins that the next chunk's search can find.  The Huffman tables live in LDS.  A ROUND decodes 64 bit positions at once -- lane
//        l the whole token that would begin at position l -- and follows the chain of real tokens with one v_readlane each; batches of
//        up to 64 tokens are placed by their lanes (output offsets by a prefix sum; the far sources of all matches in flight together).
//        Output = 16-bit symbols: a byte, or a MARKER 256 + i for "byte i of the 32 KiB before this chunk".  The last 1024 symbols live
//        in an LDS ring (nothing is read from global memory before the line that holds it has been written completely); older sources are
//        read back from the chunk's own output, which is allocated in pieces of 32 Ki symbols from a pool (no bound on a chunk's
//        expansion has to be guessed).  Two steps' decodes run ahead of the step that is being put together, on streams of their own;
//   2. ORDER (gi_order_kernel, one workgroup): a chunk that starts at a position lies in the slot of that position's nominal range, so
//      "which chunk continues this one" is a lookup for every slot at once; one thread follows the links from the stream's position; text
//      offsets, the work list, the members' lengths (ISIZE) and the list for the CRC pass come from prefix sums over the chain.  A
//      position at which no chunk starts (a false start, a block of a kind the search does not look for) stops the chain with GAP: one
//      wave decodes from the true position (gi_chunk_kernel in fix-up mode) and the order is taken again;
//   3. WINDOWS: a chunk's last 32 KiB as a FUNCTION of the window before it (a byte, or a reference into that window); gi_window_kernel
//      composes the functions of 32 chain chunks per workgroup (and keeps every chunk's "window before me in terms of the window before
//      my group"), gi_wchain_kernel goes through the groups in stream order -- a sequential depth of 32 + chunks / 32 instead of chunks;
//   4. RESOLVE (gi_resolve_kernel): symbols -> bytes of the text, markers through the chunk's table (in LDS) and its group's window;
//   5. gi_crc_kernel: CRC-32 of every member from pieces of 4 KiB, shifted to the member's end in GF(2) (crc32_combine restated) and
//      XOR-ed together; gi_cuts_kernel / gi_cut_lines_kernel: where records begin, for the caller's batches.
//
// What the search cannot find (stored and fixed blocks, final blocks) is simply decoded by the chunk before.  Data this scheme does not
// suit (hardly anything the search finds; expansion above the pool) makes gn_inflate_step fail with GN_ERANGE: the caller takes its host
// path (pgzip.cpp / zlib) -- the device never guesses, and never returns text it has not decoded bit-exactly.
#include "gn_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

// Phase timers inside the decode kernel (gn_inflate_stats::prof_ms): s_memrealtime is a scalar memory read the wave waits for, four of
// them a batch -- measured cost in scripts/inflate_variants.sh; off unless built with -DGI_PROF=1.
#ifndef GI_PROF
#define GI_PROF 0
#endif
#if GI_PROF
#define GI_NOW() wall_clock64()
#else
#define GI_NOW() 0ull
#endif

namespace
{

constexpr uint32_t GI_WINDOW     = 32768u;
constexpr uint32_t GI_PIECE_LOG2 = 15u; // symbols per output piece (>= the window: a match source is in this piece or the one before)
constexpr uint32_t GI_PIECE      = 1u << GI_PIECE_LOG2;
constexpr uint32_t GI_MAX_PIECES = 64u; // per chunk: 2 Mi symbols
// (zlib's ENOUGH_LENS for a 9-bit root is 852, ENOUGH_DISTS for a 6-bit root 592: gi_build sizes sub-tables as inftrees.c does and refuses
//  over-subscribed and incomplete sets, so no set it accepts needs more.  LDS per wave decides how many chunks a CU decodes at once.)
constexpr uint32_t GI_LIT_ROOT   = 9u, GI_LIT_CAP = 852u;
constexpr uint32_t GI_DIST_ROOT  = 6u, GI_DIST_CAP = 592u;
constexpr uint32_t GI_PRE_ROOT   = 7u;
constexpr uint32_t GI_RING       = 1024u;
constexpr uint32_t GI_MEND       = 32u; // member ends one chunk may hold
constexpr uint64_t GI_NONE       = ~0ull;

enum : uint32_t
{
    GI_F_END      = 1u,  // the stream ended in this chunk (last member's trailer, nothing that is a member behind it)
    GI_F_FAILED   = 2u,  // data error after the chunk's symbols
    GI_F_SHORT    = 4u,  // ran out of the bytes fed so far (not an error while more are to come)
    GI_F_OVERFLOW = 8u,  // pool / pieces-per-chunk / member-end capacity
    GI_F_FRESH    = 16u, // begins at a member header
    GI_F_MEMBERS  = 32u, // members end or begin inside
};

enum : uint32_t
{
    GI_R_RANGE = 0, // every chunk of the range is consumed
    GI_R_GAP,       // no chunk starts at the stream's position
    GI_R_INPUT,     // the next chunk needs bytes that are not fed yet
    GI_R_END,       // end of the gzip stream
    GI_R_DATA,      // damaged data
    GI_R_OVERFLOW,
    GI_R_MEMBER,    // a member's ISIZE is not its length
};

struct GiChunk
{
    uint64_t start_bit, end_bit;
    uint32_t out_len, flags, n_mend, markers;
    uint32_t members_begun, pad;
    uint32_t prof[8]; // 10 ns ticks: [0] screen [1] headers of candidates [2] headers [3] phase A [4] phase B [5] flush [6] whole chunk
    uint32_t mend_sym[GI_MEND], mend_crc[GI_MEND], mend_isize[GI_MEND];
    uint32_t piece[GI_MAX_PIECES];
};

struct GiReal // a chunk the chain took
{
    uint32_t slot, out_len;
    uint64_t text_off;
    uint32_t wbase, pad; // its first item in the resolve pass's work list
};

struct GiState
{
    uint64_t pos_bit;   // the stream is decoded up to here
    uint64_t run_len;   // bytes of the current member so far
    uint64_t text_off;  // bytes of text of this step so far
    uint64_t gap_stop;  // GAP: where the fix-up decode may stop (the next found start behind pos_bit)
    uint32_t cursor;    // next slot of the range to look at
    uint32_t n_real;
    uint32_t n_work;    // (real chunk, piece) items for the resolve kernel
    uint32_t reason;
    uint32_t pad0;
    // what gi_order_kernel found (the fields above it only reads: pos_bit, run_len)
    uint64_t res_pos, res_run_len, res_markers;
    uint32_t res_members, n_mlist; // member ends of the step, in stream order (gi_order_kernel's mlist_*)
    unsigned long long prof[8]; // sums of the chain chunks' GiChunk::prof
};

struct GiLds
{
    uint32_t lit[GI_LIT_CAP];
    uint32_t dst[GI_DIST_CAP];
    union // (the search's queues hold nothing a decode attempt needs, and an attempt that fails starts the search again behind its position)
    {
        uint16_t ring[GI_RING];
        struct
        {
            uint32_t cand_q[256]; // the search: positions (relative to the chunk's nominal start) that passed the first screen (a ring)
            uint32_t cand_q2[128]; // ... and the second
        };
    };
    uint32_t piece[GI_MAX_PIECES];
    union // (code lengths are read while a block's tables are built, the token list while its codes are decoded: never at the same time)
    {
        struct
        {
            uint16_t tok_len[64], tok_val[64]; // the batch's tokens (gi_codes)
        };
        uint8_t lens[320];
    };
};

struct GiParams
{
    const uint32_t* comp;      // compressed file as dwords (padded with zeros)
    uint64_t        avail_bits; // fed so far
    uint64_t        total_bits; // the file
    uint32_t        chunk_bytes;
    uint32_t        j0, n;      // chunks j0 .. j0+n of the file -> slots 0..n
    GiChunk*        chunks;
    uint16_t*       pool;
    uint32_t*       pool_next;
    uint32_t        pool_cap;
    uint32_t*       work_next;  // atomic chunk counter
    // fix-up mode (n == 1): decode from fix_start (no search) into slot fix_slot, stop at the first boundary >= fix_stop
    uint64_t fix_start, fix_stop;
    uint32_t fix_slot;
    uint32_t strict;            // search mode: the first block must be text
};

__device__ const uint16_t kLenBase[32]  = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0, 0 };
__device__ const uint8_t  kLenExtra[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0 };
__device__ const uint16_t kDistBase[32] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0 };
__device__ const uint8_t  kDistExtra[32] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0 };
__device__ const uint8_t  kClOrder[19]  = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

__device__ __forceinline__ uint32_t gi_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t gi_lane() { return threadIdx.x & 63u; }
// Results of functions the compiler may leave as calls come back in vector registers: say that they are wave-uniform, or every
// branch on them counts as divergent and the decode loop's state (bit buffer, positions) ends up in vector registers.
__device__ __forceinline__ bool     gi_ub(bool v) { return gi_rfl(v ? 1u : 0u) != 0u; }
__device__ __forceinline__ uint64_t gi_u64(uint64_t v) { return ((uint64_t)gi_rfl((uint32_t)(v >> 32)) << 32) | gi_rfl((uint32_t)v); }

// ---- input: least-significant-bit-first reader over the dword view of the file; every value is wave-uniform -------------------
struct GiIn
{
    const uint32_t* base;
    uint64_t        n_dw;  // dwords that hold fed bytes (beyond: zeros)
    uint32_t        cur, nxt; // lane l: dword win + l / win + 64 + l
    uint64_t        win;
    uint32_t        idx;
    uint64_t        buf;
    uint32_t        cnt;

    __device__ __forceinline__ uint32_t load(uint64_t i) const { return i < n_dw ? __builtin_nontemporal_load(base + i) : 0u; }
    __device__ __forceinline__ void     seek(uint64_t bit)
    {
        win = bit >> 5;
        idx = 0;
        cur = load(win + gi_lane());
        nxt = load(win + 64 + gi_lane());
        buf = 0;
        cnt = 0;
        refill();
        drop((uint32_t)(bit & 31u));
        refill();
    }
    __device__ __forceinline__ uint64_t pos() const { return ((win + idx) << 5) - cnt; }
    // at least 33 valid bits afterwards
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32u)
        {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)idx);
            buf |= (uint64_t)d << cnt;
            cnt += 32u;
            if (++idx == 64u)
            {
                cur = nxt;
                win += 64;
                idx = 0;
                nxt = load(win + 64 + gi_lane());
            }
        }
    }
    __device__ __forceinline__ void     drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ uint32_t get(uint32_t n) // n <= 16, after a refill
    {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

// ---- output: 16-bit symbols; the newest GI_RING of them in LDS, everything before `flushed` in the chunk's pieces ---------------
struct GiOut
{
    uint32_t  pos, flushed;
    uint32_t  markers;
    bool      ovf;
    uint32_t  tA, tB, tF; // profile: ticks in phase A / B / flushes
    bool      have0; // piece 0 was allocated by an earlier attempt of this chunk (a false start): taken again
    uint16_t* pool;
    uint32_t* pool_next;
    uint32_t  pool_cap;
};

__device__ __forceinline__ uint16_t* gi_sym_addr(uint16_t* pool, const uint32_t* piece, uint32_t p)
{
    return pool + (((uint64_t)piece[p >> GI_PIECE_LOG2] << GI_PIECE_LOG2) | (p & (GI_PIECE - 1u)));
}

__device__ __forceinline__ bool gi_new_piece(GiLds& L, GiOut& o, uint32_t q)
{
    if (q >= GI_MAX_PIECES)
    {
        o.ovf = true;
        return false;
    }
    if (q == 0u && o.have0)
        return true;
    uint32_t id = 0;
    if (gi_lane() == 0)
        id = atomicAdd(o.pool_next, 1u);
    id = gi_rfl(id);
    if (id >= o.pool_cap)
    {
        o.ovf = true;
        return false;
    }
    L.piece[q] = id;
    if (q == 0u)
        o.have0 = true;
    __syncthreads();
    return true;
}

// 256 symbols (whole 128-byte lines) from the ring to the chunk's output
__device__ __forceinline__ void gi_flush256(GiLds& L, GiOut& o)
{
    if ((o.flushed & (GI_PIECE - 1u)) == 0u && !gi_new_piece(L, o, o.flushed >> GI_PIECE_LOG2))
        return;
    uint16_t*   dst = gi_sym_addr(o.pool, L.piece, o.flushed);
    const uint2 v   = *reinterpret_cast<const uint2*>(&L.ring[(o.flushed + 4u * gi_lane()) & (GI_RING - 1u)]);
    *reinterpret_cast<uint2*>(dst + 4u * gi_lane()) = v;
    o.flushed += 256u;
}

__device__ __forceinline__ void gi_flush_rest(GiLds& L, GiOut& o)
{
    while (o.pos - o.flushed >= 256u && !o.ovf)
        gi_flush256(L, o);
    if (o.ovf)
        return;
    const uint32_t rest = o.pos - o.flushed; // < 256: inside one piece
    if (rest)
    {
        if ((o.flushed & (GI_PIECE - 1u)) == 0u && !gi_new_piece(L, o, o.flushed >> GI_PIECE_LOG2))
            return;
        uint16_t* dst = gi_sym_addr(o.pool, L.piece, o.flushed);
        for (uint32_t i = gi_lane(); i < rest; i += 64u)
            dst[i] = L.ring[(o.flushed + i) & (GI_RING - 1u)];
        o.flushed = o.pos;
    }
}

__device__ __forceinline__ void gi_emit(GiLds& L, GiOut& o, uint32_t sym)
{
    L.ring[o.pos & (GI_RING - 1u)] = (uint16_t)sym;
    ++o.pos;
}

// ---- canonical Huffman tables in LDS -------------------------------------------------------------------------------------------
// entry: [31:16] value (literal / base length / base distance / subtable offset)  [9:8] kind (0 literal, 1 base+extra, 2 end of
// block, 3 link to a subtable)  [7:4] extra bits (link: index bits of the subtable)  [3:0] bits to drop (0 = no such code)
template <int KIND> // 0 code-length code, 1 literal/length, 2 distance
__device__ __forceinline__ uint32_t gi_payload(uint32_t s)
{
    if (KIND == 0)
        return s << 16;
    if (KIND == 1)
    {
        if (s < 256u)
            return s << 16;
        if (s == 256u)
            return 0x200u;
        if (s > 285u)
            return ~0u; // (a code that must not occur)
        return ((uint32_t)kLenBase[s - 257u] << 16) | 0x100u | ((uint32_t)kLenExtra[s - 257u] << 4);
    }
    if (s > 29u)
        return ~0u;
    return ((uint32_t)kDistBase[s] << 16) | 0x100u | ((uint32_t)kDistExtra[s] << 4);
}

// lens[0..n) in LDS -> tab.  incomplete: 0 never accepted, 1 a single code of length 1 is, 2 any (the fixed distance code).
// NG = ceil(n / 64).  Returns false for an over-subscribed / incomplete / empty set or a table beyond cap (wave-uniform).
template <int KIND, int NG>
__device__ bool gi_build(uint32_t* tab, uint32_t cap, const uint8_t* lens, uint32_t n, uint32_t root, int incomplete)
{
    const uint32_t lane = gi_lane();
    const uint64_t lt   = (1ull << lane) - 1ull;
    __syncthreads();
    uint32_t l[NG], rank[NG];
    uint32_t cnt = 0; // lane i: codes of length i seen so far
    for (int g = 0; g < NG; ++g)
    {
        const uint32_t s = (uint32_t)g * 64u + lane;
        l[g]             = s < n ? lens[s] : 0u;
        rank[g]          = 0;
        uint64_t rem     = __ballot(l[g] != 0u);
        while (rem)
        {
            const int      src = __builtin_ctzll(rem);
            const uint32_t lv  = (uint32_t)__builtin_amdgcn_readlane((int)l[g], src);
            const uint64_t m   = __ballot(l[g] == lv);
            const uint32_t b   = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)lv);
            if (l[g] == lv)
                rank[g] = b + (uint32_t)__popcll(m & lt);
            if (lane == lv)
                cnt += (uint32_t)__popcll(m);
            rem &= ~m;
        }
    }
    // first code of every length (lane i: of length i), completeness
    int      left = 1;
    uint32_t code = 0, first = 0, used = 0, prev = 0;
    for (uint32_t i = 1; i <= 15u; ++i)
    {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)i);
        code             = (code + prev) << 1;
        if (lane == i)
            first = code;
        prev = c;
        used += c;
        left = (left << 1) - (int)c;
        if (left < 0)
            return false;
    }
    if (used == 0u)
        return false;
    if (left > 0 && !(incomplete == 2 || (incomplete == 1 && used == 1u && (uint32_t)__builtin_amdgcn_readlane((int)cnt, 1) == 1u)))
        return false;
    const uint32_t P = 1u << root;
    for (uint32_t x = lane; x < P; x += 64u)
        tab[x] = 0u;
    __syncthreads();
    uint32_t r[NG];
    for (int g = 0; g < NG; ++g)
    {
        r[g] = 0;
        const uint32_t f = (uint32_t)__shfl((int)first, (int)(l[g] & 15u));
        if (l[g])
        {
            r[g] = __brev(f + rank[g]) >> (32u - l[g]);
            if (l[g] > root)
                atomicMax(&tab[r[g] & (P - 1u)], l[g] - root);
        }
    }
    __syncthreads();
    // subtables: lane owns P/64 consecutive root entries
    const uint32_t E = P >= 64u ? P / 64u : 1u;
    uint32_t       mine = 0;
    if (lane * E < P)
        for (uint32_t x = lane * E; x < lane * E + E; ++x)
            mine += tab[x] ? 1u << tab[x] : 0u;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    const uint32_t total = P + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total > cap)
        return false;
    uint32_t at = P + incl - mine;
    if (lane * E < P)
        for (uint32_t x = lane * E; x < lane * E + E; ++x)
        {
            const uint32_t sb = tab[x];
            if (sb)
            {
                tab[x] = (at << 16) | 0x300u | (sb << 4) | root;
                at += 1u << sb;
            }
        }
    for (uint32_t x = P + lane; x < total; x += 64u)
        tab[x] = 0u;
    __syncthreads();
    for (int g = 0; g < NG; ++g)
    {
        const uint32_t s = (uint32_t)g * 64u + lane;
        if (!l[g])
            continue;
        const uint32_t pay = gi_payload<KIND>(s);
        const bool     ok  = pay != ~0u; // (a code that fills code space but must not be used keeps "no such code" entries)
        if (l[g] <= root)
        {
            const uint32_t e = ok ? (pay | l[g]) : 0u;
            for (uint32_t x = r[g]; x < P; x += 1u << l[g])
                tab[x] = e;
        }
        else
        {
            const uint32_t link = tab[r[g] & (P - 1u)];
            const uint32_t off = link >> 16, sb = (link >> 4) & 15u, hl = l[g] - root;
            const uint32_t e = ok ? (pay | hl) : 0u;
            for (uint32_t x = r[g] >> root; x < (1u << sb); x += 1u << hl)
                tab[off + x] = e;
        }
    }
    __syncthreads();
    return true;
}

// one symbol of the code in `tab`; 0 = no such code.  The caller has refilled (>= 33 bits: root + subtable <= 15).
__device__ __forceinline__ uint32_t gi_decode(const uint32_t* tab, uint32_t root_mask, GiIn& in)
{
    uint32_t e = gi_rfl(tab[(uint32_t)in.buf & root_mask]);
    if ((e & 0x300u) == 0x300u)
    {
        in.drop(e & 15u);
        e = gi_rfl(tab[(e >> 16) + ((uint32_t)in.buf & ((1u << ((e >> 4) & 15u)) - 1u))]);
    }
    in.drop(e & 15u);
    return e;
}

struct GiBlockCtx
{
    bool have_dist;
};

// dynamic block header behind BFINAL/BTYPE: tables into L.lit / L.dst
__device__ bool gi_dynamic_header(GiLds& L, GiIn& in, GiBlockCtx& bc)
{
    const uint32_t lane = gi_lane();
    in.refill();
    const uint32_t hlit = in.get(5) + 257u, hdist = in.get(5) + 1u, hclen = in.get(4) + 4u;
    if (hlit > 286u || hdist > 30u)
        return false;
    __syncthreads();
    if (lane < 19u)
        L.lens[lane] = 0;
    __syncthreads();
    for (uint32_t i = 0; i < hclen; ++i)
    {
        in.refill();
        L.lens[kClOrder[i]] = (uint8_t)in.get(3);
    }
    uint32_t* pre = L.dst; // (the distance table is built after the lengths are read)
    if (!gi_ub(gi_build<0, 1>(pre, 128u, L.lens, 19u, GI_PRE_ROOT, 0)))
        return false;
    const uint32_t tot = hlit + hdist;
    uint32_t       i = 0, prev = 0;
    // (the Kraft sum of the literal/length code, in units of 2^-15, is kept while the lengths are read: a position that is no block
    //  start -- the search tries dozens per chunk -- over-subscribes the code after a few symbols and is dropped there)
    uint32_t kraft = 0;
    __syncthreads();
    while (i < tot)
    {
        in.refill();
        const uint32_t e = gi_rfl(pre[(uint32_t)in.buf & 127u]);
        if ((e & 15u) == 0u)
            return false;
        in.drop(e & 15u);
        const uint32_t s = e >> 16;
        if (s < 16u)
        {
            if (i == hlit)
                kraft = 0;
            L.lens[i++] = (uint8_t)s;
            prev        = s;
            kraft += s ? 32768u >> s : 0u;
            if (kraft > 32768u)
                return false;
            continue;
        }
        uint32_t rep, val;
        if (s == 16u)
        {
            if (i == 0u)
                return false;
            rep = 3u + in.get(2);
            val = prev;
        }
        else if (s == 17u)
        {
            rep = 3u + in.get(3);
            val = 0u;
            prev = 0u;
        }
        else
        {
            rep = 11u + in.get(7);
            val = 0u;
            prev = 0u;
        }
        if (i + rep > tot)
            return false;
        for (uint32_t t = lane; t < rep; t += 64u)
            L.lens[i + t] = (uint8_t)val;
        if (val)
        {
            // (a run may cross from the literal/length lengths into the distance lengths: each side counts its own)
            const uint32_t in_lit = i < hlit ? (i + rep <= hlit ? rep : hlit - i) : 0u;
            if (i < hlit)
            {
                kraft += in_lit * (32768u >> val);
                if (kraft > 32768u)
                    return false;
            }
            if (i + rep > hlit)
            {
                if (i <= hlit)
                    kraft = 0;
                kraft += (rep - in_lit) * (32768u >> val);
                if (kraft > 32768u)
                    return false;
            }
        }
        else if (i <= hlit && i + rep > hlit)
            kraft = 0;
        i += rep;
    }
    __syncthreads();
    if (gi_rfl(L.lens[256]) == 0u)
        return false;
    // the distance lengths move behind the literal/length lengths' 288 slots so that both builds see arrays that begin at 0
    uint8_t dl = 0;
    if (lane < hdist)
        dl = L.lens[hlit + lane];
    __syncthreads();
    if (!gi_ub(gi_build<1, 5>(L.lit, GI_LIT_CAP, L.lens, hlit, GI_LIT_ROOT, 0)))
        return false;
    if (lane < 32u)
        L.lens[lane] = lane < hdist ? dl : 0;
    __syncthreads();
    bc.have_dist = __ballot(dl != 0) != 0ull;
    if (bc.have_dist && !gi_ub(gi_build<2, 1>(L.dst, GI_DIST_CAP, L.lens, hdist, GI_DIST_ROOT, 1)))
        return false;
    return true;
}

__device__ bool gi_fixed_header(GiLds& L, GiBlockCtx& bc)
{
    const uint32_t lane = gi_lane();
    __syncthreads();
    for (uint32_t s = lane; s < 288u; s += 64u)
        L.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8;
    __syncthreads();
    if (!gi_ub(gi_build<1, 5>(L.lit, GI_LIT_CAP, L.lens, 288u, GI_LIT_ROOT, 0)))
        return false;
    if (lane < 32u)
        L.lens[lane] = 5;
    __syncthreads();
    bc.have_dist = true;
    return gi_ub(gi_build<2, 1>(L.dst, GI_DIST_CAP, L.lens, 32u, GI_DIST_ROOT, 2));
}

// One symbol of the chunk's output by position (a marker before the chunk, the ring, or the chunk's flushed pieces)
__device__ __forceinline__ uint32_t gi_fetch(const GiLds& L, const GiOut& o, int32_t sp, int32_t ring_lo)
{
    if (sp < 0)
        return 256u + (uint32_t)((int32_t)GI_WINDOW + sp);
    if (sp >= ring_lo)
        return L.ring[(uint32_t)sp & (GI_RING - 1u)];
    return *gi_sym_addr(o.pool, L.piece, (uint32_t)sp);
}

// The symbols of one Huffman-coded block.  0 = ended with its end-of-block code, 1 = data error, 2 = capacity.
//
// What bounds this kernel is not latency but instruction issue (a SIMD issues one scalar and one vector instruction per four clocks
// whatever the number of waves; a token decoded by a wave-uniform loop costs ~75 of them).  So the 64 lanes do the decoding as well:
//   A. a ROUND looks at the next 64 bit positions: lane l decodes the whole token that would begin at position l (literal/length
//      code, extra bits, distance code, extra bits: two LDS lookups, rarely four) from its own 64 bits of the input window (a
//      register that holds 64 dwords of the file; five of them, taken as scalars, cover a round); then the wave follows the chain
//      from position 0 -- position + bits of the token there is the next token -- with one v_readlane per token, and the lanes on
//      the chain drop their tokens into a list in LDS.  Rounds repeat until the list holds about 64 tokens (or the block ends);
//   B. the lanes place the list's tokens: output offsets by a prefix sum over the lengths; literals are one LDS store; every short
//      match whose source lies before the batch's first symbol is copied by ITS lane from the ring or from the chunk's flushed output
//      (the far loads of up to 64 matches are in flight together); long ones, and matches that read what the batch itself produces
//      (or overlap their own output), follow in token order, each spread over the lanes.  A batch adds at most 512 symbols, so
//      everything it reads near-by is still in the 1024-symbol ring.
#ifndef GI_BATCH_FILL
#define GI_BATCH_FILL 52u // a batch takes rounds until it holds more tokens than this (a round adds seven or so; 64 is the limit)
#endif
#ifndef GI_SHORT_MATCH
#define GI_SHORT_MATCH 12u
#endif
template <bool STRICT>
__device__ int gi_codes(GiLds& L, GiIn& in, GiOut& o, const GiBlockCtx& bc, uint32_t base, bool markers_ok, uint64_t limit_dw)
{
    const uint32_t lane = gi_lane();
    const uint64_t lt   = (1ull << lane) - 1ull;
    // the round's input window: dwords [32 a, 32 a + 64) in wcur, the 64 from 32 (a + 1) on in wnxt
    uint64_t pos = in.pos();
    uint64_t a32 = (pos >> 5) & ~31ull;
    uint32_t wcur = in.load(a32 + lane), wnxt = in.load(a32 + 32u + lane);
    int      end = -1; // 0: end of block seen, 1: error
    while (end < 0)
    {
        // ---- A ---------------------------------------------------------------------------------------------------------------------
        const uint64_t t_a = GI_NOW();
        uint32_t       nt = 0, cum = 0;
        bool           full = false;
        while (end < 0 && !full && nt <= GI_BATCH_FILL)
        {
            uint32_t k0 = (uint32_t)((pos >> 5) - a32);
            if (k0 >= 32u)
            {
                a32 += 32u;
                k0 -= 32u;
                wcur = wnxt;
                wnxt = in.load(a32 + 32u + lane);
            }
            const uint32_t u0 = (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)k0), u1 = (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)k0 + 1),
                           u2 = (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)k0 + 2), u3 = (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)k0 + 3),
                           u4 = (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)k0 + 4);
            const uint32_t sh = (uint32_t)(pos & 31u) + lane, q = sh >> 5, r = sh & 31u;
            const uint32_t d0 = q == 0u ? u0 : q == 1u ? u1 : u2, d1 = q == 0u ? u1 : q == 1u ? u2 : u3, d2 = q == 0u ? u2 : q == 1u ? u3 : u4;
            const uint32_t xl = __funnelshift_r(d0, d1, r), xh = __funnelshift_r(d1, d2, r);
            const uint64_t x  = ((uint64_t)xh << 32) | xl;
            // the token at this lane's position: pk = bits | kind << 6 | length << 8  (kind 0 literal, 1 match, 2 end of block, 3 no such code)
            uint32_t e = L.lit[xl & ((1u << GI_LIT_ROOT) - 1u)], clen = e & 15u;
            if ((e & 0x300u) == 0x300u)
            {
                e    = L.lit[(e >> 16) + ((xl >> GI_LIT_ROOT) & ((1u << ((e >> 4) & 15u)) - 1u))];
                clen = (e & 15u) ? GI_LIT_ROOT + (e & 15u) : 0u;
            }
            uint32_t kind = (e >> 8) & 3u, bits = clen, tlen = 1u, tval = e >> 16;
            if (clen == 0u)
                kind = 3u;
            if (kind == 0u && STRICT && !((tval >= 32u && tval < 127u) || tval == '\n' || tval == '\r' || tval == '\t'))
                kind = 3u;
            if (kind == 1u)
            {
                const uint32_t xb = (e >> 4) & 15u;
                tlen              = tval + ((uint32_t)(x >> clen) & ((1u << xb) - 1u));
                const uint32_t n1 = clen + xb;
                const uint32_t y  = (uint32_t)(x >> n1);
                uint32_t       d = L.dst[y & ((1u << GI_DIST_ROOT) - 1u)], dlen = d & 15u;
                if ((d & 0x300u) == 0x300u)
                {
                    d    = L.dst[(d >> 16) + ((y >> GI_DIST_ROOT) & ((1u << ((d >> 4) & 15u)) - 1u))];
                    dlen = (d & 15u) ? GI_DIST_ROOT + (d & 15u) : 0u;
                }
                const uint32_t db = (d >> 4) & 15u;
                tval              = (d >> 16) + ((y >> dlen) & ((1u << db) - 1u));
                bits              = n1 + dlen + db;
                if (dlen == 0u || !bc.have_dist)
                    kind = 3u;
            }
            const uint32_t pk = bits | (kind << 6) | (tlen << 8);
            // the chain from position 0: one v_readlane per token.  The plain walk takes every token up to an end-of-block code or a
            // position that holds no code; only if the batch cannot take them all (64 tokens, 512 symbols) is it walked again with the caps.
            uint64_t M = 0;
            uint32_t at = 0, cnt = 0, info = 0, cumr = 0;
            do
            {
                info = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)at);
                if (info & 0x80u) // kind 2 or 3
                    break;
                M |= 1ull << at;
                cumr += info >> 8;
                at += info & 63u;
            } while (at < 64u);
            cnt = (uint32_t)__popcll(M);
            if (nt + cnt > 64u || cum + cumr > 512u)
            {
                M    = 0;
                at   = 0;
                cnt  = 0;
                cumr = 0;
                info = 0;
                while (at < 64u)
                {
                    info = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)at);
                    if ((info & 0x80u) || nt + cnt >= 64u || cum + cumr + (info >> 8) > 512u)
                        break;
                    M |= 1ull << at;
                    cumr += info >> 8;
                    ++cnt;
                    at += info & 63u;
                }
                if (!(at < 64u && (info & 0x80u)))
                    full = true;
                if (at >= 64u)
                    info = 0;
            }
            if (at < 64u && (info & 0x80u))
            {
                end = (info & 0x40u) ? 1 : 0; // kind 3: no such code; kind 2: the end-of-block code, whose bits are consumed
                if (end == 0)
                    at += info & 63u;
            }
            cum += cumr;
            pos += at;
            if ((M >> lane) & 1ull)
            {
                const uint32_t slot = nt + (uint32_t)__popcll(M & lt);
                L.tok_len[slot]     = (uint16_t)tlen;
                L.tok_val[slot]     = (uint16_t)(kind == 1u ? tval - 1u : (0x8000u | tval)); // distance - 1 (0..32767), or bit 15 + the literal
            }
            nt += cnt;
        }
        // ---- B ---------------------------------------------------------------------------------------------------------------------
        const uint64_t t_b = GI_NOW();
        o.tA += (uint32_t)(t_b - t_a);
        if (nt)
        {
            __syncthreads();
            const bool mine = lane < nt;
            uint32_t   tl = 0, td = 0, tv = 0;
            if (mine)
            {
                tl                 = L.tok_len[lane];
                const uint32_t raw = L.tok_val[lane];
                if (raw & 0x8000u)
                    tv = raw & 0xFFu;
                else
                    td = raw + 1u;
            }
            __syncthreads();
            uint32_t incl = tl;
            for (int s = 1; s < 64; s <<= 1)
            {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, s);
                if ((int)lane >= s)
                    incl += up;
            }
            const uint32_t my      = o.pos + incl - tl;
            const int32_t  bstart  = (int32_t)o.pos;
            const int32_t  bend    = (int32_t)(o.pos + cum);
            const int32_t  ring_lo = bend - (int32_t)GI_RING;
            const bool     is_m    = mine && td != 0u;
            const int32_t  src     = (int32_t)my - (int32_t)td;
            // a source outside what exists: before the member's first symbol, or further before the chunk than a window
            const bool bad = is_m && src < (int32_t)base && (!markers_ok || base != 0u || src < -(int32_t)GI_WINDOW);
            if (__ballot(bad))
                return 1;
            if (mine && td == 0u)
                L.ring[my & (GI_RING - 1u)] = (uint16_t)tv;
            const bool indep = is_m && td >= tl && src + (int32_t)tl <= bstart;
            // short independent matches, by their lanes: sources all in the ring ...
            const bool in_ring = indep && tl <= GI_SHORT_MATCH && src >= ring_lo && src >= 0;
            // ... or all in the flushed output, inside one piece
            const bool in_out = indep && tl <= GI_SHORT_MATCH && src >= 0 && src + (int32_t)tl <= ring_lo &&
                                ((uint32_t)src >> GI_PIECE_LOG2) == (((uint32_t)src + tl - 1u) >> GI_PIECE_LOG2);
            if (__ballot(is_m && src < ring_lo))
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the lines read below were stored long ago: make it certain)
            if (__ballot(in_out))
            {
                const uint16_t* g = in_out ? gi_sym_addr(o.pool, L.piece, (uint32_t)src) : o.pool;
                uint32_t        v[GI_SHORT_MATCH];
#pragma unroll
                for (uint32_t i = 0; i < GI_SHORT_MATCH; ++i)
                    v[i] = (in_out && i < tl) ? g[i] : 0u;
#pragma unroll
                for (uint32_t i = 0; i < GI_SHORT_MATCH; ++i)
                    if (in_out && i < tl)
                    {
                        L.ring[(my + i) & (GI_RING - 1u)] = (uint16_t)v[i];
                        o.markers += v[i] >= 256u ? 1u : 0u;
                    }
            }
            if (__ballot(in_ring))
            {
#pragma unroll
                for (uint32_t i = 0; i < GI_SHORT_MATCH; ++i)
                    if (in_ring && i < tl)
                    {
                        const uint32_t v = L.ring[((uint32_t)src + i) & (GI_RING - 1u)];
                        L.ring[(my + i) & (GI_RING - 1u)] = (uint16_t)v;
                        o.markers += v >= 256u ? 1u : 0u;
                    }
            }
            // everything else in token order, one match spread over the lanes
            uint64_t rest = __ballot(is_m && !in_ring && !in_out);
            while (rest)
            {
                const int      t    = __builtin_ctzll(rest);
                const uint32_t len  = (uint32_t)__builtin_amdgcn_readlane((int)tl, t);
                const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)td, t);
                const uint32_t at   = (uint32_t)__builtin_amdgcn_readlane((int)my, t);
                const int32_t  s0   = (int32_t)at - (int32_t)dist;
                for (uint32_t u = 0; u < len; u += 64u)
                {
                    const uint32_t i = u + lane;
                    if (i < len)
                    {
                        const uint32_t off = dist >= len ? i : (dist == 1u ? 0u : i % dist);
                        const uint32_t v   = gi_fetch(L, o, s0 + (int32_t)off, ring_lo);
                        o.markers += v >= 256u ? 1u : 0u;
                        L.ring[(at + i) & (GI_RING - 1u)] = (uint16_t)v;
                    }
                }
                rest &= rest - 1ull;
            }
            o.pos += cum;
            const uint64_t t_f = GI_NOW();
            o.tB += (uint32_t)(t_f - t_b);
            while (o.pos - o.flushed >= 512u)
            {
                gi_flush256(L, o);
                if (o.ovf)
                    return 2;
            }
            o.tF += (uint32_t)(GI_NOW() - t_f);
            if ((pos >> 5) > limit_dw) // (reading zeros beyond the fed bytes: whatever this is, it ends here)
                return 1;
        }
    }
    in.seek(pos);
    return end;
}

// (every caller's p is wave-uniform; the readfirstlane says so to the compiler: a branch on a loaded value would otherwise count as
//  divergent, and with it every value the decode loop carries -- bit buffer, positions -- would live in vector registers)
__device__ __forceinline__ uint32_t gi_byte(const uint32_t* comp, uint64_t p) { return gi_rfl((comp[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu); }

// gzip member header at byte p; offset of the deflate data behind it, 0 = no (complete) member header there
__device__ uint64_t gi_gzip_header(const uint32_t* comp, uint64_t size, uint64_t p)
{
    if (p + 18u > size || gi_byte(comp, p) != 0x1Fu || gi_byte(comp, p + 1) != 0x8Bu || gi_byte(comp, p + 2) != 8u || (gi_byte(comp, p + 3) & 0xE0u))
        return 0;
    const uint32_t flg = gi_byte(comp, p + 3);
    uint64_t       q   = p + 10u;
    if (flg & 4u)
    {
        if (q + 2u > size)
            return 0;
        q += 2u + (gi_byte(comp, q) | (gi_byte(comp, q + 1) << 8));
    }
    for (uint32_t bit = 8u; bit <= 16u; bit <<= 1)
        if (flg & bit)
        {
            while (q < size && gi_byte(comp, q))
                ++q;
            ++q;
        }
    if (flg & 2u)
        q += 2u;
    return q < size ? q : 0;
}

// The search's screens (per lane).  First: BFINAL = 0, BTYPE = 2, HLIT / HDIST in range -- 17 bits, passed by one position in nine.
__device__ __forceinline__ bool gi_screen_bits(uint32_t x0)
{
    return (x0 & 7u) == 4u && ((x0 >> 3) & 31u) <= 29u && ((x0 >> 8) & 31u) <= 29u;
}
// Second, for the positions that passed, 64 of them at a time: is the code-length code complete?  (96 bits d0..d2 from bit sh < 32 on)
__device__ __forceinline__ bool gi_screen_kraft(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t sh)
{
    const uint32_t x0 = __funnelshift_r(d0, d1, sh), x1 = __funnelshift_r(d1, d2, sh), x2 = __funnelshift_r(d2, d3, sh);
    const uint32_t hclen = ((x0 >> 13) & 15u) + 4u;
    uint64_t       p     = (uint64_t)__funnelshift_r(x0, x1, 17) | ((uint64_t)__funnelshift_r(x1, x2, 17) << 32);
    p &= (1ull << (3u * hclen)) - 1ull;
    // Kraft sum in units of 2^-7: a length l > 0 weighs 128 >> l
    const uint64_t weigh = 0x0102040810204000ull; // byte l = 128 >> l, byte 0 = 0
    uint32_t       sum   = 0;
#pragma unroll
    for (int i = 0; i < 19; ++i)
        sum += (uint32_t)(weigh >> (8u * ((uint32_t)(p >> (3 * i)) & 7u))) & 0xFFu;
    return sum == 128u;
}


// Third screen, per lane as well: the whole dynamic block header at `bit` (BFINAL/BTYPE included), without tables -- the code-length code is
// decoded canonically from counts held in registers.  true = the code lengths decode, the literal/length code is complete and has an
// end-of-block code, the distance code is complete, a single code, or absent.  (Stricter than the decoder where zlib is lenient; what the
// search misses the chunk before simply decodes.)  A chunk's search tries some sixty positions that pass the first two screens; this
// drops all but the real block start for the price of one header, 64 positions at a time.
__device__ __noinline__ bool gi_screen_header(const uint32_t* __restrict__ comp, uint64_t n_dw, uint64_t bit)
{
    uint64_t di = bit >> 5;
    auto     ld = [&](uint64_t i) -> uint64_t { return i < n_dw ? comp[i] : 0u; };
    uint64_t buf = (ld(di) | (ld(di + 1) << 32)) >> (bit & 31u);
    uint32_t cnt = 64u - (uint32_t)(bit & 31u);
    di += 2;
    auto need = [&](uint32_t n) { // n <= 32
        if (cnt < n)
        {
            buf |= ld(di) << cnt;
            cnt += 32u;
            ++di;
        }
    };
    auto get = [&](uint32_t n) -> uint32_t {
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    };
    need(17);
    get(3);
    const uint32_t hlit = get(5) + 257u, hdist = get(5) + 1u, hclen = get(4) + 4u;
    // the code-length code's lengths, 3 bits per symbol, and how many symbols have each length (a byte per length)
    uint64_t clp = 0, cpk = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19u; ++i)
        if (i < hclen)
        {
            need(3);
            const uint64_t l = get(3);
            clp |= l << (3u * kClOrder[i]);
            cpk += 1ull << (8u * l);
        }
    // canonical code: first code and first position in the sorted symbol list, per length
    uint32_t first[8], offs[8], num[8];
    {
        uint32_t code = 0, off = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7u; ++l)
        {
            const uint32_t c = (uint32_t)(cpk >> (8u * l)) & 0xFFu;
            first[l] = code;
            offs[l]  = off;
            num[l]   = c;
            code     = (code + c) << 1;
            off += c;
        }
    }
    // symbols sorted by (length, symbol), 5 bits each
    uint64_t s_lo = 0, s_hi = 0; // entries 0..11, 12..18
    {
        uint32_t at = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7u; ++l)
#pragma unroll
            for (uint32_t sy = 0; sy < 19u; ++sy)
                if (((uint32_t)(clp >> (3u * sy)) & 7u) == l)
                {
                    if (at < 12u)
                        s_lo |= (uint64_t)sy << (5u * at);
                    else
                        s_hi |= (uint64_t)sy << (5u * (at - 12u));
                    ++at;
                }
    }
    const uint32_t tot = hlit + hdist;
    uint32_t       i = 0, prev = 0, kraft = 0, used = 0;
    bool           eob = false;
    while (i < tot)
    {
        need(14); // a code of up to 7 bits and up to 7 extra bits
        uint32_t code = 0, sym = 32u, len = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7u; ++l)
        {
            code = (code << 1) | ((uint32_t)(buf >> (l - 1u)) & 1u);
            const uint32_t idx = code - first[l];
            if (sym == 32u && idx < num[l])
            {
                const uint32_t at = offs[l] + idx;
                sym = at < 12u ? (uint32_t)(s_lo >> (5u * at)) & 31u : (uint32_t)(s_hi >> (5u * (at - 12u))) & 31u;
                len = l;
            }
        }
        if (sym == 32u)
            return false;
        get(len);
        uint32_t rep = 1, val = sym;
        if (sym == 16u)
        {
            if (i == 0u)
                return false;
            rep = 3u + get(2);
            val = prev;
        }
        else if (sym == 17u)
        {
            rep = 3u + get(3);
            val = 0;
        }
        else if (sym == 18u)
        {
            rep = 11u + get(7);
            val = 0;
        }
        if (i + rep > tot)
            return false;
        prev = val;
        if (i < hlit && i + rep > hlit) // a run across the two codes' lengths: the literal/length side first
        {
            const uint32_t a = hlit - i;
            if (val)
            {
                kraft += a * (32768u >> val);
                eob = eob || (i <= 256u);
            }
            if (kraft != 32768u || !eob)
                return false;
            kraft = 0;
            used  = 0;
            i += a;
            rep -= a;
        }
        else if (i == hlit)
        {
            if (kraft != 32768u || !eob)
                return false;
            kraft = 0;
            used  = 0;
        }
        if (val)
        {
            kraft += rep * (32768u >> val);
            used += rep;
            if (i < hlit && i <= 256u && i + rep > 256u)
                eob = true;
            if (kraft > 32768u)
                return false;
        }
        i += rep;
    }
    // (hdist >= 1: the loop passed i == hlit or crossed it, so the literal/length code has been judged; here: the distance code)
    return kraft == 32768u || used <= 1u;
}

} // namespace

// One wave per chunk, one chunk per workgroup (the slot comes from an atomic counter: chunks start in file order whatever the dispatcher does).
__global__ __launch_bounds__(64) void gi_chunk_kernel(GiParams p)
{
    __shared__ GiLds L;
    const uint32_t   lane = gi_lane();
    const uint64_t   n_dw = (p.avail_bits + 31u) >> 5;
    for (;;)
    {
        uint32_t slot = 0;
        if (p.fix_start == GI_NONE)
        {
            if (lane == 0)
                slot = atomicAdd(p.work_next, 1u);
            slot = gi_rfl(slot);
            if (slot >= p.n)
                return;
        }
        const uint32_t j       = p.j0 + slot;
        const bool     fix     = p.fix_start != GI_NONE;
        GiChunk&       C       = p.chunks[fix ? p.fix_slot : slot];
        const uint64_t nominal = (uint64_t)j * p.chunk_bytes * 8u;
        uint64_t       stop    = fix ? p.fix_stop : nominal + (uint64_t)p.chunk_bytes * 8u;
        if (stop > p.total_bits)
            stop = p.total_bits;
        const bool all_fed = p.avail_bits >= p.total_bits;

        GiIn in;
        in.base = p.comp;
        in.n_dw = n_dw;
        GiOut o;
        o.pool      = p.pool;
        o.pool_next = p.pool_next;
        o.pool_cap  = p.pool_cap;
        o.have0     = false;
        o.tA = o.tB = o.tF = 0;
        uint32_t       t_screen = 0, t_cand = 0, t_hdr = 0, n_cand = 0;
        const uint64_t t_chunk  = GI_NOW();
        GiBlockCtx bc;
        bc.have_dist = false;

        uint64_t start      = GI_NONE;
        uint64_t search_at  = fix ? GI_NONE : nominal; // next position the search looks at (GI_NONE: no search)
        uint64_t sw_base = ~0ull, sq_m = 0; // search: base dword of the register window; candidates of the running batch that passed both screens
        uint32_t sq_h = 0, sq2_h = 0; // (heads of the two rings)
        uint32_t sw = 0, sq_n = 0, sq2_n = 0, sq_pos = 0; // ... the window; positions queued behind the first / second screen; the running batch's (per lane)
        bool     at_header  = false;
        uint32_t flags      = 0;
        if (!fix && j == 0)
        {
            start     = 0;
            at_header = true;
            search_at = GI_NONE;
        }
        else if (fix)
        {
            start = p.fix_start;
            // (the stream may stand in front of a member header -- a chunk ends there when the header lies behind the next chunk's nominal
            //  start; no block begins with 1f on a byte boundary: final + reserved type)
            const uint64_t sb = start >> 3;
            at_header = (start & 7u) == 0u && (sb + 18u) * 8u <= p.avail_bits && gi_byte(p.comp, sb) == 0x1Fu && gi_byte(p.comp, sb + 1) == 0x8Bu
                        && gi_byte(p.comp, sb + 2) == 8u;
        }

        uint32_t n_mend = 0, members = 0;
        uint64_t end_bit = 0;
        bool     done    = false;
        while (!done)
        {
            // ---- the chunk's start -----------------------------------------------------------------------------------------------
            bool probation = false; // the first block runs in strict mode and a failure resumes the search
            if (search_at != GI_NONE)
            {
                // Two screens: 64 positions a round take the cheap one, the survivors queue up in LDS; when 64 are queued (or the range
                // is through) they take the second screen together, and what is left is tried in position order.
                start = GI_NONE;
                const uint64_t t_s = GI_NOW();
                for (;;)
                {
                    if (sq_m)
                    {
                        const int t = __builtin_ctzll(sq_m);
                        sq_m &= sq_m - 1ull;
                        const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)sq_pos, t);
                        start     = nominal + (cand & 0x7FFFFFFFu);
                        at_header = (cand >> 31) != 0u; // a gzip member header, not a block header
                        break;
                    }
                    if (sq2_n >= 16u || (search_at >= stop && sq_n == 0u && sq2_n))
                    {
                        // third screen: the whole header, for up to 64 positions that passed the second (sixteen wait for one another: a
                        // wave pays for the longest header among its lanes whatever their number)
                        const uint32_t take = sq2_n < 64u ? sq2_n : 64u;
                        __syncthreads();
                        sq_pos = lane < take ? L.cand_q2[(sq2_h + lane) & 127u] : 0u;
                        sq2_h += take;
                        sq2_n -= take;
                        const bool ok = lane < take && ((sq_pos >> 31) || gi_screen_header(p.comp, n_dw, nominal + sq_pos));
                        sq_m          = __ballot(ok);
                        continue;
                    }
                    if (sq_n >= 64u || (search_at >= stop && sq_n))
                    {
                        const uint32_t take = sq_n < 64u ? sq_n : 64u;
                        __syncthreads();
                        const uint32_t mypos = lane < take ? L.cand_q[(sq_h + lane) & 255u] : 0u;
                        sq_h += take;
                        sq_n -= take;
                        const uint64_t bit = nominal + (mypos & 0x7FFFFFFFu);
                        const uint64_t di  = bit >> 5;
                        bool           ok  = lane < take && (mypos >> 31);
                        if (lane < take && !(mypos >> 31))
                        {
                            const uint32_t d0 = di < n_dw ? p.comp[di] : 0u, d1 = di + 1 < n_dw ? p.comp[di + 1] : 0u, d2 = di + 2 < n_dw ? p.comp[di + 2] : 0u,
                                           d3 = di + 3 < n_dw ? p.comp[di + 3] : 0u;
                            ok = gi_screen_kraft(d0, d1, d2, d3, (uint32_t)(bit & 31u));
                        }
                        const uint64_t m2 = __ballot(ok);
                        if (ok)
                            L.cand_q2[(sq2_h + sq2_n + (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull))) & 127u] = mypos;
                        sq2_n += (uint32_t)__popcll(m2);
                        continue;
                    }
                    if (search_at >= stop)
                        break;
                    // first screen: 256 positions a round, four to a lane, out of the register that holds 64 dwords of the file
                    const uint64_t d = search_at >> 5;
                    if (sw_base == ~0ull || d < sw_base || d + 10u > sw_base + 64u)
                    {
                        sw_base = d;
                        sw      = sw_base + lane < n_dw ? p.comp[sw_base + lane] : 0u;
                    }
                    const uint32_t off = (uint32_t)(search_at - (sw_base << 5)) + 4u * lane;
                    const uint32_t d0 = (uint32_t)__shfl((int)sw, (int)(off >> 5)), d1 = (uint32_t)__shfl((int)sw, (int)(off >> 5) + 1);
                    const uint32_t x  = __funnelshift_r(d0, d1, off & 31u);
                    const uint64_t bit = search_at + 4u * lane;
                    const uint64_t lim = stop < p.avail_bits - 96u ? stop : p.avail_bits - 96u; // (a position is looked at with 96 bits behind it)
                    bool     okj[4], isj[4];
                    uint64_t mj[4];
                    uint32_t before = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j)
                    {
                        // a block header -- or, on a byte boundary, a gzip member header (1f 8b 08: what follows is checked when it is tried)
                        const bool member = ((bit + j) & 7u) == 0u && ((x >> j) & 0xFFFFFFu) == 0x088B1Fu;
                        okj[j] = bit + j < lim && (gi_screen_bits(x >> j) || member);
                        isj[j] = member;
                        mj[j]  = __ballot(okj[j]);
                        before += (uint32_t)__popcll(mj[j] & ((1ull << lane) - 1ull));
                    }
                    uint32_t total = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j)
                    {
                        const uint32_t slot = sq_n + before;
                        if (okj[j] && slot < 256u) // (a ring of 256: what does not fit is dropped -- the chunk before decodes what the search misses)
                            L.cand_q[(sq_h + slot) & 255u] = (uint32_t)(bit + j - nominal) | (isj[j] ? 0x80000000u : 0u);
                        before += okj[j] ? 1u : 0u;
                        total += (uint32_t)__popcll(mj[j]);
                    }
                    sq_n = sq_n + total < 256u ? sq_n + total : 256u;
                    search_at += 256u;
                }
                t_screen += (uint32_t)(GI_NOW() - t_s);
                if (start == GI_NONE)
                {
                    flags = 0;
                    break; // nothing in this chunk's range
                }
                probation = true;
            }
            // ---- decode ----------------------------------------------------------------------------------------------------------
            o.pos = o.flushed = 0;
            o.markers = 0;
            o.ovf     = false;
            n_mend    = 0;
            members   = 0;
            flags     = 0;
            uint32_t base         = 0;     // first symbol of the current member
            bool     fresh_member = false; // a member began inside this chunk: no markers
            uint64_t pos          = start;
            bool     failed       = false;
            if (at_header)
            {
                const uint64_t q = gi_u64(gi_gzip_header(p.comp, p.total_bits >> 3, start >> 3));
                if (!q && probation) // (three bytes that look like a member's first: the search goes on behind them)
                {
                    search_at = start + 1u;
                    sq_n = sq2_n = sq_h = sq2_h = 0;
                    sq_m      = 0;
                    continue;
                }
                if (!q)
                {
                    flags   = GI_F_FAILED;
                    end_bit = start;
                    done    = true;
                    break;
                }
                pos          = q * 8u;
                fresh_member = true;
                members      = 1;
                flags |= GI_F_FRESH;
            }
            in.seek(pos);
            bool first_block = true;
            for (;;)
            {
                const uint64_t here = in.pos();
                if (here + 3u > p.avail_bits)
                {
                    if (!all_fed)
                        flags |= GI_F_SHORT;
                    else
                        failed = true;
                    end_bit = here;
                    break;
                }
                in.refill();
                // A chunk ends at the first block boundary at or behind the next chunk's nominal start at which a block begins that
                // the next chunk's search can find (not final, dynamic): stored, fixed and final blocks there are this chunk's.
                if (!first_block && here >= stop && in.peek(3) == 4u)
                {
                    end_bit = here;
                    break;
                }
                const uint32_t bfinal = in.get(1), btype = in.get(2);
                const bool     mk     = !fresh_member;
                int            rc     = 0;
                if (btype == 0u)
                {
                    in.drop(in.cnt & 7u);
                    in.refill();
                    const uint32_t len = in.get(16);
                    in.refill();
                    const uint32_t nlen = in.get(16);
                    if ((len ^ 0xFFFFu) != nlen || in.pos() + (uint64_t)len * 8u > p.avail_bits)
                        rc = (in.pos() + (uint64_t)len * 8u > p.avail_bits && !all_fed && (len ^ 0xFFFFu) == nlen) ? 3 : 1;
                    else
                        for (uint32_t i = 0; i < len && !o.ovf; ++i)
                        {
                            in.refill();
                            gi_emit(L, o, in.get(8));
                            if (o.pos - o.flushed >= 512u)
                                gi_flush256(L, o);
                        }
                    if (o.ovf)
                        rc = 2;
                }
                else if (btype == 3u)
                    rc = 1;
                else
                {
                    const uint64_t t_h = GI_NOW();
                    const bool hdr = gi_ub(btype == 2u ? gi_dynamic_header(L, in, bc) : gi_fixed_header(L, bc));
                    if (probation && first_block)
                    {
                        t_cand += (uint32_t)(GI_NOW() - t_h);
                        ++n_cand;
                    }
                    else
                        t_hdr += (uint32_t)(GI_NOW() - t_h);
                    if (!hdr)
                        rc = 1;
                    else if (probation && first_block && p.strict)
                        rc = (int)gi_rfl((uint32_t)gi_codes<true>(L, in, o, bc, base, mk, n_dw + 2u));
                    else
                        rc = (int)gi_rfl((uint32_t)gi_codes<false>(L, in, o, bc, base, mk, n_dw + 2u));
                }
                if (rc == 0 && in.pos() > p.avail_bits)
                    rc = all_fed ? 1 : 3;
                if (rc == 3)
                {
                    flags |= GI_F_SHORT;
                    end_bit = here;
                    break;
                }
                if (rc == 2)
                {
                    flags |= GI_F_OVERFLOW;
                    end_bit = here;
                    break;
                }
                if (rc == 1)
                {
                    // (beyond the fed bytes a decode sees zeros: that is "short", not "damaged", while more is to come)
                    if (!all_fed && in.pos() + 64u > p.avail_bits)
                        flags |= GI_F_SHORT;
                    else
                        failed = true;
                    end_bit = in.pos();
                    break;
                }
                first_block = false;
                if (bfinal)
                {
                    in.drop(in.cnt & 7u);
                    const uint64_t at = in.pos() >> 3;
                    if ((at + 8u) * 8u > p.avail_bits)
                    {
                        if (!all_fed)
                            flags |= GI_F_SHORT;
                        else
                            failed = true;
                        end_bit = in.pos();
                        break;
                    }
                    if (n_mend >= GI_MEND)
                    {
                        flags |= GI_F_OVERFLOW;
                        end_bit = in.pos();
                        break;
                    }
                    if (lane == 0)
                    {
                        C.mend_sym[n_mend]   = o.pos;
                        C.mend_crc[n_mend]   = gi_byte(p.comp, at) | (gi_byte(p.comp, at + 1) << 8) | (gi_byte(p.comp, at + 2) << 16) | (gi_byte(p.comp, at + 3) << 24);
                        C.mend_isize[n_mend] = gi_byte(p.comp, at + 4) | (gi_byte(p.comp, at + 5) << 8) | (gi_byte(p.comp, at + 6) << 16) | (gi_byte(p.comp, at + 7) << 24);
                    }
                    ++n_mend;
                    const uint64_t next = at + 8u;
                    // (the next member's header must lie inside what is fed, or the file must be at its end)
                    if (!all_fed && (next + 4096u) * 8u > p.avail_bits)
                    {
                        flags |= GI_F_SHORT;
                        end_bit = here;
                        // the member end just recorded belongs to a decode that will be repeated
                        break;
                    }
                    const uint64_t q = next < (p.total_bits >> 3) ? gi_u64(gi_gzip_header(p.comp, p.total_bits >> 3, next)) : 0;
                    if (!q)
                    {
                        flags |= GI_F_END;
                        end_bit = next * 8u;
                        break;
                    }
                    if (next * 8u >= stop) // the next chunk's search finds this header: the chunk ends in front of it
                    {
                        end_bit = next * 8u;
                        break;
                    }
                    in.seek(q * 8u);
                    base         = o.pos;
                    fresh_member = true;
                    ++members;
                }
            }
            if (failed && probation && first_block) // a false start: the search goes on behind it (its queues shared the ring's memory: from scratch)
            {
                search_at = start + 1u;
                sq_n = sq2_n = sq_h = sq2_h = 0;
                sq_m      = 0;
                continue;
            }
            if (failed)
                flags |= GI_F_FAILED;
            done = true;
        }
        if (start != GI_NONE && !(flags & GI_F_OVERFLOW))
        {
            gi_flush_rest(L, o);
            if (o.ovf)
                flags |= GI_F_OVERFLOW;
        }
        uint32_t mk = start != GI_NONE ? o.markers : 0u;
        for (int s = 32; s > 0; s >>= 1)
            mk += (uint32_t)__shfl_xor((int)mk, s);
        __syncthreads();
        if (start != GI_NONE)
            for (uint32_t q = lane; q < GI_MAX_PIECES; q += 64u)
                C.piece[q] = L.piece[q];
        if (lane == 0)
        {
            C.start_bit     = start;
            C.end_bit       = end_bit;
            C.out_len       = start != GI_NONE ? o.pos : 0u;
            C.flags         = flags | ((n_mend || members) ? GI_F_MEMBERS : 0u);
            C.n_mend        = n_mend;
            C.markers       = mk;
            C.members_begun = members;
            C.prof[0] = t_screen;
            C.prof[1] = t_cand;
            C.prof[2] = t_hdr;
            C.prof[3] = o.tA;
            C.prof[4] = o.tB;
            C.prof[5] = o.tF;
            C.prof[6] = (uint32_t)(GI_NOW() - t_chunk);
            C.prof[7] = n_cand * 100000u; // (shown as a count by the /1e5 of the millisecond conversion)
        }
        __syncthreads();
        // One chunk a workgroup: waves that ran on for a whole launch kept every other kernel -- the filter loader's copies, the
        // classification's -- waiting for a launch's length; the hardware's dispatcher hands out the slots as well as the counter did.
        return;
    }
}

// ---- the stream's order -------------------------------------------------------------------------------------------------------------
// One workgroup, the slots' key fields in LDS.  A chunk that starts at a position lies in the slot of that position's nominal range (or
// in a fix-up slot), so "which chunk continues this one" is a lookup, done for every slot at once; thread 0 then only follows the
// successor links from the stream's position (two LDS reads a chunk), and everything else -- text offsets and work-list offsets by
// prefix sums over the chain, the chain records, the work list -- is done by all threads again.  The kernel only READS the step's
// initial state (pos_bit, run_len) and writes its result fields, so the host can run it again after a fix-up decode.
#define GI_ORDER_MAX 8256u // regular slots of a step (<= 8192) + fix-up slots (<= 64); 16 bytes of LDS each
#define GI_NOIDX 0xFFFFu

__global__ __launch_bounds__(1024) void gi_order_kernel(GiState* st, const GiChunk* chunks, uint32_t n_slots, uint32_t slots_cap, uint32_t n_fix, uint32_t j0,
                                                        uint32_t chunk_bytes, uint64_t range_end_bits, uint64_t total_bits, GiReal* real, uint2* work,
                                                        uint32_t real_cap, uint32_t work_cap, uint64_t text_cap, unsigned long long* mlist_pos,
                                                        uint32_t* mlist_crc, uint32_t mlist_cap)
{
    extern __shared__ uint64_t gi_order_lds[];
    const uint32_t             n_all   = n_slots + n_fix;
    uint64_t*                  s_start = gi_order_lds;                                  // (a prefix-sum buffer once the links exist)
    uint32_t*                  s_lf    = reinterpret_cast<uint32_t*>(s_start + n_all); // out_len | flags << 24
    uint16_t*                  s_succ  = reinterpret_cast<uint16_t*>(s_lf + n_all);    // the chunk that starts where this one ends
    uint16_t*                  s_order = s_succ + n_all;                                // chain order
    __shared__ uint32_t        s_R, s_head;
    __shared__ uint64_t        s_part[1024];
    const uint32_t             tid   = threadIdx.x;
    const uint64_t             cbits = (uint64_t)chunk_bytes * 8u;
    auto slot_of = [&](uint32_t idx) { return idx < n_slots ? idx : slots_cap + (idx - n_slots); };
    for (uint32_t i = tid; i < n_all; i += 1024u)
    {
        const GiChunk& c = chunks[slot_of(i)];
        s_start[i]       = c.start_bit;
        s_lf[i]          = c.out_len | (c.flags << 24);
    }
    __syncthreads();
    // the chunk (index) that starts at pos; GI_NOIDX: none
    auto lookup = [&](uint64_t pos) -> uint32_t {
        for (uint32_t f = 0; f < n_fix; ++f)
            if (s_start[n_slots + f] == pos)
                return n_slots + f;
        const uint64_t sj = pos / cbits;
        if (sj >= j0 && sj - j0 < n_slots && s_start[sj - j0] == pos)
            return (uint32_t)(sj - j0);
        return GI_NOIDX;
    };
    for (uint32_t i = tid; i < n_all; i += 1024u)
        s_succ[i] = s_start[i] == GI_NONE ? (uint16_t)GI_NOIDX : (uint16_t)lookup(chunks[slot_of(i)].end_bit);
    if (tid == 0)
        s_head = lookup(st->pos_bit);
    __syncthreads();
    if (tid == 0)
    {
        uint32_t r = 0, reason = ~0u, cursor = n_slots, idx = s_head;
        bool     is_short = false;
        while (idx != GI_NOIDX)
        {
            const uint32_t fl = s_lf[idx] >> 24;
            if (fl & (GI_F_SHORT | GI_F_OVERFLOW))
            {
                reason   = (fl & GI_F_SHORT) ? GI_R_INPUT : GI_R_OVERFLOW;
                is_short = (fl & GI_F_SHORT) != 0u;
                break;
            }
            if (r >= real_cap)
            {
                reason = GI_R_OVERFLOW;
                break;
            }
            s_order[r++] = (uint16_t)idx;
            if (fl & (GI_F_FAILED | GI_F_END))
            {
                reason = (fl & GI_F_FAILED) ? GI_R_DATA : GI_R_END;
                break;
            }
            idx = s_succ[idx];
        }
        // where the stream stands: behind the last chunk taken
        const uint64_t pos = r ? chunks[slot_of(s_order[r - 1u])].end_bit : st->pos_bit;
        if (is_short) // the next step begins with this chunk (a fix-up decode that ran short: with the chunk in whose range the stream stands)
        {
            const uint64_t sj = pos / cbits;
            cursor = idx < n_slots ? idx : (sj >= j0 ? (uint32_t)(sj - j0 < n_slots ? sj - j0 : n_slots) : 0u);
        }
        uint64_t gap_stop = 0;
        if (reason == ~0u)
        {
            // nothing starts at pos: a gap up to the next start behind pos, or the end of the range's chunks
            const uint64_t sj = pos / cbits;
            uint32_t       s  = sj >= j0 ? (uint32_t)(sj - j0 < n_slots ? sj - j0 : n_slots) : 0u;
            while (s < n_slots && (s_start[s] == GI_NONE || s_start[s] <= pos))
                ++s;
            if (s < n_slots)
            {
                reason   = GI_R_GAP;
                gap_stop = s_start[s];
            }
            else if (pos < range_end_bits && pos < total_bits)
            {
                reason   = GI_R_GAP;
                gap_stop = range_end_bits < total_bits ? range_end_bits : total_bits;
            }
            else
                reason = GI_R_RANGE;
            cursor = s;
        }
        st->reason   = reason;
        st->cursor   = cursor;
        st->gap_stop = gap_stop;
        st->res_pos  = pos;
        s_R          = r;
    }
    __syncthreads();
    const uint32_t R = s_R;
    // text offsets and work-list offsets: exclusive prefix sums over the chain (both in one 64-bit sum: pieces << 40 | symbols)
    const uint32_t per = (R + 1023u) / 1024u;
    uint64_t       mine = 0;
    for (uint32_t k = 0; k < per; ++k)
    {
        const uint32_t r = tid * per + k;
        if (r < R)
        {
            const uint64_t n = s_lf[s_order[r]] & 0xFFFFFFu;
            mine += n | (((n + GI_PIECE - 1u) >> GI_PIECE_LOG2) << 40);
        }
    }
    s_part[tid] = mine;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1)
    {
        const uint64_t v = tid >= o ? s_part[tid - o] : 0ull;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const uint64_t all = s_part[1023];
    uint64_t       run = s_part[tid] - mine;
    unsigned long long mk = 0;
    for (uint32_t k = 0; k < per; ++k)
    {
        const uint32_t r = tid * per + k;
        if (r < R)
        {
            const uint32_t idx = s_order[r], n = s_lf[idx] & 0xFFFFFFu;
            const uint64_t text = run & ((1ull << 40) - 1ull);
            const uint32_t w    = (uint32_t)(run >> 40);
            real[r].slot     = slot_of(idx);
            real[r].out_len  = n;
            real[r].text_off = text;
            real[r].wbase    = w;
            if ((all >> 40) <= work_cap)
                for (uint32_t q = 0; (q << GI_PIECE_LOG2) < n; ++q)
                    work[w + q] = make_uint2(r, q);
            const GiChunk& c = chunks[slot_of(idx)];
            mk += c.markers;
            for (int kk = 0; kk < 8; ++kk)
                atomicAdd(&st->prof[kk], (unsigned long long)c.prof[kk]);
            run += n | (((uint64_t)(n + GI_PIECE - 1u) >> GI_PIECE_LOG2) << 40);
        }
    }
    if (mk)
        atomicAdd(reinterpret_cast<unsigned long long*>(&st->res_markers), mk);
    // members that end in the step: lengths against ISIZE, the list for the CRC pass.  A blocked-gzip file ends a member every 64 KiB of
    // text, so this is done by all threads as well: per chain chunk the number of member ends and the position of its last one, summed /
    // max-ed over the chain (an exclusive scan of both), then every thread judges the member ends of its own chunks.
    __shared__ uint64_t s_last[1024];
    __shared__ uint32_t s_bad, s_members;
    if (tid == 0)
    {
        s_bad     = 0;
        s_members = 0;
    }
    __syncthreads();
    uint64_t cnt_mine = 0, last_mine = 0; // member ends in this thread's chunks; position + 1 of the last of them (0: none)
    for (uint32_t k = 0; k < per; ++k)
    {
        const uint32_t r = tid * per + k;
        if (r < R && ((s_lf[s_order[r]] >> 24) & GI_F_MEMBERS))
        {
            const GiChunk& c = chunks[slot_of(s_order[r])];
            if (c.n_mend)
            {
                cnt_mine += c.n_mend;
                last_mine = real[r].text_off + c.mend_sym[c.n_mend - 1u] + 1u;
            }
            atomicAdd(&s_members, c.members_begun);
        }
    }
    __syncthreads(); // (s_part is read above)
    s_part[tid] = cnt_mine;
    s_last[tid] = last_mine;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1)
    {
        const uint64_t v = tid >= o ? s_part[tid - o] : 0ull, w = tid >= o ? s_last[tid - o] : 0ull;
        __syncthreads();
        s_part[tid] += v;
        s_last[tid] = s_last[tid] > w ? s_last[tid] : w;
        __syncthreads();
    }
    {
        uint64_t       at_list = s_part[tid] - cnt_mine;                // index of this thread's first member end in the step's list
        uint64_t       prev    = tid ? s_last[tid - 1u] : 0ull;         // position + 1 of the last member end before this thread's chunks
        const uint64_t run_in  = st->run_len;                           // bytes the open member had before the step
        for (uint32_t k = 0; k < per; ++k)
        {
            const uint32_t r = tid * per + k;
            if (r < R && ((s_lf[s_order[r]] >> 24) & GI_F_MEMBERS))
            {
                const GiChunk& c = chunks[slot_of(s_order[r])];
                const uint64_t t = real[r].text_off;
                for (uint32_t e = 0; e < c.n_mend; ++e)
                {
                    const uint64_t end = t + c.mend_sym[e];
                    const uint64_t len = prev ? end - (prev - 1u) : run_in + end;
                    if ((uint32_t)len != c.mend_isize[e])
                        atomicOr(&s_bad, 1u);
                    if (at_list < mlist_cap)
                    {
                        mlist_pos[at_list] = end;
                        mlist_crc[at_list] = c.mend_crc[e];
                    }
                    ++at_list;
                    prev = end + 1u;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0)
    {
        const uint64_t text_all = all & ((1ull << 40) - 1ull);
        uint32_t       reason   = st->reason;
        if (text_all > text_cap || (all >> 40) > work_cap)
            reason = GI_R_OVERFLOW;
        const uint64_t n_ml = s_part[1023], last = s_last[1023];
        if (s_bad && reason != GI_R_OVERFLOW)
            reason = GI_R_MEMBER;
        if (n_ml > mlist_cap && reason != GI_R_DATA && reason != GI_R_MEMBER)
            reason = GI_R_OVERFLOW;
        st->reason      = reason;
        st->n_real      = R;
        st->n_work      = (uint32_t)(all >> 40);
        st->text_off    = text_all;
        st->res_run_len = last ? text_all - (last - 1u) : st->run_len + text_all;
        st->res_members = s_members;
        st->n_mlist     = (uint32_t)(n_ml < 0xFFFFFFFFull ? n_ml : 0xFFFFFFFFull);
    }
}

#define GI_GROUP 32u // chain chunks whose windows one workgroup composes

__device__ __forceinline__ uint32_t gi_tail_sym(const GiChunk& c, const uint16_t* pool, uint32_t n, uint32_t i)
{
    // entry i of the chunk's tail function: the symbol that ends up at byte i of the window BEHIND the chunk, in terms of the window before it
    if (n >= GI_WINDOW || i >= GI_WINDOW - n)
    {
        const uint32_t sp = n >= GI_WINDOW ? n - GI_WINDOW + i : i - (GI_WINDOW - n);
        return pool[((uint64_t)c.piece[sp >> GI_PIECE_LOG2] << GI_PIECE_LOG2) | (sp & (GI_PIECE - 1u))];
    }
    return 256u + i + n;
}

// Windows, level 1: workgroup g composes the tail functions of chain chunks [32 g, 32 g + 32).  P_r = "window before chunk r in terms
// of the window before the group" (u16[32768]: a byte, or 256 + index) is stored for every chunk (the resolve pass reads markers through
// it), the group's whole function G_g at the end.
__global__ __launch_bounds__(1024) void gi_window_kernel(const GiState* st, const GiChunk* chunks, const GiReal* real, const uint16_t* pool,
                                                         uint16_t* p_store, uint16_t* g_store)
{
    __shared__ uint16_t P[GI_WINDOW];
    const uint32_t      tid = threadIdx.x, R = st->n_real;
    const uint32_t      r0 = blockIdx.x * GI_GROUP, r1 = min(R, r0 + GI_GROUP);
    if (r0 >= R)
        return;
#pragma unroll
    for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
        P[t * 1024u + tid] = (uint16_t)(256u + t * 1024u + tid);
    __syncthreads();
    for (uint32_t r = r0; r < r1; ++r)
    {
        const GiChunk& c = chunks[real[r].slot];
        const uint32_t n = real[r].out_len;
        if (c.markers)
        {
            uint32_t* dst = reinterpret_cast<uint32_t*>(p_store + (uint64_t)r * GI_WINDOW);
            for (uint32_t i = tid; i < GI_WINDOW / 2u; i += 1024u)
                dst[i] = reinterpret_cast<const uint32_t*>(P)[i];
        }
        uint16_t nw[GI_WINDOW / 1024u];
#pragma unroll
        for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
        {
            uint32_t v = gi_tail_sym(c, pool, n, t * 1024u + tid);
            if (v >= 256u)
                v = P[v - 256u];
            nw[t] = (uint16_t)v;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
            P[t * 1024u + tid] = nw[t];
        __syncthreads();
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(g_store + (uint64_t)blockIdx.x * GI_WINDOW);
    for (uint32_t i = tid; i < GI_WINDOW / 2u; i += 1024u)
        dst[i] = reinterpret_cast<const uint32_t*>(P)[i];
}

// Windows, level 2: one workgroup, group after group: W_g = the stream's window (bytes) before group g; the window behind the last group
// becomes the stream's.
__global__ __launch_bounds__(1024) void gi_wchain_kernel(const GiState* st, const uint16_t* g_store, uint8_t* window, uint8_t* w_store)
{
    __shared__ uint8_t W[GI_WINDOW];
    const uint32_t     tid = threadIdx.x, R = st->n_real, n_groups = (R + GI_GROUP - 1u) / GI_GROUP;
    for (uint32_t i = tid; i < GI_WINDOW / 16u; i += 1024u)
        reinterpret_cast<uint4*>(W)[i] = reinterpret_cast<const uint4*>(window)[i];
    __syncthreads();
    uint16_t cur[GI_WINDOW / 1024u], nxt[GI_WINDOW / 1024u];
    if (n_groups)
#pragma unroll
        for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
            cur[t] = g_store[t * 1024u + tid];
    for (uint32_t g = 0; g < n_groups; ++g)
    {
        if (g + 1 < n_groups)
#pragma unroll
            for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
                nxt[t] = g_store[(uint64_t)(g + 1) * GI_WINDOW + t * 1024u + tid];
        for (uint32_t i = tid; i < GI_WINDOW / 16u; i += 1024u)
            reinterpret_cast<uint4*>(w_store + (uint64_t)g * GI_WINDOW)[i] = reinterpret_cast<const uint4*>(W)[i];
        uint8_t nw[GI_WINDOW / 1024u];
#pragma unroll
        for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
        {
            const uint32_t v = cur[t];
            nw[t]            = (uint8_t)(v < 256u ? v : W[v - 256u]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < GI_WINDOW / 1024u; ++t)
        {
            W[t * 1024u + tid] = nw[t];
            cur[t]             = nxt[t];
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < GI_WINDOW / 16u; i += 1024u)
        reinterpret_cast<uint4*>(window)[i] = reinterpret_cast<const uint4*>(W)[i];
}

// symbols -> bytes.  One work item = one piece of one chain chunk; a marker goes through the chunk's P (level 1: staged in LDS, four
// markers in five symbols of FASTQ text take this turn) and, if that is a marker still, through its group's window (level 2).
__global__ __launch_bounds__(512) void gi_resolve_kernel(const GiState* st, const GiChunk* chunks, const GiReal* real, const uint2* work,
                                                         const uint16_t* pool, const uint16_t* p_store, const uint8_t* w_store, uint8_t* text)
{
    __shared__ uint16_t P[GI_WINDOW];
    const uint32_t      n_work = st->n_work;
    uint32_t            have = ~0u; // the chain chunk whose P is in LDS
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x)
    {
        const uint2     it = work[w];
        const GiReal    R  = real[it.x];
        const GiChunk&  c  = chunks[R.slot];
        const uint32_t  lo = it.y << GI_PIECE_LOG2, hi = min(R.out_len, lo + GI_PIECE);
        const uint16_t* s  = pool + ((uint64_t)c.piece[it.y] << GI_PIECE_LOG2);
        const uint8_t*  wg = w_store + (uint64_t)(it.x / GI_GROUP) * GI_WINDOW;
        uint8_t*        d  = text + R.text_off + lo;
        if (c.markers && have != it.x)
        {
            __syncthreads();
            const uint4* src = reinterpret_cast<const uint4*>(p_store + (uint64_t)it.x * GI_WINDOW);
            for (uint32_t i = threadIdx.x; i < GI_WINDOW / 8u; i += 512u)
                reinterpret_cast<uint4*>(P)[i] = src[i];
            have = it.x;
            __syncthreads();
        }
        const uint32_t cnt4 = (hi - lo) & ~3u;
        for (uint32_t i = threadIdx.x * 4u; i < cnt4; i += 2048u)
        {
            const uint2 two = *reinterpret_cast<const uint2*>(s + i); // (a piece begins on a 64 KiB boundary of the pool)
            uint32_t    v[4] = { two.x & 0xFFFFu, two.x >> 16, two.y & 0xFFFFu, two.y >> 16 };
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] >= 256u)
                {
                    v[k] = P[v[k] - 256u];
                    if (v[k] >= 256u)
                        v[k] = wg[v[k] - 256u];
                }
            const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            if (((uintptr_t)(d + i) & 3u) == 0u)
                *reinterpret_cast<uint32_t*>(d + i) = packed;
            else
            {
                d[i]     = (uint8_t)v[0];
                d[i + 1] = (uint8_t)v[1];
                d[i + 2] = (uint8_t)v[2];
                d[i + 3] = (uint8_t)v[3];
            }
        }
        for (uint32_t i = cnt4 + threadIdx.x; i < hi - lo; i += 512u)
        {
            uint32_t v = s[i];
            if (v >= 256u)
            {
                v = P[v - 256u];
                if (v >= 256u)
                    v = wg[v - 256u];
            }
            d[i] = (uint8_t)v;
        }
    }
}

// ---- where the records of a step's text begin: cut points for the caller's batches ----------------------------------------------------
#define GI_CUT_TILE 4096u
__device__ __forceinline__ uint32_t gi_nl_count(uint32_t x)
{
    const uint32_t t = x ^ 0x0A0A0A0Au; // a byte of t is 0 where the text has '\n'
    return (uint32_t)__popc(~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu));
}

__global__ __launch_bounds__(256) void gi_nl_tiles_kernel(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ cnt)
{
    __shared__ uint32_t part[4];
    const uint64_t      p = (uint64_t)blockIdx.x * GI_CUT_TILE + threadIdx.x * 16u;
    uint32_t            c = 0;
    if (p + 16u <= n)
    {
        const uint4 v = *reinterpret_cast<const uint4*>(text + p);
        c             = gi_nl_count(v.x) + gi_nl_count(v.y) + gi_nl_count(v.z) + gi_nl_count(v.w);
    }
    else
        for (uint64_t q = p; q < n; ++q)
            c += text[q] == '\n';
    for (int o = 32; o > 0; o >>= 1)
        c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0)
        part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// One wave per cut; its lanes look at 64 bytes at a time.
// newlines in text[from, to)
__device__ __forceinline__ uint64_t gi_count_nl_wave(const uint8_t* __restrict__ text, uint64_t from, uint64_t to)
{
    uint64_t c = 0;
    for (uint64_t q = from; q < to; q += 64u)
    {
        const uint64_t i = q + gi_lane();
        c += (uint64_t)__popcll(__ballot(i < to && text[i] == '\n'));
    }
    return c;
}
// offset of newline number `left` (from 0) at or behind text[from] (n: the text's size); n when there is none
__device__ __forceinline__ uint64_t gi_find_nl_wave(const uint8_t* __restrict__ text, uint64_t n, uint64_t from, uint64_t left)
{
    for (uint64_t q = from; q < n; q += 64u)
    {
        const uint64_t i = q + gi_lane();
        uint64_t       m = __ballot(i < n && text[i] == '\n');
        const uint64_t c = (uint64_t)__popcll(m);
        if (left < c)
        {
            for (uint64_t k = 0; k < left; ++k)
                m &= m - 1ull;
            return q + (uint64_t)__builtin_ctzll(m);
        }
        left -= c;
    }
    return n;
}
// the tile (index into pre, the exclusive sums of the tiles' newline counts) that holds newline number want
__device__ __forceinline__ uint32_t gi_tile_of(const uint32_t* __restrict__ pre, uint32_t tiles, uint64_t want)
{
    uint32_t lo = 0, hi = tiles;
    while (hi - lo > 1u)
    {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (pre[mid] <= want)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// wave j < m: the first record boundary at or behind (j + 1) * piece; wave m: the last record boundary of the text.  A record
// boundary = the byte behind newline number l (from 0) with (l + 1) % lpr == 0.  ~0 = none.
__global__ __launch_bounds__(64) void gi_cuts_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ pre /* tiles + 1: exclusive sums */,
                                                     uint32_t tiles, uint32_t lpr, uint64_t piece, uint32_t m,
                                                     unsigned long long* __restrict__ cuts /* m + 1 offsets, then m + 1 line counts */)
{
    const uint32_t j     = blockIdx.x;
    const uint64_t total = pre[tiles];
    uint64_t       want; // the newline behind which the cut lies
    bool           none = false;
    if (j == m)
    {
        none = total < lpr;
        want = none ? 0 : total / lpr * lpr - 1u;
    }
    else
    {
        const uint64_t T = (uint64_t)(j + 1) * piece; // (>= 1)
        none             = T > n;
        want             = 0;
        if (!none)
        {
            const uint64_t P    = T - 1u; // newlines in [0, P)
            const uint64_t t0   = P / GI_CUT_TILE;
            const uint64_t lmin = pre[t0] + gi_count_nl_wave(text, t0 * GI_CUT_TILE, P);
            want                = (lmin + lpr) / lpr * lpr - 1u;
            none                = want >= total;
        }
    }
    uint64_t q = 0;
    if (!none)
    {
        const uint32_t lo = gi_tile_of(pre, tiles, want);
        q                 = gi_find_nl_wave(text, n, (uint64_t)lo * GI_CUT_TILE, want - pre[lo]);
    }
    if (gi_lane() == 0)
    {
        cuts[j]          = none ? ~0ull : q + 1u;
        cuts[m + 1u + j] = none ? 0ull : want + 1u; // lines (newlines) in front of the cut
    }
}

// wave j: the offset behind line number lines[j] (a count of newlines from the text's first byte; 0 -> 0), ~0 when the text holds fewer
__global__ __launch_bounds__(64) void gi_cut_lines_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ pre, uint32_t tiles,
                                                          const unsigned long long* __restrict__ lines, uint32_t m, unsigned long long* __restrict__ out)
{
    const uint32_t j = blockIdx.x;
    if (j >= m)
        return;
    const uint64_t total = pre[tiles], ask = lines[j];
    uint64_t       r;
    if (ask == 0 || ask > total)
        r = ask == 0 ? 0ull : ~0ull;
    else
    {
        const uint64_t want = ask - 1u;
        const uint32_t lo   = gi_tile_of(pre, tiles, want);
        r                   = gi_find_nl_wave(text, n, (uint64_t)lo * GI_CUT_TILE, want - pre[lo]) + 1u;
    }
    if (gi_lane() == 0)
        out[j] = r;
}

// ---- CRC-32 of the members (RFC 1952: the trailer's first field) ------------------------------------------------------------------------
// The text of a step is cut on a 4 KiB grid and at the member ends; a thread takes a piece, computes its CRC-32 (byte-wise, table in LDS)
// and moves it to where its member ends: crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] mod the CRC polynomial (what zlib's
// crc32_combine computes; restated from its definition, checked against zlib in tests/test_gpu_inflate.py), so the pieces' values,
// each multiplied by x^(8 * bytes between its end and the member's end), XOR into the member's CRC in any order.  A member that began in
// an earlier step arrives as the CRC of what it had so far; one that goes on leaves the same behind.
__device__ const uint32_t kX2n[32] = { 0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u, 0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu, 0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu, 0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu }; // x^(2^k) mod p, reflected

__device__ __forceinline__ uint32_t gi_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;)
    {
        if (a & m)
        {
            p ^= b;
            if ((a & (m - 1u)) == 0u)
                break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}

// x^(n * 2^k) mod p
__device__ __forceinline__ uint32_t gi_x2nmodp(uint64_t n, uint32_t k)
{
    uint32_t p = 1u << 31;
    while (n)
    {
        if (n & 1u)
            p = gi_multmodp(kX2n[k & 31u], p);
        n >>= 1;
        ++k;
    }
    return p;
}

#define GI_CRC_PIECE 4096u
__global__ __launch_bounds__(256) void gi_crc_kernel(const uint8_t* __restrict__ text, uint64_t n, const unsigned long long* __restrict__ mend_pos,
                                                     uint32_t n_mend, uint32_t carry_crc, uint32_t* __restrict__ acc)
{
    __shared__ uint32_t tab[4][256]; // slicing by four: tab[k][i] = CRC of byte i followed by k zero bytes
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k)
            c = (c & 1u) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        tab[0][threadIdx.x] = c;
    }
    __syncthreads();
    for (int k = 1; k < 4; ++k)
    {
        const uint32_t c = tab[k - 1][threadIdx.x];
        tab[k][threadIdx.x] = (c >> 8) ^ tab[0][c & 0xFFu];
        __syncthreads();
    }
    const uint64_t b  = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t lo = b * GI_CRC_PIECE;
    if (b == 0 && carry_crc) // the member that was open when the step began
        atomicXor(&acc[0], gi_multmodp(gi_x2nmodp(n_mend ? mend_pos[0] : n, 3), carry_crc));
    if (lo >= n)
        return;
    const uint64_t hi = min(n, lo + GI_CRC_PIECE);
    // member of byte lo: the number of member ends at or before it
    uint32_t m = 0;
    {
        uint32_t a = 0, z = n_mend;
        while (a < z)
        {
            const uint32_t mid = a + (z - a) / 2u;
            if (mend_pos[mid] <= lo)
                a = mid + 1u;
            else
                z = mid;
        }
        m = a;
    }
    uint64_t cur = lo;
    while (cur < hi)
    {
        const uint64_t mend = m < n_mend ? mend_pos[m] : n;
        const uint64_t end  = min(hi, mend);
        uint32_t       c    = 0xFFFFFFFFu;
        uint64_t q = cur;
        for (; q < end && ((uintptr_t)(text + q) & 3u); ++q)
            c = tab[0][(c ^ text[q]) & 0xFFu] ^ (c >> 8);
        for (; q < end && ((uintptr_t)(text + q) & 15u) && q + 4u <= end; q += 4u)
        {
            c ^= *reinterpret_cast<const uint32_t*>(text + q);
            c = tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
        }
        // (sixteen bytes a load: a lane's piece is its own, so every load of a wave touches 64 lines -- four times fewer of them this way)
        for (; q + 16u <= end; q += 16u)
        {
            const uint4    v = *reinterpret_cast<const uint4*>(text + q);
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                c ^= w[k];
                c = tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
            }
        }
        for (; q + 4u <= end; q += 4u)
        {
            c ^= *reinterpret_cast<const uint32_t*>(text + q);
            c = tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
        }
        for (; q < end; ++q)
            c = tab[0][(c ^ text[q]) & 0xFFu] ^ (c >> 8);
        c = ~c;
        if (end > cur)
            atomicXor(&acc[m], gi_multmodp(gi_x2nmodp(mend - end, 3), c));
        cur = end;
        if (m < n_mend && cur == mend)
            ++m;
    }
}

__global__ void gi_crc_check_kernel(const uint32_t* __restrict__ acc, const uint32_t* __restrict__ want, uint32_t n_mend, uint32_t* __restrict__ result /* [0] bad [1] carry */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mend && acc[i] != want[i])
        atomicOr(&result[0], 1u);
    if (i == 0)
        result[1] = acc[n_mend];
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
#define GI_SETS 3 // decodes that may be in flight: the one a step is putting together and two ahead

struct gn_inflate
{
    int         device = 0;
    int         n_cu   = 256;
    hipStream_t st = nullptr, st_copy = nullptr, st_out = nullptr;
    uint64_t    total = 0;
    std::atomic<uint64_t> fed{ 0 }; // (gn_inflate_feed may run on another host thread than gn_inflate_step)
    uint64_t    fed_step = 0;       // what the running step sees of it
    uint32_t    chunk_bytes = 32768;
    uint64_t    step_bytes  = 256ull << 20;
    uint32_t    n_chunks_file = 0;
    uint32_t    next_chunk = 0; // first chunk of the file the next step decodes
    uint8_t*    d_comp = nullptr;
    // per step
    uint32_t  slots_cap = 0, fix_cap = 64;
    // Two sets of what a step's decode writes (chunk records, symbol pool, counters): the NEXT step's decode is launched on a stream
    // of its own as soon as its range is known, and runs beside this step's tail, order, window and resolve passes.
    GiChunk*  d_chunks_set[GI_SETS] = {};
    uint16_t* d_pool_set[GI_SETS]   = {};
    uint32_t* d_ctr_set[GI_SETS]    = {};
    hipStream_t st_dec[GI_SETS] = {}; // (one per set: the decodes run side by side, a later one fills what the tail of an earlier one leaves idle)
    hipEvent_t  ev_dec[GI_SETS][2] = {};
    int       n_sets = 0;
    uint32_t  launches = 0; // decodes launched so far (the first ones are smaller)
    struct Pend // a decode in flight (or done): q[0] is what the next gn_inflate_step finishes, q[1] what the one after it does, ... --
    {           // each launched on the assumption that the step before ends where its range does (almost always; otherwise dropped)
        bool     valid = false;
        uint32_t j0 = 0, j1 = 0;
        uint64_t fed = 0;
        int      set = 0;
    } q[GI_SETS];
    int nq = 0;
    int  set_in_use = -1;        // the set the running step reads
    bool ahead_from_next = false; // the step's true end is known: decodes ahead begin there
    GiChunk*  d_chunks = nullptr; // (the set of the step being finished)
    uint16_t* d_pool   = nullptr;
    uint32_t  pool_cap = 0;
    uint32_t* d_ctr    = nullptr; // [0] pool_next [1] work_next
    GiState*  d_state  = nullptr;
    GiState*  h_state  = nullptr; // page-locked
    uint8_t*  d_window = nullptr;
    uint16_t* d_p_store = nullptr; // per chain chunk: its P (64 KiB)
    uint16_t* d_g_store = nullptr; // per group: its function
    uint8_t*  d_w_store = nullptr; // per group: the window before it
    GiReal*   d_real = nullptr;
    uint2*    d_work = nullptr;
    uint32_t  work_cap = 0;
    uint8_t*  d_text[2] = { nullptr, nullptr };
    uint64_t  text_cap = 0;
    int       cur = 1; // buffer of the last step
    uint64_t  n_text_last = 0;
    uint64_t  carry = 0; // bytes at the end of the last step's text that the next step's text begins with (gn_inflate_set_carry)
    unsigned long long* d_mlist_pos = nullptr;
    uint32_t* d_mlist_crc = nullptr; // [mlist_cap] wanted, then [mlist_cap + 1] accumulated, then [2] result
    uint32_t  mlist_cap = 65536;
    uint32_t  crc_carry = 0; // CRC-32 of the open member's bytes so far
    uint32_t* h_crc = nullptr;
    uint32_t* d_cut_cnt = nullptr; // newline counts per tile, then their exclusive sums
    void*     d_cut_tmp = nullptr;
    size_t    cut_tmp_bytes = 0;
    unsigned long long* d_cuts = nullptr;
    uint32_t  cut_tiles_cap = 0, cuts_cap = 0;
    uint64_t  lines_of_step = ~0ull; // the step whose text the line index in d_cut_cnt describes
    bool      ended = false;
    // Several inflaters of the SAME file on different devices take its steps in turn (gn_inflate_set_turns / gn_inflate_handoff): the
    // decode of a step -- nine tenths of the work -- depends on nothing before it; what the next step needs of this one is the stream
    // position, the member's length and CRC so far, the 32 KiB window and the text the step's end cut off (the carried record).
    uint32_t  turns = 1, turn = 0;
    uint8_t*  d_carry_in = nullptr; // the carried text, when it came from another inflater
    uint64_t  carry_in_cap = 0;
    bool      carry_in = false;
    // totals
    gn_inflate_stats stats{};
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
};

static void gi_free(gn_inflate* z)
{
    if (!z)
        return;
    hipSetDevice(z->device);
    hipDeviceSynchronize();
    for (void* p : { (void*)z->d_comp, (void*)z->d_chunks_set[0], (void*)z->d_pool_set[0], (void*)z->d_ctr_set[0], (void*)z->d_chunks_set[1], (void*)z->d_pool_set[1], (void*)z->d_ctr_set[1],
                     (void*)z->d_chunks_set[2], (void*)z->d_pool_set[2], (void*)z->d_ctr_set[2], (void*)z->d_state, (void*)z->d_window, (void*)z->d_p_store, (void*)z->d_g_store, (void*)z->d_w_store,
                     (void*)z->d_real, (void*)z->d_work, (void*)z->d_mlist_pos, (void*)z->d_mlist_crc, (void*)z->d_cut_cnt, z->d_cut_tmp, (void*)z->d_cuts, (void*)z->d_text[0], (void*)z->d_text[1],
                     (void*)z->d_carry_in })
        if (p)
            hipFree(p);
    if (z->h_state)
        hipHostFree(z->h_state);
    if (z->h_crc)
        hipHostFree(z->h_crc);
    for (hipStream_t s : { z->st, z->st_copy, z->st_out, z->st_dec[0], z->st_dec[1], z->st_dec[2] })
        if (s)
            hipStreamDestroy(s);
    for (hipEvent_t e : z->ev)
        if (e)
            hipEventDestroy(e);
    for (auto& pair : z->ev_dec)
        for (hipEvent_t e : pair)
            if (e)
                hipEventDestroy(e);
    delete z;
}

extern "C" int gn_inflate_create(int device, uint64_t compressed_bytes, uint32_t chunk_bytes, uint64_t step_bytes, gn_inflate** out)
{
    if (!out || compressed_bytes < 18)
        return gn_fail(GN_EINVAL, "gn_inflate_create: a gzip file has at least 18 bytes");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev)
        return gn_fail(GN_ENODEV, "gn_inflate_create: no usable HIP device %d", device);
    if (compressed_bytes >= (1ull << 36))
        return gn_fail(GN_ERANGE, "gn_inflate_create: compressed files of 64 GiB and more are not taken");
    GN_HIP(hipSetDevice(device));
    gn_inflate* z = new (std::nothrow) gn_inflate();
    if (!z)
        return gn_fail(GN_ENOMEM, "gn_inflate_create: out of memory");
    z->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        z->n_cu = prop.multiProcessorCount;
    z->total       = compressed_bytes;
    z->chunk_bytes = chunk_bytes ? std::min<uint32_t>(std::max<uint32_t>(chunk_bytes & ~3u, 256u), 16u << 20) : 32768u; // (positions inside a chunk are 31-bit numbers)
    // (device memory is cleared when it is allocated -- 27 GB took 0.15 s, and several times that beside other allocations: the buffers of a
    //  file are sized to the file; steps of 128 MiB for files below 1.5 GB, 256 MiB above, where a tenth of a second does not count)
    z->step_bytes  = step_bytes ? std::max<uint64_t>(step_bytes, z->chunk_bytes) : (compressed_bytes < (1536ull << 20) ? 128ull << 20 : 256ull << 20);
    z->step_bytes  = (z->step_bytes + z->chunk_bytes - 1) / z->chunk_bytes * z->chunk_bytes;
    z->step_bytes  = std::min<uint64_t>(z->step_bytes, 8192ull * z->chunk_bytes); // (gi_order_kernel keeps a step's slots in LDS)
    z->n_chunks_file = (uint32_t)((compressed_bytes + z->chunk_bytes - 1) / z->chunk_bytes);
    const uint64_t step = std::min<uint64_t>(z->step_bytes, (uint64_t)z->n_chunks_file * z->chunk_bytes);
    z->slots_cap        = (uint32_t)(step / z->chunk_bytes);
    // text of a step: up to 8 x its compressed bytes (FASTQ: 3.5-6 x); beyond that the caller's host path takes the file
    z->text_cap = std::max<uint64_t>(step * 8u, 1u << 20) + (4u << 20);
    z->pool_cap = (uint32_t)(z->text_cap / GI_PIECE) + 2u * (z->slots_cap + z->fix_cap) + 64u;
    z->work_cap = z->pool_cap;
    auto fail = [&](hipError_t e, const char* what) {
        gi_free(z);
        return gn_fail(e == hipErrorOutOfMemory ? GN_ENOMEM : GN_ENODEV, "gn_inflate_create: %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
#define GI_TRY(x, what)                                                                                                                  \
    if ((e = (x)) != hipSuccess)                                                                                                         \
        return fail(e, what);
    GI_TRY(hipStreamCreateWithFlags(&z->st, hipStreamNonBlocking), "stream");
    GI_TRY(hipStreamCreateWithFlags(&z->st_copy, hipStreamNonBlocking), "stream");
    GI_TRY(hipStreamCreateWithFlags(&z->st_out, hipStreamNonBlocking), "stream");
    for (auto& ev : z->ev)
        GI_TRY(hipEventCreate(&ev), "event");
    const uint64_t comp_alloc = ((compressed_bytes + 3) & ~3ull) + 1024;
    GI_TRY(hipMalloc((void**)&z->d_comp, comp_alloc), "compressed bytes");
    GI_TRY(hipMemsetAsync(z->d_comp + (compressed_bytes & ~3ull), 0, comp_alloc - (compressed_bytes & ~3ull), z->st), "memset");
    for (int k = 0; k < GI_SETS; ++k)
    {
        // (a file of k steps needs k sets at most)
        if (k >= 1 && (uint64_t)z->n_chunks_file * z->chunk_bytes <= (uint64_t)k * step)
            break;
        z->n_sets = k + 1;
        GI_TRY(hipMalloc((void**)&z->d_chunks_set[k], (size_t)(z->slots_cap + z->fix_cap) * sizeof(GiChunk)), "chunk records");
        GI_TRY(hipMalloc((void**)&z->d_pool_set[k], (size_t)z->pool_cap * GI_PIECE * 2u), "symbol pool");
        GI_TRY(hipMalloc((void**)&z->d_ctr_set[k], 64), "counters");
        {
            // the decodes run at the lowest stream priority: whatever else the device has to do (the step's own small passes, the
            // classification, a filter being loaded) goes first when a slot frees up
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess)
                lo = 0;
            GI_TRY(hipStreamCreateWithPriority(&z->st_dec[k], hipStreamNonBlocking, lo), "stream");
        }
        for (auto& e2 : z->ev_dec[k])
            GI_TRY(hipEventCreate(&e2), "event");
    }
    GI_TRY(hipMalloc((void**)&z->d_state, sizeof(GiState)), "state");
    GI_TRY(hipHostMalloc((void**)&z->h_state, sizeof(GiState), hipHostMallocDefault), "state (host)");
    GI_TRY(hipMalloc((void**)&z->d_window, GI_WINDOW), "window");
    GI_TRY(hipMemsetAsync(z->d_window, 0, GI_WINDOW, z->st), "memset");
    {
        const size_t real_cap = z->slots_cap + z->fix_cap, groups = (real_cap + GI_GROUP - 1) / GI_GROUP;
        GI_TRY(hipMalloc((void**)&z->d_p_store, real_cap * GI_WINDOW * 2u), "windows");
        GI_TRY(hipMalloc((void**)&z->d_g_store, groups * GI_WINDOW * 2u), "windows");
        GI_TRY(hipMalloc((void**)&z->d_w_store, groups * GI_WINDOW), "windows");
    }
    GI_TRY(hipMalloc((void**)&z->d_real, (size_t)(z->slots_cap + z->fix_cap) * sizeof(GiReal)), "chain");
    GI_TRY(hipMalloc((void**)&z->d_work, (size_t)z->work_cap * sizeof(uint2)), "work list");
    GI_TRY(hipMalloc((void**)&z->d_mlist_pos, (size_t)z->mlist_cap * sizeof(unsigned long long)), "member list");
    GI_TRY(hipMalloc((void**)&z->d_mlist_crc, ((size_t)z->mlist_cap * 2u + 8u) * sizeof(uint32_t)), "member list");
    GI_TRY(hipHostMalloc((void**)&z->h_crc, 2 * sizeof(uint32_t), hipHostMallocDefault), "crc result");
    for (int b = 0; b < 2; ++b)
        GI_TRY(hipMalloc((void**)&z->d_text[b], z->text_cap + 64), "text");
    std::memset(z->h_state, 0, sizeof(GiState));
    GI_TRY(hipMemcpyAsync(z->d_state, z->h_state, sizeof(GiState), hipMemcpyHostToDevice, z->st), "state");
    GI_TRY(hipStreamSynchronize(z->st), "sync");
    GI_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gi_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GI_ORDER_MAX * 16u)), "LDS size");
#undef GI_TRY
    *out = z;
    return GN_OK;
}

extern "C" int gn_inflate_destroy(gn_inflate* z)
{
    gi_free(z);
    return GN_OK;
}

extern "C" int gn_inflate_feed(gn_inflate* z, const uint8_t* data, uint64_t n)
{
    if (!z || (!data && n))
        return gn_fail(GN_EINVAL, "gn_inflate_feed: null argument");
    const uint64_t fed0 = z->fed.load();
    if (fed0 + n > z->total)
        return gn_fail(GN_EINVAL, "gn_inflate_feed: more bytes than the file was announced with");
    GN_HIP(hipSetDevice(z->device));
    GN_HIP(hipMemcpyAsync(z->d_comp + fed0, data, n, hipMemcpyHostToDevice, z->st_copy));
    GN_HIP(hipStreamSynchronize(z->st_copy));
    z->fed.store(fed0 + n);
    return GN_OK;
}

static int gi_launch_chunks(gn_inflate* z, int set, hipStream_t st, uint64_t fed, uint32_t j0, uint32_t n, uint64_t fix_start, uint64_t fix_stop, uint32_t fix_slot)
{
    GiParams p;
    p.comp        = reinterpret_cast<const uint32_t*>(z->d_comp);
    p.avail_bits  = fed * 8u;
    p.total_bits  = z->total * 8u;
    p.chunk_bytes = z->chunk_bytes;
    p.j0          = j0;
    p.n           = n;
    p.chunks      = z->d_chunks_set[set];
    p.pool        = z->d_pool_set[set];
    p.pool_next   = z->d_ctr_set[set];
    p.pool_cap    = z->pool_cap;
    p.work_next   = z->d_ctr_set[set] + 1;
    p.fix_start   = fix_start;
    p.fix_stop    = fix_stop;
    p.fix_slot    = fix_slot;
    p.strict      = 1;
    const uint32_t grid = fix_start != GI_NONE ? 1u : n;
    hipLaunchKernelGGL(gi_chunk_kernel, dim3(grid), dim3(64), 0, st, p);
    GN_HIP(hipGetLastError());
    return GN_OK;
}

// Plans a step's chunk range, from chunk `from` on, out of what is fed, and launches its decode on the stream of set `set` (which no
// step still reads: the caller's business); the range
// goes to the first free entry of z->q.  must: nothing to launch is an error (the caller is gn_inflate_step itself, with nothing
// pending); otherwise it is simply not launched yet.
static int gi_start_decode(gn_inflate* z, bool must, uint32_t from, int set)
{
    if (z->nq >= GI_SETS)
        return GN_OK;
    gn_inflate::Pend& slot = z->q[z->nq];
    const uint64_t fed     = z->fed.load();
    const bool     all_fed = fed >= z->total;
    // chunks a step may decode: those whose range and a margin behind it are fed
    const uint64_t margin = std::min<uint64_t>(4ull << 20, std::max<uint64_t>(z->step_bytes / 4u, 2ull * z->chunk_bytes)); // (too little only costs a repeat)
    uint32_t       j1;
    if (all_fed)
        j1 = z->n_chunks_file;
    else
    {
        const uint64_t usable = fed > margin ? fed - margin : 0;
        j1                    = (uint32_t)(usable / z->chunk_bytes);
    }
    // the first step is half a step: as many chunks as the device holds waves of this kernel (a chunk takes its six milliseconds whatever
    // runs beside it, so fewer chunks would not bring the first text any sooner), and the caller's pipeline gets text after a sixth of a gigabyte
    // (of several inflaters taking turns only the first one's first step is the file's first)
    const uint32_t ramp = z->launches == 0 && z->turn == 0 ? std::max<uint32_t>(64u, z->slots_cap / 2u) : z->slots_cap;
    j1 = std::min<uint32_t>(j1, from + ramp);
    if (j1 <= from)
    {
        if (!must)
            return GN_OK;
        if (all_fed)
            return gn_fail(GN_ERANGE, "gn_inflate_step: the gzip stream does not end inside the file");
        return gn_fail(GN_EINVAL, "gn_inflate_step: feed more bytes first (a step needs its chunks and a margin behind them -- 4 MiB at the default sizes -- or the whole file)");
    }
    if (set < 0 || set >= GI_SETS || !z->d_chunks_set[set])
        return must ? gn_fail(GN_EINVAL, "gn_inflate_step: no decode set %d", set) : GN_OK; // (a one-step file has one set: nothing runs beside a step)
    GN_HIP(hipMemsetAsync(z->d_ctr_set[set], 0, 64, z->st_dec[set]));
    GN_HIP(hipEventRecord(z->ev_dec[set][0], z->st_dec[set]));
    const int rc = gi_launch_chunks(z, set, z->st_dec[set], fed, from, j1 - from, GI_NONE, 0, 0);
    if (rc != GN_OK)
        return rc;
    GN_HIP(hipEventRecord(z->ev_dec[set][1], z->st_dec[set]));
    slot.valid  = true;
    slot.j0     = from;
    slot.j1     = j1;
    slot.fed    = fed;
    slot.set    = set;
    ++z->nq;
    ++z->launches;
    return GN_OK;
}

extern "C" int gn_inflate_step(gn_inflate* z, uint64_t* n_text, int* done)
{
    if (!z || !n_text || !done)
        return gn_fail(GN_EINVAL, "gn_inflate_step: null argument");
    *n_text = 0;
    *done   = z->ended ? 1 : 0;
    if (z->ended)
        return GN_OK;
    GN_HIP(hipSetDevice(z->device));
    // a set no pending decode uses and no step reads (-1: none)
    auto free_set = [&](int busy) {
        for (int k = 0; k < z->n_sets; ++k)
        {
            bool used = k == busy;
            for (int i = 0; i < z->nq; ++i)
                used = used || z->q[i].set == k;
            if (!used)
                return k;
        }
        return -1;
    };
    if (z->nq == 0)
    {
        const int rc0 = gi_start_decode(z, true, z->next_chunk, free_set(-1));
        if (rc0 != GN_OK)
            return rc0;
    }
    // this step's decode is in flight (launched by an earlier step, or just now); the ones behind it start beside it
    const gn_inflate::Pend A = z->q[0];
    for (int i = 1; i < z->nq; ++i)
        z->q[i - 1] = z->q[i];
    --z->nq;
    auto launch_ahead = [&]() {
        while (!gn_sw().inflate_ahead)
        {
            // (taking turns: this inflater's next step begins behind the other inflaters' steps -- a full range each, as far as one can tell
            //  now; gn_inflate_handoff drops what was decoded from a wrong guess)
            const uint32_t from = (z->nq ? z->q[z->nq - 1].j1 : (z->ahead_from_next ? z->next_chunk : A.j1)) + (z->turns - 1u) * z->slots_cap;
            const int      k    = free_set(z->set_in_use);
            const int      had  = z->nq;
            if (from >= z->n_chunks_file || k < 0 || gi_start_decode(z, false, from, k) != GN_OK || z->nq == had)
                break;
        }
    };
    z->set_in_use      = A.set;
    z->ahead_from_next = false;
    launch_ahead();
    const int set      = A.set;
    z->fed_step        = A.fed;
    const bool all_fed = z->fed_step >= z->total;
    const uint32_t j0 = A.j0, j1 = A.j1, n = j1 - j0;
    z->d_chunks       = z->d_chunks_set[set];
    z->d_pool         = z->d_pool_set[set];
    const int      buf = 1 - z->cur;
    const uint64_t carry = z->carry; // (<= n_text_last, checked by gn_inflate_set_carry)
    auto           t0 = std::chrono::steady_clock::now();
    GN_HIP(hipStreamWaitEvent(z->st, z->ev_dec[set][1], 0));
    GN_HIP(hipEventRecord(z->ev[1], z->st));
    int rc = GN_OK;
    // order of the stream: the host copy of the state is what the last step left; the kernel reads pos_bit / run_len and writes results
    uint32_t       fixes = 0;
    const uint64_t range_end = (uint64_t)j1 * z->chunk_bytes * 8u;
    for (;;)
    {
        z->h_state->res_markers = 0;
        std::memset(z->h_state->prof, 0, sizeof(z->h_state->prof));
        GN_HIP(hipMemcpyAsync(z->d_state, z->h_state, sizeof(GiState), hipMemcpyHostToDevice, z->st));
        const size_t lds = (size_t)(n + fixes) * 16u;
        hipLaunchKernelGGL(gi_order_kernel, dim3(1), dim3(1024), lds, z->st, z->d_state, z->d_chunks, n, z->slots_cap, fixes, j0, z->chunk_bytes, range_end,
                           z->total * 8u, z->d_real, z->d_work, z->slots_cap + z->fix_cap, z->work_cap, z->text_cap - carry, z->d_mlist_pos, z->d_mlist_crc, z->mlist_cap);
        GN_HIP(hipGetLastError());
        GN_HIP(hipMemcpyAsync(z->h_state, z->d_state, sizeof(GiState), hipMemcpyDeviceToHost, z->st));
        GN_HIP(hipStreamSynchronize(z->st));
        const GiState& s = *z->h_state;
        if (s.reason != GI_R_GAP)
            break;
        if (fixes >= z->fix_cap || z->stats.fixups + fixes > 16u + z->stats.chunks / 8u)
            return gn_fail(GN_ERANGE, "gn_inflate_step: too many positions the block search does not find (%u in this step): not a file for this decoder",
                           fixes);
        rc = gi_launch_chunks(z, set, z->st, z->fed_step, 0, 1, s.res_pos, s.gap_stop, z->slots_cap + fixes);
        if (rc != GN_OK)
            return rc;
        ++fixes;
    }
    GN_HIP(hipEventRecord(z->ev[2], z->st));
    const GiState s = *z->h_state;
    if (s.reason == GI_R_DATA)
        return gn_fail(GN_ERANGE, "gn_inflate_step: damaged or truncated gzip stream near compressed byte %llu", (unsigned long long)(s.res_pos >> 3));
    if (s.reason == GI_R_MEMBER)
        return gn_fail(GN_ERANGE, "gn_inflate_step: gzip member with a wrong length (ISIZE)");
    if (s.reason == GI_R_OVERFLOW)
        return gn_fail(GN_ERANGE, "gn_inflate_step: the data expands beyond what a step holds (more than 8-fold, or a chunk beyond 2 Mi symbols)");
    // where the next step begins is known now: its decode starts beside this step's remaining passes
    z->next_chunk = s.reason == GI_R_INPUT ? j0 + std::min<uint32_t>(s.cursor, n) : j1;
    if (z->turns == 1 && z->nq && z->q[0].j0 != z->next_chunk) // the decodes that ran ahead assumed another start: dropped (their sets are free again when they are through)
        z->nq = 0;
    z->ahead_from_next = true;
    if (s.n_real)
    {
        const uint32_t groups = (s.n_real + GI_GROUP - 1u) / GI_GROUP;
        hipLaunchKernelGGL(gi_window_kernel, dim3(groups), dim3(1024), 0, z->st, z->d_state, z->d_chunks, z->d_real, z->d_pool, z->d_p_store, z->d_g_store);
        GN_HIP(hipGetLastError());
        hipLaunchKernelGGL(gi_wchain_kernel, dim3(1), dim3(1024), 0, z->st, z->d_state, z->d_g_store, z->d_window, z->d_w_store);
        GN_HIP(hipGetLastError());
    }
    if (carry)
        GN_HIP(hipMemcpyAsync(z->d_text[buf], z->carry_in ? z->d_carry_in : z->d_text[z->cur] + (z->n_text_last - carry), carry, hipMemcpyDeviceToDevice, z->st));
    z->carry_in = false;
    if (s.n_work)
    {
        hipLaunchKernelGGL(gi_resolve_kernel, dim3(std::min<uint32_t>(s.n_work, (uint32_t)z->n_cu * 2u)), dim3(512), 0, z->st, z->d_state, z->d_chunks, z->d_real,
                           z->d_work, z->d_pool, z->d_p_store, z->d_w_store, z->d_text[buf] + carry);
        GN_HIP(hipGetLastError());
    }
    {
        uint32_t*      acc    = z->d_mlist_crc + z->mlist_cap;
        uint32_t*      result = acc + z->mlist_cap + 1u;
        const uint32_t M      = s.n_mlist;
        GN_HIP(hipMemsetAsync(acc, 0, ((size_t)z->mlist_cap + 3u) * sizeof(uint32_t), z->st));
        const uint32_t blocks = (uint32_t)((s.text_off + (uint64_t)GI_CRC_PIECE * 256u - 1u) / ((uint64_t)GI_CRC_PIECE * 256u));
        hipLaunchKernelGGL(gi_crc_kernel, dim3(std::max(1u, blocks)), dim3(256), 0, z->st, z->d_text[buf] + carry, s.text_off, z->d_mlist_pos, M, z->crc_carry, acc);
        hipLaunchKernelGGL(gi_crc_check_kernel, dim3((M + 256u) / 256u), dim3(256), 0, z->st, acc, z->d_mlist_crc, M, result);
        GN_HIP(hipGetLastError());
        GN_HIP(hipMemcpyAsync(z->h_crc, result, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, z->st));
    }
    GN_HIP(hipEventRecord(z->ev[3], z->st));
    GN_HIP(hipStreamSynchronize(z->st));
    if (z->h_crc[0])
        return gn_fail(GN_ERANGE, "gn_inflate_step: gzip member with a wrong CRC-32");
    z->crc_carry = z->h_crc[1];
    float ms = 0;
    if (hipEventElapsedTime(&ms, z->ev_dec[set][0], z->ev_dec[set][1]) == hipSuccess)
        z->stats.ms_decode += ms;
    // this step's set is free: one more decode ahead
    z->set_in_use = -1;
    if (s.reason != GI_R_END && !(s.reason == GI_R_INPUT && all_fed))
        launch_ahead();
    if (hipEventElapsedTime(&ms, z->ev[1], z->ev[2]) == hipSuccess)
        z->stats.ms_chain += ms;
    if (hipEventElapsedTime(&ms, z->ev[2], z->ev[3]) == hipSuccess)
        z->stats.ms_resolve += ms;
    z->stats.ms_step_wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    z->stats.steps += 1;
    z->stats.chunks += s.n_real;
    z->stats.fixups += fixes;
    for (int k = 0; k < 8; ++k)
        z->stats.prof_ms[k] += (double)s.prof[k] * 1e-5; // 10 ns ticks
    z->stats.markers += s.res_markers;
    z->stats.members += s.res_members;
    z->h_state->pos_bit = s.res_pos;
    z->h_state->run_len = s.res_run_len;
    z->stats.text_bytes += s.text_off;
    z->cur         = buf;
    z->n_text_last = carry + s.text_off;
    z->carry       = 0;
    *n_text        = carry + s.text_off;
    if (s.reason == GI_R_END)
    {
        z->ended = true;
        *done    = 1;
        return GN_OK;
    }
    if (s.reason == GI_R_INPUT)
    {
        // cursor = the slot that has to be decoded again with more bytes
        if (all_fed)
            return gn_fail(GN_ERANGE, "gn_inflate_step: truncated gzip stream");
        return GN_OK;
    }
    // the range is done
    if (j1 >= z->n_chunks_file)
    {
        if (s.res_pos >= z->total * 8u)
            return gn_fail(GN_ERANGE, "gn_inflate_step: truncated gzip stream");
        // (the stream stands before the file's end with every chunk consumed: handled above as a gap)
    }
    return GN_OK;
}

extern "C" int gn_inflate_set_turns(gn_inflate* z, uint32_t n_turns, uint32_t my_turn)
{
    if (!z || n_turns == 0 || my_turn >= n_turns)
        return gn_fail(GN_EINVAL, "gn_inflate_set_turns: need my_turn < n_turns");
    if (z->stats.steps || z->launches)
        return gn_fail(GN_EINVAL, "gn_inflate_set_turns: the inflater has begun");
    z->turns = n_turns;
    z->turn  = my_turn;
    return GN_OK;
}

extern "C" int gn_inflate_handoff(gn_inflate* from, gn_inflate* to)
{
    if (!from || !to || from == to)
        return gn_fail(GN_EINVAL, "gn_inflate_handoff: two inflaters, please");
    if (from->total != to->total || from->chunk_bytes != to->chunk_bytes || from->slots_cap != to->slots_cap)
        return gn_fail(GN_EINVAL, "gn_inflate_handoff: the inflaters were not created for the same file with the same chunk and step sizes");
    to->ended = from->ended;
    if (from->ended)
        return GN_OK;
    // (the step of `from` is through: gn_inflate_step returns after its stream's last pass)
    to->h_state->pos_bit = from->h_state->pos_bit;
    to->h_state->run_len = from->h_state->run_len;
    to->crc_carry        = from->crc_carry;
    to->next_chunk       = from->next_chunk;
    GN_HIP(hipSetDevice(to->device));
    // (on the receiving inflater's stream: a device-to-device copy on the null stream returns before it is done and orders nothing against a
    //  non-blocking stream -- the step's window pass would race it; the source is at rest, its inflater's step has synchronised)
    GN_HIP(hipMemcpyPeerAsync(to->d_window, to->device, from->d_window, from->device, GI_WINDOW, to->st));
    const uint64_t c = from->carry;
    if (c)
    {
        if (c > to->text_cap / 4u)
            return gn_fail(GN_EINVAL, "gn_inflate_handoff: %llu carried bytes are more than a quarter of a step's capacity", (unsigned long long)c);
        if (to->carry_in_cap < c)
        {
            if (to->d_carry_in)
                hipFree(to->d_carry_in);
            to->d_carry_in   = nullptr;
            to->carry_in_cap = 0;
            const uint64_t cap = std::max<uint64_t>(c + c / 2u, 1u << 20);
            if (hipMalloc((void**)&to->d_carry_in, cap) != hipSuccess)
                return gn_fail(GN_ENOMEM, "gn_inflate_handoff: no room for %llu carried bytes", (unsigned long long)c);
            to->carry_in_cap = cap;
        }
        GN_HIP(hipMemcpyPeerAsync(to->d_carry_in, to->device, from->d_text[from->cur] + (from->n_text_last - c), from->device, c, to->st));
    }
    to->carry    = c;
    to->carry_in = c != 0;
    from->carry  = 0; // (its next step's carry comes from the inflater before it)
    // what `to` decoded ahead for this step assumed a start: kept only if that is where the stream stands
    if (to->nq && to->q[0].j0 != to->next_chunk)
        to->nq = 0;
    return GN_OK;
}

extern "C" int gn_inflate_text(gn_inflate* z, uint8_t* dst, uint64_t off, uint64_t n)
{
    if (!z || (!dst && n))
        return gn_fail(GN_EINVAL, "gn_inflate_text: null argument");
    if (off + n > z->n_text_last)
        return gn_fail(GN_EINVAL, "gn_inflate_text: beyond the text of the last step");
    GN_HIP(hipSetDevice(z->device));
    GN_HIP(hipMemcpyAsync(dst, z->d_text[z->cur] + off, n, hipMemcpyDeviceToHost, z->st_out));
    GN_HIP(hipStreamSynchronize(z->st_out));
    return GN_OK;
}

extern "C" int gn_inflate_text_device(gn_inflate* z, const uint8_t** text, uint64_t* n)
{
    if (!z || !text || !n)
        return gn_fail(GN_EINVAL, "gn_inflate_text_device: null argument");
    *text = z->d_text[z->cur];
    *n    = z->n_text_last;
    return GN_OK;
}

extern "C" int gn_inflate_get_stats(gn_inflate* z, gn_inflate_stats* out)
{
    if (!z || !out)
        return gn_fail(GN_EINVAL, "gn_inflate_get_stats: null argument");
    *out = z->stats;
    return GN_OK;
}

extern "C" int gn_inflate_set_carry(gn_inflate* z, uint64_t n_tail)
{
    if (!z)
        return gn_fail(GN_EINVAL, "gn_inflate_set_carry: null argument");
    if (n_tail > z->n_text_last || n_tail > z->text_cap / 4u)
        return gn_fail(GN_EINVAL, "gn_inflate_set_carry: %llu bytes are more than the last step's text (or a quarter of a step's capacity) holds",
                       (unsigned long long)n_tail);
    z->carry = n_tail;
    return GN_OK;
}

// newline counts per 4 KiB tile of the last step's text and their exclusive sums (once per step)
static int gi_line_index(gn_inflate* z, uint32_t m_cuts)
{
    const uint64_t n     = z->n_text_last;
    const uint32_t tiles = (uint32_t)((n + GI_CUT_TILE - 1) / GI_CUT_TILE);
    if (tiles + 1u > z->cut_tiles_cap)
    {
        if (z->d_cut_cnt)
            hipFree(z->d_cut_cnt);
        if (z->d_cut_tmp)
            hipFree(z->d_cut_tmp);
        z->d_cut_cnt = nullptr;
        z->d_cut_tmp = nullptr;
        const uint32_t want = (uint32_t)((z->text_cap + GI_CUT_TILE - 1) / GI_CUT_TILE) + 2u;
        GN_HIP(hipMalloc((void**)&z->d_cut_cnt, (size_t)want * 2u * sizeof(uint32_t)));
        size_t tmp = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, z->d_cut_cnt, z->d_cut_cnt, (int)want, z->st);
        z->cut_tmp_bytes = tmp + 256;
        GN_HIP(hipMalloc(&z->d_cut_tmp, z->cut_tmp_bytes));
        z->cut_tiles_cap = want;
    }
    if (3u * m_cuts + 8u > z->cuts_cap)
    {
        if (z->d_cuts)
            hipFree(z->d_cuts);
        z->d_cuts = nullptr;
        GN_HIP(hipMalloc((void**)&z->d_cuts, (size_t)(3u * m_cuts + 256u) * sizeof(unsigned long long)));
        z->cuts_cap = 3u * m_cuts + 256u;
    }
    if (z->lines_of_step == z->stats.steps)
        return GN_OK;
    uint32_t* cnt = z->d_cut_cnt;
    uint32_t* pre = z->d_cut_cnt + z->cut_tiles_cap;
    GN_HIP(hipMemsetAsync(cnt + tiles, 0, sizeof(uint32_t), z->st));
    if (tiles)
        hipLaunchKernelGGL(gi_nl_tiles_kernel, dim3(tiles), dim3(256), 0, z->st, z->d_text[z->cur], n, cnt);
    size_t tmp = z->cut_tmp_bytes;
    GN_HIP(hipcub::DeviceScan::ExclusiveSum(z->d_cut_tmp, tmp, cnt, pre, (int)(tiles + 1u), z->st));
    GN_HIP(hipGetLastError());
    z->lines_of_step = z->stats.steps;
    return GN_OK;
}

extern "C" int gn_inflate_cuts_lines(gn_inflate* z, uint32_t lines_per_record, uint64_t piece_bytes, uint64_t* cuts, uint64_t* cut_lines, uint32_t cap,
                                     uint32_t* n_cuts)
{
    if (!z || !cuts || !n_cuts || lines_per_record == 0 || piece_bytes == 0)
        return gn_fail(GN_EINVAL, "gn_inflate_cuts: bad argument");
    *n_cuts = 0;
    const uint64_t n = z->n_text_last;
    if (n == 0)
        return GN_OK;
    GN_HIP(hipSetDevice(z->device));
    const uint32_t tiles = (uint32_t)((n + GI_CUT_TILE - 1) / GI_CUT_TILE);
    const uint32_t m     = (uint32_t)(n / piece_bytes);
    if (m + 1u > cap)
        return gn_fail(GN_EINVAL, "gn_inflate_cuts: room for %u cuts, %u needed", cap, m + 1u);
    const int rc = gi_line_index(z, m + 1u);
    if (rc != GN_OK)
        return rc;
    uint32_t* pre = z->d_cut_cnt + z->cut_tiles_cap;
    hipLaunchKernelGGL(gi_cuts_kernel, dim3(m + 1u), dim3(64), 0, z->st, z->d_text[z->cur], n, pre, tiles, lines_per_record, piece_bytes, m, z->d_cuts);
    GN_HIP(hipGetLastError());
    std::vector<unsigned long long> h(2u * (m + 1u));
    GN_HIP(hipMemcpyAsync(h.data(), z->d_cuts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, z->st));
    GN_HIP(hipStreamSynchronize(z->st));
    uint64_t last = 0;
    uint32_t k    = 0;
    for (uint32_t j = 0; j <= m; ++j)
        if (h[j] != ~0ull && h[j] > last && h[j] <= n)
        {
            if (cut_lines)
                cut_lines[k] = h[m + 1u + j];
            cuts[k++] = h[j];
            last      = h[j];
        }
    *n_cuts = k;
    return GN_OK;
}

extern "C" int gn_inflate_cuts(gn_inflate* z, uint32_t lines_per_record, uint64_t piece_bytes, uint64_t* cuts, uint32_t cap, uint32_t* n_cuts)
{
    return gn_inflate_cuts_lines(z, lines_per_record, piece_bytes, cuts, nullptr, cap, n_cuts);
}

extern "C" int gn_inflate_cut_at_lines(gn_inflate* z, const uint64_t* lines, uint32_t n_lines, uint64_t* offsets, uint64_t* total_lines)
{
    if (!z || (!lines && n_lines) || (!offsets && n_lines) || !total_lines)
        return gn_fail(GN_EINVAL, "gn_inflate_cut_at_lines: null argument");
    *total_lines     = 0;
    const uint64_t n = z->n_text_last;
    GN_HIP(hipSetDevice(z->device));
    const uint32_t tiles = (uint32_t)((n + GI_CUT_TILE - 1) / GI_CUT_TILE);
    const int      rc    = gi_line_index(z, n_lines + 1u);
    if (rc != GN_OK)
        return rc;
    uint32_t*           pre  = z->d_cut_cnt + z->cut_tiles_cap;
    unsigned long long* d_in = z->d_cuts + (n_lines + 8u);
    if (n_lines)
    {
        GN_HIP(hipMemcpyAsync(d_in, lines, (size_t)n_lines * sizeof(unsigned long long), hipMemcpyHostToDevice, z->st));
        hipLaunchKernelGGL(gi_cut_lines_kernel, dim3(n_lines), dim3(64), 0, z->st, z->d_text[z->cur], n, pre, tiles, d_in, n_lines, z->d_cuts);
        GN_HIP(hipGetLastError());
        GN_HIP(hipMemcpyAsync(offsets, z->d_cuts, (size_t)n_lines * sizeof(unsigned long long), hipMemcpyDeviceToHost, z->st));
    }
    uint32_t total32 = 0;
    GN_HIP(hipMemcpyAsync(&total32, pre + tiles, sizeof(uint32_t), hipMemcpyDeviceToHost, z->st));
    GN_HIP(hipStreamSynchronize(z->st));
    *total_lines = total32;
    return GN_OK;
}
