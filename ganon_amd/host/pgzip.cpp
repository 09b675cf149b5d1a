// pgzip.cpp -- see pgzip.hpp.  DEFLATE per RFC 1951, gzip framing per RFC 1952; written for this reader (a decoder that can
// start inside a stream and emit markers for the window it does not have is not something zlib offers).
#include "pgzip.hpp"

#include <zlib.h> // crc32 / crc32_combine only

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gnhost
{
namespace
{

constexpr uint64_t kNone   = ~0ull;
constexpr uint32_t kWindow = 32768;
constexpr size_t   kMaxChunkSymbols = 96u << 20; // symbols one decode keeps (192 MiB): bounds memory on data that inflates 100-fold

// ---- bits, least significant first --------------------------------------------------------------------------------
struct Bits
{
    const uint8_t* base;
    const uint8_t* p;
    const uint8_t* end;
    uint64_t       buf = 0;
    unsigned       cnt = 0;   // valid bits in buf
    bool           over = false; // more bits were asked for than the input holds

    Bits(const uint8_t* data, size_t size, uint64_t bitpos) : base(data), p(data + (bitpos >> 3)), end(data + size)
    {
        if (p > end)
            p = end;
        refill();
        const unsigned skip = (unsigned)(bitpos & 7);
        if (skip)
            drop(skip);
    }
    uint64_t pos() const { return (uint64_t)(p - base) * 8 - cnt; }
    inline void refill()
    {
        if (p + 8 <= end)
        {
            uint64_t w;
            std::memcpy(&w, p, 8);
            buf |= w << cnt;
            const unsigned adv = (63u - cnt) >> 3;
            p += adv;
            cnt += adv * 8;
        }
        else
            while (cnt <= 56 && p < end)
            {
                buf |= (uint64_t)*p++ << cnt;
                cnt += 8;
            }
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void     drop(unsigned n)
    {
        if (n > cnt)
        {
            over = true;
            n    = cnt;
        }
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t get(unsigned n)
    {
        if (cnt < n)
            refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    void align_byte() { drop(cnt & 7); }
};

// ---- Huffman decoding tables: a primary table of kPrimary bits, longer codes through per-prefix subtables ------------------
struct Entry
{
    uint16_t sym;  // symbol, or offset of the subtable
    uint8_t  len;  // code length (bits to drop); for a subtable link: primary bits to drop
    uint8_t  sub;  // 0: a symbol.  otherwise: index bits of the subtable
};

struct Table
{
    std::vector<Entry> e;
    unsigned           primary = 0;
    bool               single = false; // (a distance code of one symbol: RFC 1951 allows it, zlib accepts it)

    // lens[0..n): code lengths 0..15.  Returns false for an over-subscribed or incomplete set, like zlib's inflate_table.
    // incomplete: 0 = never accepted, 1 = a single code of length 1 is (distance codes), 2 = any (the fixed distance code:
    // 30 of 32 five-bit patterns)
    bool build(const uint8_t* lens, unsigned n, unsigned primary_bits, int incomplete)
    {
        unsigned count[16] = { 0 };
        for (unsigned i = 0; i < n; ++i)
            count[lens[i]]++;
        if (count[0] == n)
            return false;
        int left = 1;
        for (unsigned l = 1; l <= 15; ++l)
        {
            left <<= 1;
            left -= (int)count[l];
            if (left < 0)
                return false;
        }
        single = false;
        if (left > 0)
        {
            if (!(incomplete == 2 || (incomplete == 1 && n - count[0] == 1 && count[1] == 1)))
                return false;
            single = true;
        }
        unsigned maxlen = 15;
        while (maxlen > 1 && count[maxlen] == 0)
            --maxlen;
        primary = std::min(primary_bits, maxlen);
        uint16_t next_code[16];
        unsigned code = 0;
        count[0]      = 0;
        for (unsigned l = 1; l <= 15; ++l)
        {
            code         = (code + count[l - 1]) << 1;
            next_code[l] = (uint16_t)code;
        }
        // subtables: one per primary-bit prefix that has longer codes; its width = longest code with that prefix - primary
        const size_t       P = (size_t)1 << primary;
        std::vector<uint8_t> sub_bits(P, 0);
        std::vector<uint16_t> codes(n, 0);
        for (unsigned i = 0; i < n; ++i)
        {
            const unsigned l = lens[i];
            if (!l)
                continue;
            const unsigned c = next_code[l]++;
            // bit-reverse the code: the stream delivers it least significant bit first
            unsigned r = 0;
            for (unsigned b = 0; b < l; ++b)
                r |= ((c >> b) & 1u) << (l - 1 - b);
            codes[i] = (uint16_t)r;
            if (l > primary)
            {
                const unsigned pre = r & (P - 1);
                sub_bits[pre]      = std::max<uint8_t>(sub_bits[pre], (uint8_t)(l - primary));
            }
        }
        size_t total = P;
        std::vector<uint32_t> sub_at(P, 0);
        for (size_t pre = 0; pre < P; ++pre)
            if (sub_bits[pre])
            {
                sub_at[pre] = (uint32_t)total;
                total += (size_t)1 << sub_bits[pre];
            }
        if (total > 0xFFFF)
            return false;
        e.assign(total, Entry{ 0xFFFF, 0, 0 }); // (0xFFFF/len 0 = no code: only reachable with the single-code set)
        for (size_t pre = 0; pre < P; ++pre)
            if (sub_bits[pre])
                e[pre] = Entry{ (uint16_t)sub_at[pre], (uint8_t)primary, sub_bits[pre] };
        for (unsigned i = 0; i < n; ++i)
        {
            const unsigned l = lens[i];
            if (!l)
                continue;
            const unsigned r = codes[i];
            if (l <= primary)
                for (size_t x = r; x < P; x += (size_t)1 << l)
                    e[x] = Entry{ (uint16_t)i, (uint8_t)l, 0 };
            else
            {
                const unsigned pre = r & (P - 1), sb = sub_bits[pre], hi = r >> primary, hl = l - primary;
                for (size_t x = hi; x < ((size_t)1 << sb); x += (size_t)1 << hl)
                    e[sub_at[pre] + x] = Entry{ (uint16_t)i, (uint8_t)hl, 0 };
            }
        }
        return true;
    }

    // decodes one symbol; 0xFFFF = invalid code
    inline unsigned decode(Bits& in) const
    {
        Entry x = e[in.buf & (((uint64_t)1 << primary) - 1)];
        if (x.sub)
        {
            in.drop(x.len);
            x = e[x.sym + (in.buf & (((uint64_t)1 << x.sub) - 1))];
        }
        in.drop(x.len);
        return x.len ? x.sym : 0xFFFFu;
    }
};

const uint16_t kLenBase[29]  = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t  kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t  kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
const uint8_t  kClOrder[19]  = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

struct BlockTables
{
    Table lit, dist;
    bool  have_dist = true;
};

const BlockTables& fixed_tables()
{
    static const BlockTables t = [] {
        BlockTables b;
        uint8_t     l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        b.lit.build(l, 288, 10, 0);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        b.dist.build(d, 30, 8, 2); // (30 of 32 five-bit codes: incomplete by definition; the two other patterns stay invalid)
        return b;
    }();
    return t;
}

// The dynamic block header at the reader's position (after BFINAL/BTYPE).  false = not a valid header.
bool read_dynamic_header(Bits& in, BlockTables& t)
{
    in.refill();
    const unsigned hlit = in.get(5) + 257, hdist = in.get(5) + 1, hclen = in.get(4) + 4;
    if (hlit > 286 || hdist > 30)
        return false;
    uint8_t cl[19] = { 0 };
    for (unsigned i = 0; i < hclen; ++i)
        cl[kClOrder[i]] = (uint8_t)in.get(3);
    Table pre;
    if (!pre.build(cl, 19, 7, 0))
        return false;
    uint8_t  lens[286 + 30];
    unsigned i = 0;
    while (i < hlit + hdist)
    {
        in.refill();
        const unsigned s = pre.decode(in);
        if (s < 16)
            lens[i++] = (uint8_t)s;
        else if (s == 16)
        {
            if (i == 0)
                return false;
            unsigned r = 3 + in.get(2);
            if (i + r > hlit + hdist)
                return false;
            const uint8_t v = lens[i - 1];
            while (r--)
                lens[i++] = v;
        }
        else if (s == 17 || s == 18)
        {
            unsigned r = s == 17 ? 3 + in.get(3) : 11 + in.get(7);
            if (i + r > hlit + hdist)
                return false;
            while (r--)
                lens[i++] = 0;
        }
        else
            return false;
        if (in.over)
            return false;
    }
    if (lens[256] == 0) // no end-of-block code
        return false;
    if (!t.lit.build(lens, hlit, 11, 0))
        return false;
    // a block of literals only may come with an empty distance code (one zero length)
    bool any = false;
    for (unsigned d = 0; d < hdist; ++d)
        any = any || lens[hlit + d] != 0;
    t.have_dist = any;
    if (any && !t.dist.build(lens + hlit, hdist, 8, 1))
        return false;
    return true;
}

struct SymBuf // 16-bit symbols: < 256 a byte, >= 256 marker for byte (v - 256) of the 32 KiB before the chunk
{
    std::vector<uint16_t> v;
    size_t                n = 0;
    void                  need(size_t more)
    {
        if (n + more > v.size())
            v.resize(std::max(v.size() * 2, n + more + (1u << 20)));
    }
};

enum class BlockEnd { ok, bad };

// One compressed block's symbols (Huffman-coded kinds).  STRICT: the block finder's run -- literals must look like text and
// nothing is kept beyond `out`'s scratch.  member_base: first symbol of the current gzip member in `out` (matches may not
// reach before it unless markers are allowed: chunk started inside that member).
template <bool STRICT>
BlockEnd inflate_codes(Bits& in, const BlockTables& t, SymBuf& out, size_t member_base, bool markers_ok, uint64_t& n_markers)
{
    for (;;)
    {
        in.refill();
        unsigned s = t.lit.decode(in);
        if (s < 256)
        {
            if (STRICT && !(s >= 32 && s < 127) && s != '\n' && s != '\r' && s != '\t')
                return BlockEnd::bad;
            out.need(1);
            out.v[out.n++] = (uint16_t)s;
            continue;
        }
        if (s == 256)
            return in.over ? BlockEnd::bad : BlockEnd::ok;
        if (s > 285 || !t.have_dist)
            return BlockEnd::bad;
        s -= 257;
        const unsigned len = kLenBase[s] + in.get(kLenExtra[s]);
        in.refill();
        const unsigned ds = t.dist.decode(in);
        if (ds > 29)
            return BlockEnd::bad;
        const unsigned dist = kDistBase[ds] + in.get(kDistExtra[ds]);
        if (in.over)
            return BlockEnd::bad;
        const size_t have = out.n - member_base;
        out.need(len);
        uint16_t* o = out.v.data() + out.n;
        if (dist <= have)
        {
            const uint16_t* src = o - dist;
            if (dist >= len)
                std::memcpy(o, src, len * 2);
            else
                for (unsigned i = 0; i < len; ++i)
                    o[i] = src[i];
        }
        else
        {
            // reaches back over the start of what this decoder has seen
            if (!markers_ok || member_base != 0 || dist > out.n + kWindow)
                return BlockEnd::bad;
            const long first = (long)out.n - (long)dist; // negative: position before the chunk
            for (unsigned i = 0; i < len; ++i)
            {
                const long at = first + (long)i;
                if (at < 0)
                {
                    o[i] = (uint16_t)(256 + (kWindow + at));
                    ++n_markers;
                }
                else
                    o[i] = out.v[(size_t)at];
            }
        }
        out.n += len;
    }
}

// gzip member header at byte p; returns the offset of the deflate data or 0 when it is no (complete) gzip header
size_t gzip_header(const uint8_t* d, size_t size, size_t p)
{
    if (p + 18 > size || d[p] != 0x1F || d[p + 1] != 0x8B || d[p + 2] != 8 || (d[p + 3] & 0xE0))
        return 0;
    const unsigned flg = d[p + 3];
    size_t         q   = p + 10;
    if (flg & 4)
    {
        if (q + 2 > size)
            return 0;
        q += 2 + (d[q] | (d[q + 1] << 8));
    }
    for (unsigned bit : { 8u, 16u })
        if (flg & bit)
        {
            while (q < size && d[q])
                ++q;
            ++q;
        }
    if (flg & 2)
        q += 2;
    return q < size ? q : 0;
}

struct MemberEnd
{
    size_t   sym_index; // symbols of the chunk that belong to members ending here or earlier
    uint32_t crc, isize;
};

struct Decoded
{
    SymBuf                 out;
    uint64_t               start_bit = kNone, end_bit = 0;
    bool                   at_stream_end = false;
    bool                   failed = false; // data error (or truncation) after the symbols in `out`
    std::vector<MemberEnd> member_ends;
    uint64_t               markers = 0;
    uint64_t               members_begun = 0;
};

// Decodes from start_bit (a block header inside a member; or, with at_header, a gzip member header at that byte) to the first
// block boundary at or after stop_bit, or to the end of the last member.
void decode_range(const uint8_t* data, size_t size, uint64_t start_bit, bool at_header, uint64_t stop_bit, bool markers_ok, Decoded& r)
{
    r.start_bit = start_bit;
    size_t member_base = 0;
    bool   fresh_member = false; // a member began inside this range: matches may not reach before it
    uint64_t pos = start_bit;
    if (at_header)
    {
        const size_t q = gzip_header(data, size, (size_t)(start_bit >> 3));
        if (!q)
        {
            r.failed  = true;
            r.end_bit = start_bit;
            return;
        }
        pos          = (uint64_t)q * 8;
        fresh_member = true;
        ++r.members_begun;
    }
    Bits        in(data, size, pos);
    BlockTables dyn;
    for (;;)
    {
        if (!(at_header && in.pos() == pos) && in.pos() >= stop_bit) // (a block boundary at or past the next chunk's start)
            break;
        if (r.out.n >= kMaxChunkSymbols) // extremely compressible data: hand over at this block boundary (the stitcher goes on from here)
            break;
        in.refill();
        const unsigned bfinal = in.get(1), btype = in.get(2);
        BlockEnd       be     = BlockEnd::ok;
        const bool     mk     = markers_ok && !fresh_member;
        if (btype == 0)
        {
            in.align_byte();
            in.refill();
            const unsigned len = in.get(16), nlen = in.get(16);
            if (in.over || (len ^ 0xFFFFu) != nlen)
                be = BlockEnd::bad;
            else
            {
                // the rest of the bit buffer is whole bytes; then straight from the input
                r.out.need(len);
                unsigned done = 0;
                while (done < len && in.cnt >= 8)
                {
                    r.out.v[r.out.n++] = (uint16_t)in.get(8);
                    ++done;
                }
                const size_t left = len - done;
                if (left) // (the bit buffer is empty now: the rest comes straight from the input)
                {
                    if ((size_t)(in.end - in.p) < left)
                        be = BlockEnd::bad;
                    else
                    {
                        for (size_t i = 0; i < left; ++i)
                            r.out.v[r.out.n++] = in.p[i];
                        in.p += left;
                        in.buf = 0;
                        in.cnt = 0;
                    }
                }
            }
        }
        else if (btype == 1)
            be = inflate_codes<false>(in, fixed_tables(), r.out, member_base, mk, r.markers);
        else if (btype == 2)
            be = read_dynamic_header(in, dyn) ? inflate_codes<false>(in, dyn, r.out, member_base, mk, r.markers) : BlockEnd::bad;
        else
            be = BlockEnd::bad;
        if (be == BlockEnd::bad || in.over)
        {
            r.failed  = true;
            r.end_bit = in.pos();
            return;
        }
        if (bfinal)
        {
            in.align_byte();
            const size_t at = (size_t)(in.pos() >> 3);
            if (at + 8 > size)
            {
                r.failed  = true;
                r.end_bit = in.pos();
                return;
            }
            const uint8_t* t = data + at;
            r.member_ends.push_back(MemberEnd{ r.out.n, (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24),
                                               (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24) });
            const size_t next = at + 8;
            const size_t q    = next < size ? gzip_header(data, size, next) : 0;
            if (!q) // the end of the file, or something that is no gzip member: zlib stops here as well
            {
                r.at_stream_end = true;
                r.end_bit       = (uint64_t)next * 8;
                return;
            }
            in           = Bits(data, size, (uint64_t)q * 8);
            member_base  = r.out.n;
            fresh_member = true;
            ++r.members_begun;
        }
        at_header = false;
    }
    r.end_bit = in.pos();
}

// First bit position in [lo, hi) at which a non-final dynamic block begins that decodes, start to end, into text, and is
// followed by something that looks like another block.  kNone if there is none.
uint64_t find_block_start(const uint8_t* data, size_t size, uint64_t lo, uint64_t hi)
{
    BlockTables t;
    SymBuf      scratch;
    scratch.v.resize(1u << 20);
    for (uint64_t bit = lo; bit < hi; ++bit)
    {
        // cheap rejections on the first 17 bits: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29
        const size_t byte = (size_t)(bit >> 3);
        if (byte + 4 > size)
            return kNone;
        uint32_t w;
        std::memcpy(&w, data + byte, 4);
        w >>= (bit & 7);
        if ((w & 7u) != 4u) // bit0 = 0 (not final), bits 1-2 = 10b (dynamic)
            continue;
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u)
            continue;
        Bits in(data, size, bit + 3);
        if (!read_dynamic_header(in, t))
            continue;
        scratch.n        = 0;
        uint64_t markers = 0;
        if (inflate_codes<true>(in, t, scratch, 0, true, markers) != BlockEnd::ok || scratch.n < 64)
            continue;
        // what follows: another block header of a legal kind (and, if dynamic, one that parses)
        in.refill();
        const unsigned nb = in.get(1), nt = in.get(2);
        (void)nb;
        if (in.over || nt == 3)
            continue;
        if (nt == 2)
        {
            BlockTables t2;
            if (!read_dynamic_header(in, t2))
                continue;
        }
        else if (nt == 0)
        {
            in.align_byte();
            in.refill();
            const unsigned len = in.get(16), nlen = in.get(16);
            if ((len ^ 0xFFFFu) != nlen)
                continue;
        }
        return bit;
    }
    return kNone;
}

} // namespace

// ---- the pipeline ---------------------------------------------------------------------------------------------------------
struct ParallelGzip::Impl
{
    // input
    int            fd = -1;
    const uint8_t* data = nullptr;
    size_t         size = 0, chunk_bytes = 0;
    std::atomic<size_t> n_chunks{ 0 }; // (grows by one per early hand-over, see stitch(): the slot is filled before the count moves)
    unsigned       n_threads = 1;

    struct Chunk
    {
        std::once_flag once_start;
        uint64_t       found_start = kNone; // search result (chunk 0: the file's first byte, a member header)
        // decode (speculative, by a worker)
        Decoded d;
        bool    decoded = false;
        // settled by the stitcher
        bool                 used = false;  // contributes output
        std::vector<uint8_t> window;        // the 32 KiB before it (markers refer to it)
        // resolved by a worker
        std::vector<char>     bytes;
        std::vector<uint32_t> seg_crc; // CRC-32 of the pieces between member ends
        bool                  resolved = false;
    };
    // One slot per chunk of the file plus spare slots for the chunks stitch() appends; the array of slots never moves (workers index it
    // without the lock), a spare's Chunk is created under the lock before n_chunks says it is there.
    std::vector<std::unique_ptr<Chunk>> chunks;

    std::mutex              m;
    std::condition_variable cv_work, cv_done, cv_read;
    size_t                  next_decode = 0;   // next chunk to hand to a worker for decoding
    size_t                  next_settle = 0;   // stitcher's position
    size_t                  next_publish = 0;
    std::deque<size_t>      resolve_q;
    bool                    stop = false;
    size_t                  window_chunks = 8;

    // published output
    struct Piece
    {
        uint64_t                                 off;
        std::shared_ptr<const std::vector<char>> bytes; // (readers copy outside the lock: a piece may be released meanwhile)
    };
    std::deque<Piece> pieces;
    uint64_t          published = 0, released = 0;
    uint64_t          retain_limit = 1ull << 30;
    bool              finished = false;
    std::string       error;      // set once: the stream is unusable from `published` on
    uint64_t          total = ~0ull;

    std::vector<std::thread> workers;
    std::thread              stitcher;
    Stats                    st;

    uint64_t start_of(size_t k)
    {
        Chunk& c = *chunks[k];
        std::call_once(c.once_start, [&] {
            if (k == 0)
                c.found_start = 0;
            else
                c.found_start = find_block_start(data, size, (uint64_t)k * chunk_bytes * 8, std::min<uint64_t>((uint64_t)(k + 1) * chunk_bytes, size) * 8);
        });
        return c.found_start;
    }
    // the first search result strictly after `bit` among the chunks behind k (the place a decode from k should stop at)
    uint64_t stop_after(size_t k, uint64_t bit)
    {
        for (size_t j = k + 1; j < n_chunks; ++j)
        {
            const uint64_t s = start_of(j);
            if (s != kNone && s > bit)
                return s;
        }
        return (uint64_t)size * 8;
    }

    void worker()
    {
        for (;;)
        {
            size_t k = 0;
            int    what = 0; // 1 decode, 2 resolve
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] {
                    return stop || !resolve_q.empty() || (next_decode < n_chunks && next_decode < next_publish + window_chunks && error.empty());
                });
                if (stop)
                    return;
                if (!resolve_q.empty())
                {
                    k    = resolve_q.front();
                    resolve_q.pop_front();
                    what = 2;
                }
                else
                {
                    k    = next_decode++;
                    what = 1;
                }
            }
            Chunk& c = *chunks[k];
            if (what == 1)
            {
                const uint64_t s = start_of(k);
                if (s != kNone)
                    decode_range(data, size, s, k == 0, stop_after(k, s), k != 0, c.d);
                std::lock_guard<std::mutex> lk(m);
                c.decoded = true;
                cv_done.notify_all();
            }
            else
            {
                resolve(c);
                std::lock_guard<std::mutex> lk(m);
                c.resolved = true;
                cv_done.notify_all();
            }
        }
    }

    void resolve(Chunk& c)
    {
        const size_t n = c.d.out.n;
        c.bytes.resize(n);
        const uint16_t* s = c.d.out.v.data();
        const uint8_t*  w = c.window.data();
        char*           o = c.bytes.data();
        for (size_t i = 0; i < n; ++i)
        {
            const uint16_t v = s[i];
            o[i]             = (char)(v < 256 ? v : w[v - 256]);
        }
        c.seg_crc.clear();
        size_t at = 0;
        for (auto const& me : c.d.member_ends)
        {
            c.seg_crc.push_back((uint32_t)crc32(0L, reinterpret_cast<const Bytef*>(o + at), (uInt)(me.sym_index - at)));
            at = me.sym_index;
        }
        c.seg_crc.push_back((uint32_t)crc32(0L, reinterpret_cast<const Bytef*>(o + at), (uInt)(n - at)));
        std::vector<uint16_t>().swap(c.d.out.v); // the symbols are not needed any more
    }

    void fail(const std::string& msg)
    {
        std::lock_guard<std::mutex> lk(m);
        if (error.empty())
            error = msg;
        finished = true;
        cv_read.notify_all();
        cv_work.notify_all();
    }

    // In file order: is the chunk's speculative decode the continuation of what came before?  If not, decode again from the
    // true position.  Settle its window, hand it to a worker for resolving, publish resolved chunks, check the members.
    void stitch()
    {
        std::vector<uint8_t> win(kWindow, 0);
        uint64_t             pos = 0;       // bit position the stream has been decoded to
        bool                 ended = false;
        uint32_t             run_crc = 0;   // CRC of the current member so far
        uint64_t             run_len = 0;
        size_t               k = 0, pub = 0;
        while (pub < n_chunks)
        {
            // settle as many chunks as are decoded
            bool progressed = false;
            if (k == n_chunks && !ended && pos < (uint64_t)size * 8)
            {
                // every chunk is settled but the stream is not at its end: the last decode handed over early (the cap on the
                // symbols one decode keeps).  One more chunk, behind the file's last, that continues from `pos`.
                const size_t at = n_chunks.load();
                if (at >= chunks.size())
                {
                    fail("more early hand-overs than a deflate stream of this size can need");
                    break;
                }
                std::lock_guard<std::mutex> lk(m);
                chunks[at].reset(new Chunk());
                chunks[at]->decoded = true; // (no speculative decode: the "does not begin at pos" path below decodes it)
                n_chunks.store(at + 1);
                next_decode = at + 1;
            }
            if (k < n_chunks)
            {
                Chunk* c = chunks[k].get();
                bool   ready;
                {
                    std::unique_lock<std::mutex> lk(m);
                    ready = c->decoded;
                    if (stop)
                        return;
                }
                if (ready)
                {
                    progressed = true;
                    c->used    = false;
                    if (!ended)
                    {
                        const uint64_t s = c->d.start_bit;
                        if (k == 0 || (s != kNone && s == pos && !c->d.failed))
                            c->used = true;
                        else if ((uint64_t)(k + 1) * chunk_bytes * 8 > pos || k + 1 == n_chunks)
                        {
                            // the stream stands at `pos` inside (or before) this chunk's range and the chunk does not begin
                            // there: a false start, a start the predecessor ran over, or a failed decode -- again, from `pos`
                            if (pos < (uint64_t)size * 8)
                            {
                                c->d = Decoded();
                                decode_range(data, size, pos, false, stop_after(k, pos), true, c->d);
                                c->used = true;
                                ++st.redone;
                            }
                        }
                        if (c->used)
                        {
                            if (c->d.failed && c->d.out.n == 0 && c->d.member_ends.empty())
                            {
                                fail("damaged or truncated gzip stream");
                                return;
                            }
                            pos   = c->d.end_bit;
                            ended = c->d.at_stream_end || c->d.failed;
                            st.markers += c->d.markers;
                            st.members += c->d.members_begun;
                            ++st.chunks;
                            // the window it was decoded against, and the window it leaves behind
                            c->window = win;
                            const size_t    n = c->d.out.n;
                            const uint16_t* sy = c->d.out.v.data();
                            if (n >= kWindow)
                            {
                                std::vector<uint8_t> nw(kWindow);
                                for (size_t i = 0; i < kWindow; ++i)
                                {
                                    const uint16_t v = sy[n - kWindow + i];
                                    nw[i]            = v < 256 ? (uint8_t)v : c->window[v - 256];
                                }
                                win.swap(nw);
                            }
                            else if (n)
                            {
                                std::memmove(win.data(), win.data() + n, kWindow - n);
                                for (size_t i = 0; i < n; ++i)
                                {
                                    const uint16_t v      = sy[i];
                                    win[kWindow - n + i]  = v < 256 ? (uint8_t)v : c->window[v - 256];
                                }
                            }
                            std::lock_guard<std::mutex> lk(m);
                            resolve_q.push_back(k);
                            cv_work.notify_all();
                        }
                    }
                    if (!c->used)
                    {
                        std::lock_guard<std::mutex> lk(m);
                        c->resolved = true; // nothing to resolve
                    }
                    ++k;
                }
            }
            // publish in order
            for (;;)
            {
                Chunk* c = pub < n_chunks ? chunks[pub].get() : nullptr;
                if (!c || pub >= k)
                    break;
                {
                    std::unique_lock<std::mutex> lk(m);
                    if (!c->resolved)
                        break;
                }
                progressed = true;
                if (c->used)
                {
                    // members that end in this chunk: CRC-32 and length must be what their trailers say
                    size_t at = 0;
                    for (size_t e = 0; e < c->d.member_ends.size(); ++e)
                    {
                        const MemberEnd& me = c->d.member_ends[e];
                        run_crc = (uint32_t)crc32_combine(run_crc, c->seg_crc[e], (z_off_t)(me.sym_index - at));
                        run_len += me.sym_index - at;
                        if (run_crc != me.crc || (uint32_t)run_len != me.isize)
                        {
                            fail("gzip member with a wrong CRC-32 or length");
                            return;
                        }
                        run_crc = 0;
                        run_len = 0;
                        at      = me.sym_index;
                    }
                    const size_t rest = c->bytes.size() - at;
                    run_crc = (uint32_t)crc32_combine(run_crc, c->seg_crc.back(), (z_off_t)rest);
                    run_len += rest;
                    const bool broken = c->d.failed;
                    std::unique_lock<std::mutex> lk(m);
                    cv_read.wait(lk, [&] { return stop || published - released < retain_limit; });
                    if (stop)
                        return;
                    if (!c->bytes.empty())
                    {
                        pieces.push_back(Piece{ published, std::make_shared<const std::vector<char>>(std::move(c->bytes)) });
                        published += pieces.back().bytes->size();
                    }
                    if (broken && error.empty())
                    {
                        error    = "damaged or truncated gzip stream";
                        finished = true;
                    }
                    cv_read.notify_all();
                }
                {
                    std::lock_guard<std::mutex> lk(m);
                    std::vector<uint8_t>().swap(c->window);
                    ++pub;
                    next_publish = pub;
                    cv_work.notify_all();
                }
            }
            if (!progressed)
            {
                std::unique_lock<std::mutex> lk(m);
                cv_done.wait_for(lk, std::chrono::milliseconds(50));
                if (stop)
                    return;
            }
        }
        std::lock_guard<std::mutex> lk(m);
        if (error.empty() && !ended)
            error = "truncated gzip stream";
        finished = true;
        total    = published;
        cv_read.notify_all();
    }
};

ParallelGzip::ParallelGzip(Impl* i) : impl_(i) {}

std::unique_ptr<ParallelGzip> ParallelGzip::open(const std::string& path, unsigned threads, size_t min_bytes, size_t chunk_bytes)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0)
        return nullptr;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < std::max<size_t>(min_bytes, 64))
    {
        ::close(fd);
        return nullptr;
    }
    void* p = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED)
    {
        ::close(fd);
        return nullptr;
    }
    const uint8_t* d = static_cast<const uint8_t*>(p);
    if (gzip_header(d, (size_t)sb.st_size, 0) == 0)
    {
        munmap(p, (size_t)sb.st_size);
        ::close(fd);
        return nullptr;
    }
    madvise(p, (size_t)sb.st_size, MADV_SEQUENTIAL);
    Impl* im        = new Impl;
    im->fd          = fd;
    im->data        = d;
    im->size        = (size_t)sb.st_size;
    im->chunk_bytes = chunk_bytes ? std::max<size_t>(chunk_bytes, 1024) : (2u << 20);
    im->n_chunks    = (im->size + im->chunk_bytes - 1) / im->chunk_bytes;
    im->n_threads   = std::max(1u, threads);
    im->window_chunks = 2 * im->n_threads + 2;
    // (deflate expands at most 1032-fold; a decode hands over after 96 M symbols at the latest: the spare slots cover every hand-over)
    im->chunks.resize(im->n_chunks + (size_t)(((uint64_t)im->size * 1032ull) / (96ull << 20)) + 16);
    for (size_t i = 0; i < im->n_chunks; ++i)
        im->chunks[i].reset(new Impl::Chunk());
    std::unique_ptr<ParallelGzip> pg(new ParallelGzip(im));
    for (unsigned t = 0; t < im->n_threads; ++t)
        im->workers.emplace_back([im] { im->worker(); });
    im->stitcher = std::thread([im] { im->stitch(); });
    return pg;
}

ParallelGzip::~ParallelGzip()
{
    Impl& s = *impl_;
    {
        std::lock_guard<std::mutex> lk(s.m);
        s.stop = true;
        s.cv_work.notify_all();
        s.cv_done.notify_all();
        s.cv_read.notify_all();
    }
    for (auto& t : s.workers)
        t.join();
    if (s.stitcher.joinable())
        s.stitcher.join();
    munmap(const_cast<uint8_t*>(s.data), s.size);
    ::close(s.fd);
}

size_t ParallelGzip::pread(char* dst, size_t n, uint64_t off)
{
    Impl&                        s = *impl_;
    std::unique_lock<std::mutex> lk(s.m);
    size_t                       got = 0;
    while (got < n)
    {
        const uint64_t at = off + got;
        s.cv_read.wait(lk, [&] { return s.stop || s.published > at || s.finished; });
        if (s.published <= at)
        {
            if (!s.error.empty())
                throw std::runtime_error(s.error);
            break; // the end of the stream
        }
        if (at < s.released)
            throw std::runtime_error("pgzip: read below the released offset");
        // the piece that holds `at` (pieces are few: a linear walk from the front); the copy runs without the lock
        std::shared_ptr<const std::vector<char>> hold;
        uint64_t                                 piece_off = 0;
        for (auto const& pc : s.pieces)
            if (at >= pc.off && at < pc.off + pc.bytes->size())
            {
                hold      = pc.bytes;
                piece_off = pc.off;
                break;
            }
        if (!hold)
            throw std::runtime_error("pgzip: offset not retained");
        const size_t k = (size_t)std::min<uint64_t>(n - got, piece_off + hold->size() - at);
        lk.unlock();
        std::memcpy(dst + got, hold->data() + (at - piece_off), k);
        lk.lock();
        got += k;
    }
    return got;
}

void ParallelGzip::release_below(uint64_t off)
{
    Impl&                       s = *impl_;
    std::lock_guard<std::mutex> lk(s.m);
    if (off <= s.released)
        return;
    s.released = std::min(off, s.published);
    while (!s.pieces.empty() && s.pieces.front().off + s.pieces.front().bytes->size() <= s.released)
        s.pieces.pop_front();
    s.cv_read.notify_all();
}

void ParallelGzip::set_retain_limit(uint64_t bytes)
{
    std::lock_guard<std::mutex> lk(impl_->m);
    impl_->retain_limit = std::max<uint64_t>(bytes, 64u << 20);
    impl_->cv_read.notify_all();
}

uint64_t ParallelGzip::known_size() const
{
    std::lock_guard<std::mutex> lk(impl_->m);
    return impl_->total;
}

ParallelGzip::Stats ParallelGzip::stats() const
{
    std::lock_guard<std::mutex> lk(impl_->m);
    return impl_->st;
}

} // namespace gnhost
