// seq_io.cpp -- see seq_io.hpp
#include "seq_io.hpp"
#include "pgzip.hpp"
#include "cpu_tally.hpp"
#include "tunables.hpp"

#include <tmmintrin.h>
#include <zlib.h>
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cstring>
#include <map>
#include <string_view>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gnhost
{

namespace
{
bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = std::strlen(suf);
    return s.size() >= n && std::equal(s.end() - n, s.end(), suf, [](char a, char b) { return std::tolower(a) == b; });
}

// dna15 legal letters
struct LegalTable
{
    bool ok[256];
    LegalTable()
    {
        std::fill(ok, ok + 256, false);
        for (const char* p = "ACGTURYSWKMBDHVN"; *p; ++p)
        {
            ok[(unsigned char)*p]               = true;
            ok[(unsigned char)std::tolower(*p)] = true;
        }
    }
};
const LegalTable kLegal;

bool all_legal_scalar(const char* p, size_t n)
{
    unsigned bad = 0;
    for (size_t i = 0; i < n; ++i)
        bad |= !kLegal.ok[(unsigned char)p[i]];
    return !bad;
}

// 16 letters per step: a letter is legal iff its high nibble selects the A-O / P-Z half (either case) and its low
// nibble is a member of that half's set -- two 16-entry byte shuffles and an AND.
__attribute__((target("ssse3"))) bool all_legal_ssse3(const char* p, size_t n)
{
    // low-nibble classes: bit 0 = legal in A-O (A B C D G H K M N), bit 1 = legal in P-Z (R S T U V W Y)
    const __m128i lo_tbl = _mm_setr_epi8(0, 1, 3, 3, 3, 2, 2, 3, 1, 2, 0, 1, 0, 1, 1, 0);
    // high-nibble classes: 0x4_/0x6_ -> bit 0, 0x5_/0x7_ -> bit 1
    const __m128i hi_tbl = _mm_setr_epi8(0, 0, 0, 0, 1, 2, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m128i nib    = _mm_set1_epi8(0x0F);
    __m128i       ok_all = _mm_set1_epi8((char)0xFF);
    size_t        i      = 0;
    for (; i + 16 <= n; i += 16)
    {
        const __m128i v  = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i));
        const __m128i lo = _mm_shuffle_epi8(lo_tbl, _mm_and_si128(v, nib));
        const __m128i hi = _mm_shuffle_epi8(hi_tbl, _mm_and_si128(_mm_srli_epi16(v, 4), nib));
        const __m128i ok = _mm_cmpgt_epi8(_mm_and_si128(lo, hi), _mm_setzero_si128());
        ok_all           = _mm_and_si128(ok_all, ok);
    }
    if (_mm_movemask_epi8(ok_all) != 0xFFFF)
        return false;
    return all_legal_scalar(p + i, n - i);
}

bool all_legal(const char* p, size_t n)
{
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3");
    return have_ssse3 ? all_legal_ssse3(p, n) : all_legal_scalar(p, n);
}
} // namespace

// ---- BGZF (blocked gzip: bgzip, Illumina bcl2fastq/BCL Convert) ---------------------------------------------------
// A BGZF file is a series of gzip members of at most 64 KiB, each with a 'BC' extra subfield that holds its compressed
// size and a trailer that holds its inflated size -- so the members can be found without decoding and inflated
// independently.  A producer thread reads a batch of members, helper threads inflate them straight into one output
// buffer at the offsets the trailers give, and the parser consumes the buffers in order.  Plain gzip (one long
// deflate stream) cannot be split this way and keeps going through gzread().
class BgzfSource
{
public:
    // nullptr when the file is not BGZF
    static std::unique_ptr<BgzfSource> open(const std::string& path)
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f)
            return nullptr;
        unsigned char h[18];
        const bool    ok = std::fread(h, 1, 18, f) == 18 && member_size(h) != 0;
        if (!ok)
        {
            std::fclose(f);
            return nullptr;
        }
        std::fseek(f, 0, SEEK_SET);
        return std::unique_ptr<BgzfSource>(new BgzfSource(f));
    }
    ~BgzfSource()
    {
        stop_ = true;
        {
            std::unique_lock<std::mutex> lk(m_);
            q_.clear();
            cv_space_.notify_all();
        }
        producer_.join();
        std::fclose(f_);
    }
    // up to n bytes; 0 at end of file; throws ParseError on a damaged file
    size_t read(char* dst, size_t n)
    {
        size_t got = 0;
        while (got < n)
        {
            if (pos_ == cur_.size())
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_data_.wait(lk, [&] { return !q_.empty() || done_; });
                if (q_.empty())
                {
                    if (!error_.empty())
                        throw ParseError(error_);
                    break;
                }
                cur_ = std::move(q_.front());
                q_.pop_front();
                pos_ = 0;
                cv_space_.notify_one();
                continue;
            }
            const size_t k = std::min(n - got, cur_.size() - pos_);
            std::memcpy(dst + got, cur_.data() + pos_, k);
            got += k;
            pos_ += k;
        }
        return got;
    }

private:
    explicit BgzfSource(FILE* f) : f_(f), producer_([this] { produce(); }) {}

    // total size of the gzip member whose first 18 bytes are h, 0 if it is not a BGZF member
    static size_t member_size(const unsigned char* h)
    {
        if (h[0] != 0x1F || h[1] != 0x8B || h[2] != 8 || !(h[3] & 4))
            return 0;
        const unsigned xlen = h[10] | (h[11] << 8);
        if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0)
            return 0; // (bgzip always writes BC as the first and only subfield)
        return (size_t)(h[16] | (h[17] << 8)) + 1;
    }

    void fail(const std::string& msg)
    {
        std::lock_guard<std::mutex> lk(m_);
        error_ = msg;
    }

    void produce()
    {
        const unsigned T = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        struct Member
        {
            size_t in_off, in_len, out_off, out_len;
        };
        std::vector<unsigned char> in;
        bool                       eof = false;
        while (!stop_ && !eof)
        {
            // a batch: members until ~16 MiB of output
            in.clear();
            std::vector<Member> mem;
            size_t              out_total = 0;
            while (out_total < (16u << 20))
            {
                unsigned char h[18];
                const size_t  r = std::fread(h, 1, 18, f_);
                if (r == 0)
                {
                    eof = true;
                    break;
                }
                const size_t sz = r == 18 ? member_size(h) : 0;
                if (sz < 18 + 8)
                {
                    fail(" damaged BGZF member header");
                    eof = true;
                    break;
                }
                const size_t at = in.size();
                in.resize(at + sz);
                std::memcpy(in.data() + at, h, 18);
                if (std::fread(in.data() + at + 18, 1, sz - 18, f_) != sz - 18)
                {
                    fail(" truncated BGZF member");
                    in.resize(at);
                    eof = true;
                    break;
                }
                const unsigned char* t     = in.data() + at + sz - 4;
                const size_t         isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                mem.push_back(Member{ at, sz, out_total, isize });
                out_total += isize;
            }
            if (mem.empty())
                break;
            std::vector<char> out(out_total);
            std::atomic<size_t> next{ 0 };
            std::atomic<bool>   bad{ false };
            auto work = [&] {
                z_stream z;
                for (size_t i = next++; i < mem.size() && !bad; i = next++)
                {
                    const Member& m = mem[i];
                    if (m.out_len == 0) // (the empty end-of-file member)
                        continue;
                    std::memset(&z, 0, sizeof(z));
                    const unsigned xlen = in[m.in_off + 10] | (in[m.in_off + 11] << 8);
                    const size_t   hdr  = 12 + xlen;
                    if (hdr + 8 > m.in_len || inflateInit2(&z, -15) != Z_OK)
                    {
                        bad = true;
                        break;
                    }
                    z.next_in   = in.data() + m.in_off + hdr;
                    z.avail_in  = (uInt)(m.in_len - hdr - 8);
                    z.next_out  = reinterpret_cast<Bytef*>(out.data() + m.out_off);
                    z.avail_out = (uInt)m.out_len;
                    const int rc = inflate(&z, Z_FINISH);
                    if (rc != Z_STREAM_END || z.total_out != m.out_len)
                        bad = true;
                    inflateEnd(&z);
                }
            };
            std::vector<std::thread> helpers;
            for (unsigned t = 1; t < T && t < mem.size(); ++t)
                helpers.emplace_back(work);
            work();
            for (auto& th : helpers)
                th.join();
            if (bad)
            {
                fail(" damaged BGZF member (inflate failed)");
                break;
            }
            std::unique_lock<std::mutex> lk(m_);
            cv_space_.wait(lk, [&] { return q_.size() < 4 || stop_; });
            if (stop_)
                break;
            q_.push_back(std::move(out));
            cv_data_.notify_one();
        }
        std::lock_guard<std::mutex> lk(m_);
        done_ = true;
        cv_data_.notify_all();
    }

    FILE*                         f_;
    std::mutex                    m_;
    std::condition_variable       cv_data_, cv_space_;
    std::deque<std::vector<char>> q_;
    bool                          done_ = false;
    std::string                   error_;
    std::atomic<bool>             stop_{ false };
    std::vector<char>             cur_;
    size_t                        pos_ = 0;
    std::thread                   producer_; // last member
};

// bzip2 input (the reference reads it through seqan3 when it is built with bzip2).  The image has libbz2's shared object but
// not its header, so the library is opened at run time and the few entry points of its stable low-level interface
// (bzlib.h of bzip2 1.0: bz_stream, BZ2_bzDecompressInit / BZ2_bzDecompress / BZ2_bzDecompressEnd) are declared here.
// Concatenated streams (pbzip2 output) are read one after the other.
struct Bz2Source
{
    struct Stream // bz_stream
    {
        char*        next_in;
        unsigned int avail_in, total_in_lo32, total_in_hi32;
        char*        next_out;
        unsigned int avail_out, total_out_lo32, total_out_hi32;
        void*        state;
        void* (*bzalloc)(void*, int, int);
        void (*bzfree)(void*, void*);
        void* opaque;
    };
    using InitFn = int (*)(Stream*, int, int);
    using RunFn  = int (*)(Stream*);
    void*             lib = nullptr;
    InitFn            init = nullptr;
    RunFn             run = nullptr, end = nullptr;
    FILE*             fp = nullptr;
    Stream            st{};
    bool              open_stream = false, file_eof = false;
    bool              broken = false; // an error is due with the next call (what was decoded before it has been delivered)
    unsigned          streams_done = 0;
    std::vector<char> in;

    static std::unique_ptr<Bz2Source> open(const std::string& path, std::string& why)
    {
        std::unique_ptr<Bz2Source> b(new Bz2Source);
        for (const char* name : { "libbz2.so.1.0", "libbz2.so.1", "libbz2.so" })
            if ((b->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!b->lib)
        {
            why = " bzip2 input needs libbz2 (libbz2.so.1.0 could not be loaded)";
            return nullptr;
        }
        b->init = reinterpret_cast<InitFn>(dlsym(b->lib, "BZ2_bzDecompressInit"));
        b->run  = reinterpret_cast<RunFn>(dlsym(b->lib, "BZ2_bzDecompress"));
        b->end  = reinterpret_cast<RunFn>(dlsym(b->lib, "BZ2_bzDecompressEnd"));
        if (!b->init || !b->run || !b->end)
        {
            why = " libbz2 lacks the decompression entry points";
            return nullptr;
        }
        b->fp = std::fopen(path.c_str(), "rb");
        if (!b->fp)
        {
            why = " cannot open file";
            return nullptr;
        }
        b->in.resize(1 << 20);
        return b;
    }
    ~Bz2Source()
    {
        if (open_stream)
            end(&st);
        if (fp)
            std::fclose(fp);
        if (lib)
            dlclose(lib);
    }
    // up to `want` decompressed bytes; 0 at the end of the file; -1 on a damaged stream
    long read(char* out, size_t want)
    {
        size_t done = 0;
        if (broken)
            return -1;
        while (done < want)
        {
            if (st.avail_in == 0 && !file_eof)
            {
                const size_t n = std::fread(in.data(), 1, in.size(), fp);
                if (n == 0)
                    file_eof = true;
                st.next_in  = in.data();
                st.avail_in = (unsigned int)n;
            }
            if (!open_stream)
            {
                if (st.avail_in == 0) // nothing after the last stream
                    break;
                char*              keep_in = st.next_in;
                const unsigned int keep_n  = st.avail_in;
                st = Stream{};
                if (init(&st, 0, 0) != 0)
                    return -1;
                st.next_in  = keep_in;
                st.avail_in = keep_n;
                open_stream = true;
            }
            st.next_out  = out + done;
            st.avail_out = (unsigned int)std::min<size_t>(want - done, 1u << 30);
            const unsigned int before = st.avail_out;
            const int          rc     = run(&st);
            done += before - st.avail_out;
            if (rc == -5 && streams_done > 0) // BZ_DATA_ERROR_MAGIC
            {
                // what follows the last complete stream does not begin like a bzip2 stream (padding, garbage): the bzip2 tool
                // warns and stops there ("trailing garbage after EOF ignored"); so does this reader
                end(&st);
                open_stream = false;
                st          = Stream{};
                file_eof    = true;
                break;
            }
            if (rc == 4) // BZ_STREAM_END: another stream may follow
            {
                ++streams_done;
                char*              keep_in = st.next_in;
                const unsigned int keep_n  = st.avail_in;
                end(&st);
                open_stream = false;
                st          = Stream{};
                st.next_in  = keep_in;
                st.avail_in = keep_n;
                continue;
            }
            if (rc != 0) // BZ_OK
            {
                if (done) // what was decoded before the damage is delivered first; the error comes with the next call
                {
                    broken = true;
                    return (long)done;
                }
                return -1;
            }
            if (st.avail_in == 0 && file_eof) // the stream wants more than the file holds
                return before == st.avail_out ? -1 : (long)done;
        }
        return (long)done;
    }
};

struct SeqReader::Impl
{
    gzFile                      gz = nullptr;
    std::unique_ptr<BgzfSource> bgzf; // set instead of gz for blocked gzip
    std::unique_ptr<Bz2Source>  bz2;  // set instead of gz for bzip2
    std::string                 path;
    bool              fastq = false;
    std::vector<char> buf;
    size_t            pos = 0, len = 0;
    bool              eof = false;
    bool              have_pending = false; // FASTA: header of the next record already read
    std::string       pending;

    // moves the unread tail to the front and reads more; false when nothing could be added
    bool refill()
    {
        if (eof)
            return false;
        if (pos > 0)
        {
            std::memmove(buf.data(), buf.data() + pos, len - pos);
            len -= pos;
            pos = 0;
        }
        if (len == buf.size())
            buf.resize(buf.size() * 2); // a single line longer than the buffer
        const size_t want = std::min<size_t>(buf.size() - len, 1u << 30);
        const int    n    = bz2 ? (int)bz2->read(buf.data() + len, want)
                                : bgzf ? (int)bgzf->read(buf.data() + len, want) : gzread(gz, buf.data() + len, (unsigned)want);
        if (n < 0 && bz2)
            throw ParseError(" damaged bzip2 stream");
        if (n < 0)
        {
            int         err = 0;
            const char* msg = gzerror(gz, &err);
            throw ParseError(" " + std::string(msg ? msg : "decompression error"));
        }
        if (n == 0)
        {
            eof = true;
            return false;
        }
        len += (size_t)n;
        return true;
    }
    // next line as a view into the buffer (no '\n', trailing '\r' stripped), valid until the next call;
    // false at EOF with nothing read
    bool line(std::string_view& out)
    {
        size_t scanned = 0; // bytes after pos already known to hold no newline
        while (true)
        {
            const char* b  = buf.data() + pos;
            const char* nl = static_cast<const char*>(std::memchr(b + scanned, '\n', len - pos - scanned));
            if (nl)
            {
                size_t n = (size_t)(nl - b);
                pos += n + 1;
                if (n && b[n - 1] == '\r')
                    --n;
                out = std::string_view(b, n);
                return true;
            }
            scanned = len - pos;
            if (!refill())
            {
                if (len == pos)
                    return false;
                size_t n = len - pos; // last line without a newline
                b        = buf.data() + pos;
                pos      = len;
                if (n && b[n - 1] == '\r')
                    --n;
                out = std::string_view(b, n);
                return true;
            }
        }
    }
};

[[noreturn]] void bad_letter(std::string_view l)
{
    for (char c : l)
        if (!kLegal.ok[(unsigned char)c])
            throw ParseError(std::string(" Encountered an unexpected letter: char_is_valid_for<dna15> evaluated to false on '") + c
                             + "'");
    throw ParseError(" Encountered an unexpected letter");
}

SeqReader::SeqReader(const std::string& path, uint64_t start_offset) : impl_(new Impl)
{
    impl_->path = path;
    std::string base = path;
    for (const char* z : { ".gz", ".bgzf", ".bgz" })
        if (ends_with(base, z))
        {
            base = base.substr(0, base.size() - std::strlen(z));
            break;
        }
    const bool is_bz2 = ends_with(base, ".bz2");
    if (is_bz2)
        base = base.substr(0, base.size() - 4);
    bool known = false;
    for (const char* e : { ".fq", ".fastq" })
        if (ends_with(base, e))
        {
            impl_->fastq = true;
            known        = true;
        }
    for (const char* e : { ".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn", ".fas" })
        if (ends_with(base, e))
            known = true;
    if (!known)
        throw ParseError(" unknown sequence file extension (expected FASTA or FASTQ, optionally gzipped)");
    impl_->buf.resize(4 << 20);
    if (is_bz2)
    {
        if (start_offset)
            throw ParseError(" cannot seek in a bzip2 file");
        std::string why;
        impl_->bz2 = Bz2Source::open(path, why);
        if (!impl_->bz2)
            throw ParseError(why);
        return;
    }
    if (!tun().is_set(Knob::no_bgzf) && start_offset == 0)
        impl_->bgzf = BgzfSource::open(path); // blocked gzip: members inflated in parallel
    if (impl_->bgzf)
        return;
    impl_->gz = gzopen(path.c_str(), "rb"); // transparently reads uncompressed files too
    if (!impl_->gz)
        throw ParseError(" cannot open file");
    gzbuffer(impl_->gz, 1 << 20);
    if (start_offset && gzseek(impl_->gz, (z_off_t)start_offset, SEEK_SET) < 0)
        throw ParseError(" cannot seek in file");
}

SeqReader::~SeqReader()
{
    if (impl_ && impl_->gz)
        gzclose(impl_->gz);
}

bool SeqReader::next(std::string& ids, ByteBuf& bases)
{
    Impl&            s = *impl_;
    const size_t     ids0 = ids.size(), bases0 = bases.size();
    std::string_view l;
    try
    {
        if (s.fastq)
        {
            // @id / sequence line(s) / +[id] / quality line(s)
            do
            {
                if (!s.line(l))
                    return false;
            } while (l.empty());
            if (l[0] != '@')
                throw ParseError(" FASTQ record does not start with '@'");
            ids.append(l.data() + 1, l.size() - 1);
            while (true)
            {
                if (!s.line(l))
                    throw ParseError(" unexpected end of FASTQ record");
                if (!l.empty() && l[0] == '+')
                    break;
                if (!all_legal(l.data(), l.size()))
                    bad_letter(l);
                const size_t at = bases.size();
                bases.resize(at + l.size());
                std::memcpy(bases.data() + at, l.data(), l.size());
            }
            const size_t want = bases.size() - bases0;
            size_t       q    = 0;
            while (q < want)
            {
                if (!s.line(l))
                    throw ParseError(" unexpected end of FASTQ qualities");
                q += l.size();
            }
            if (q != want)
                throw ParseError(" sequence and quality lengths differ");
            return true;
        }
        // FASTA
        if (s.have_pending)
            l = s.pending;
        else
            do
            {
                if (!s.line(l))
                    return false;
            } while (l.empty());
        s.have_pending = false;
        if (l[0] != '>' && l[0] != ';')
            throw ParseError(" FASTA record does not start with '>'");
        ids.append(l.data() + 1, l.size() - 1);
        while (s.line(l))
        {
            if (!l.empty() && (l[0] == '>' || l[0] == ';'))
            {
                s.pending.assign(l);
                s.have_pending = true;
                break;
            }
            // a line of nothing but legal letters (every line of an ordinary genome file) is appended in one piece;
            // whitespace, digits and illegal letters take the character-wise path (a legal letter is neither)
            if (all_legal(l.data(), l.size()))
            {
                bases.insert(bases.end(), reinterpret_cast<const uint8_t*>(l.data()), reinterpret_cast<const uint8_t*>(l.data()) + l.size());
                continue;
            }
            for (char c : l)
            {
                if (std::isspace((unsigned char)c) || std::isdigit((unsigned char)c))
                    continue;
                if (!kLegal.ok[(unsigned char)c])
                    bad_letter(std::string_view(&c, 1));
                bases.push_back((uint8_t)c);
            }
        }
        return true;
    }
    catch (ParseError const&)
    {
        ids.resize(ids0); // the record that failed leaves nothing behind
        bases.resize(bases0);
        throw;
    }
}

// ---- ParallelFastq ----------------------------------------------------------------------------------------------------
namespace
{
// Lines of a byte range of a file, read with pread into a private buffer (no shared page tables, no page faults: the
// mapping-based first version did not scale past a few threads on mmap_lock).
class RangeLines
{
public:
    // an uncompressed file (pread), or the decompressed stream of a gzip file (ParallelGzip::pread; its size is known only at its
    // end; a damaged stream makes refill throw)
    RangeLines(int fd, uint64_t file_size, ParallelGzip* gz = nullptr, size_t buffer = 4u << 20) : fd_(fd), gz_(gz), size_(file_size), buf_(buffer)
    {
        data_ = buf_.data();
    }
    // the whole file mapped into memory: lines are views of the mapping, nothing is copied
    RangeLines(const char* map, uint64_t file_size) : fd_(-1), gz_(nullptr), size_(file_size), data_(map), len_((size_t)file_size), mapped_(true) {}
    void seek(uint64_t off)
    {
        if (off >= base_ && off <= base_ + len_)
            pos_ = (size_t)(off - base_);
        else
        {
            base_ = off;
            len_ = pos_ = 0;
        }
    }
    uint64_t tell() const { return base_ + pos_; }
    // next line (no '\n', trailing '\r' stripped), valid until the next call; false at the end of the file
    bool line(std::string_view& out)
    {
        size_t scanned = 0;
        for (;;)
        {
            const char* b  = data_ + pos_;
            const char* nl = static_cast<const char*>(std::memchr(b + scanned, '\n', len_ - pos_ - scanned));
            if (nl)
            {
                size_t n = (size_t)(nl - b);
                pos_ += n + 1;
                if (n && b[n - 1] == '\r')
                    --n;
                out = std::string_view(b, n);
                return true;
            }
            scanned = len_ - pos_;
            if (!refill())
            {
                if (len_ == pos_)
                    return false;
                size_t n = len_ - pos_; // last line without a newline
                b        = data_ + pos_;
                pos_     = len_;
                if (n && b[n - 1] == '\r')
                    --n;
                out = std::string_view(b, n);
                return true;
            }
        }
    }
    // n bytes followed by '\n' right here?  (the quality line of a four-line record: no search needed)
    bool skip_exact_line(size_t n, std::string_view& out)
    {
        while (len_ - pos_ < n + 1)
            if (!refill())
                return false;
        const char* b = data_ + pos_;
        if (b[n] != '\n' || (n && std::memchr(b, '\n', n)))
            return false;
        out = std::string_view(b, n);
        pos_ += n + 1;
        return true;
    }

private:
    bool refill()
    {
        if (mapped_ || base_ + len_ >= size_)
            return false;
        if (pos_ > 0)
        {
            std::memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
            base_ += pos_;
            len_ -= pos_;
            pos_ = 0;
        }
        if (len_ == buf_.size())
            buf_.resize(buf_.size() * 2); // a single line longer than the buffer
        data_ = buf_.data();
        const size_t  want = (size_t)std::min<uint64_t>(buf_.size() - len_, size_ - (base_ + len_));
        const ssize_t got  = gz_ ? (ssize_t)gz_->pread(buf_.data() + len_, want, base_ + len_)
                                 : ::pread(fd_, buf_.data() + len_, want, (off_t)(base_ + len_));
        if (got <= 0)
            return false;
        len_ += (size_t)got;
        return true;
    }
    int               fd_;
    ParallelGzip*     gz_;
    uint64_t          size_, base_ = 0;
    std::vector<char> buf_;
    const char*       data_ = nullptr;
    size_t            pos_ = 0, len_ = 0;
    bool              mapped_ = false;
};
} // namespace

struct ParallelFastq::Impl
{
    size_t      size = 0;   // bytes of the (decompressed) input; for a gzip stream unknown until its end has been read
    int         fd   = -1;
    const char* map  = nullptr; // the file mapped read-only (uncompressed input): the parsers read the page cache in place
    std::unique_ptr<ParallelGzip> gz; // the input is the decompressed stream of a plain gzip file
    size_t      end_size() const { return gz ? (size_t)gz->known_size() : size; }
    size_t      slab_bytes = 0, n_slabs = 0;
    std::vector<std::thread> workers;
    std::mutex               m;
    std::condition_variable  cv;
    std::map<size_t, Slab>   ready;
    std::vector<Slab>        free_slabs;
    size_t                   next_to_parse = 0, next_to_take = 0, window = 0;
    uint64_t                 delivered_end = 0; // where the last whole slab handed out by next() ended (= the next slab's first record)
    bool                     stop = false, ended = false;
    bool                     mate_room = false;
    bool                     fasta = false; // records start at lines that begin with '>' (no sequence line can: '>' is no legal letter)
    bool                     raw = false;   // slabs are delivered as text (Slab::text)

    // first byte of the first record at or after p (== size when there is none): a line that begins with '@' and whose
    // next-but-one line begins with '+'
    size_t record_at_or_after(RangeLines& in, size_t p) const
    {
        if (p == 0)
            return 0;
        if (p >= end_size())
            return end_size();
        std::string_view l;
        in.seek(p - 1);
        if (!in.line(l)) // the rest of the line that holds byte p-1 (empty when a line starts exactly at p)
            return end_size();
        for (;;)
        {
            const uint64_t o0 = in.tell();
            if (!in.line(l))
                return end_size();
            if (fasta)
            {
                if (!l.empty() && l[0] == '>')
                    return (size_t)o0;
                continue;
            }
            if (l.empty() || l[0] != '@')
                continue;
            const uint64_t o1 = in.tell();
            std::string_view l1, l2;
            if (!in.line(l1) || !in.line(l2))
                return end_size();
            if (!l2.empty() && l2[0] == '+')
                return (size_t)o0;
            in.seek(o1); // not a record start: go on with the line after it
        }
    }

    // FASTA records of [begin, end): header line, then sequence lines up to the next header (a record that starts before
    // `end` is read to its end).  Same letters, same skipping of white space and digits, same errors as the sequential
    // reader; a ';' header or a first line that is no header hands over to it.
    void parse_fasta(RangeLines& in, size_t begin, size_t end, Slab& out) const
    {
        in.seek(begin);
        std::string_view l;
        while (in.tell() < end)
        {
            const uint64_t rec = in.tell();
            out.rec_at.push_back(rec);
            if (!in.line(l) || l.empty() || l[0] != '>')
            {
                out.irregular = true;
                out.resume_at = rec;
                return;
            }
            out.ids.append(l.data() + 1, l.size() - 1);
            for (;;)
            {
                const uint64_t at_line = in.tell();
                if (!in.line(l))
                    break; // end of the file
                if (!l.empty() && l[0] == '>')
                {
                    in.seek(at_line); // the next record's header
                    break;
                }
                if (!l.empty() && l[0] == ';')
                {
                    out.ids.resize(out.id_off.back());
                    out.bases.resize(out.off.back());
                    out.irregular = true;
                    out.resume_at = rec;
                    return;
                }
                if (all_legal(l.data(), l.size()))
                {
                    out.bases.insert(out.bases.end(), reinterpret_cast<const uint8_t*>(l.data()),
                                     reinterpret_cast<const uint8_t*>(l.data()) + l.size());
                    continue;
                }
                for (char c : l)
                {
                    if (std::isspace((unsigned char)c) || std::isdigit((unsigned char)c))
                        continue;
                    if (!kLegal.ok[(unsigned char)c])
                    {
                        out.ids.resize(out.id_off.back());
                        out.bases.resize(out.off.back());
                        try
                        {
                            bad_letter(std::string_view(&c, 1));
                        }
                        catch (ParseError const& x)
                        {
                            out.error = x.what();
                        }
                        return;
                    }
                    out.bases.push_back((uint8_t)c);
                }
            }
            out.id_off.push_back(out.ids.size());
            out.off.push_back(out.bases.size());
        }
        out.rec_at.push_back(in.tell());
    }

    void parse(RangeLines& in, size_t begin, size_t end, Slab& out) const
    {
        out.ids.reserve((end - begin) / 8);
        // bases of this slab, roughly -- by the slab size, not by this slab's few bytes more or less: the buffers are page-locked blocks that go round
        const size_t expect = fasta ? std::max(end - begin, slab_bytes) + 4096
                              : raw ? std::max(end - begin, slab_bytes + slab_bytes / 16 + 65536) // (one size of page-locked block for text and bases)
                                    : std::max((end - begin) / 2, slab_bytes / 2 + slab_bytes / 32 + 65536);
        out.bases.reserve(mate_room ? 2 * expect + expect / 8 : expect);
        out.rec_at.reserve((end - begin) / 256);
        if (fasta)
        {
            parse_fasta(in, begin, end, out);
            return;
        }
        in.seek(begin);
        while (in.tell() < end)
        {
            const uint64_t rec = in.tell();
            out.rec_at.push_back(rec);
            std::string_view id, seq, plus, qual;
            // (a line view is valid until the next call: id and sequence are stored before the next line is read)
            bool ok = in.line(id) && !id.empty() && id[0] == '@';
            if (ok)
            {
                out.ids.append(id.data() + 1, id.size() - 1);
                ok = in.line(seq);
            }
            size_t seq_len = 0;
            if (ok)
            {
                seq_len = seq.size();
                if (seq_len && seq[0] == '+')
                    ok = false;
                else if (!all_legal(seq.data(), seq_len))
                {
                    out.ids.resize(out.id_off.back());
                    try
                    {
                        bad_letter(seq);
                    }
                    catch (ParseError const& x)
                    {
                        out.error = x.what();
                    }
                    return;
                }
                else
                {
                    const size_t at = out.bases.size();
                    out.bases.resize(at + seq_len);
                    std::memcpy(out.bases.data() + at, seq.data(), seq_len);
                }
            }
            if (ok)
                ok = in.line(plus) && !plus.empty() && plus[0] == '+';
            if (ok && !in.skip_exact_line(seq_len, qual))
                ok = false; // shorter / longer / wrapped quality, or the end of the file: the sequential parser decides
            if (!ok)
            {
                out.ids.resize(out.id_off.back());
                out.bases.resize(out.off.back());
                out.irregular = true; // blank line, wrapped record, truncated tail, ...
                out.resume_at = rec;
                return;
            }
            out.id_off.push_back(out.ids.size());
            out.off.push_back(out.bases.size());
        }
        out.rec_at.push_back(in.tell());
    }

    // raw mode: bytes [begin, end) of the file into the slab's (page-locked) text buffer
    void read_text(size_t begin, size_t end, Slab& out) const
    {
        out.text_at = begin;
        // (one capacity for every piece: the buffers are page-locked blocks that go round -- a piece a few bytes longer than the last
        //  must not need a block of another size)
        out.text.reserve(std::max(end - begin, slab_bytes + slab_bytes / 16 + 65536));
        out.text.resize(end - begin);
        size_t got = 0;
        while (got < end - begin)
        {
            // (a gzip file: bytes of the decompressed stream, which the inflate threads produce -- a damaged stream throws, see work())
            const ssize_t k = gz ? (ssize_t)gz->pread(reinterpret_cast<char*>(out.text.data()) + got, end - begin - got, begin + got)
                                 : ::pread(fd, out.text.data() + got, end - begin - got, (off_t)(begin + got));
            if (k <= 0)
                break; // (the file shrank under us: what is there is delivered; the records end where the text ends)
            got += (size_t)k;
        }
        out.text.resize(got);
        out.text_lines = count_newlines(out.text.data(), got);
    }

    void work()
    {
        // (raw slabs of a plain file: the lines read here are the few around a slab's borders.  Of a gzip stream: the same read-ahead as
        //  the parsing slabs have, so that a damaged stream ends the slabs at the same place whichever way they are delivered)
        RangeLines in = map ? RangeLines(map, size) : RangeLines(fd, gz ? ~0ull : size, gz.get(), raw && !gz ? (64u << 10) : (4u << 20));
        for (;;)
        {
            size_t i;
            Slab   s;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || next_to_parse >= n_slabs || next_to_parse < next_to_take + window; });
                if (stop || next_to_parse >= n_slabs)
                {
                    g_cpu.parse.add_this_thread();
                    return;
                }
                i = next_to_parse++;
                if (!free_slabs.empty())
                {
                    s = std::move(free_slabs.back());
                    free_slabs.pop_back();
                }
            }
            s.ids.clear();
            s.id_off.assign(1, 0);
            s.bases.clear();
            s.off.assign(1, 0);
            s.rec_at.clear();
            s.error.clear();
            s.irregular = false;
            s.resume_after_previous = false;
            s.text.clear();
            s.text_at = 0;
            s.text_lines = 0;
            size_t b = 0;
            bool   have_b = false, last = false;
            try
            {
                b      = record_at_or_after(in, i * slab_bytes);
                have_b = true;
                // (a gzip stream's size turns up when a search runs into its end: no record starts at or after this slab's
                //  first byte, so this slab is the last, and empty)
                last = gz && b >= end_size();
                const size_t e = (!gz && i + 1 == n_slabs) || last ? end_size() : record_at_or_after(in, (i + 1) * slab_bytes);
                if (b < e && raw && e - b >= (3ull << 30))
                {
                    // (a record of gigabytes: offsets inside a piece are 32 bits wide -- the sequential reader's case)
                    s.irregular = true;
                    s.resume_at = b;
                }
                else if (b < e && raw)
                    read_text(b, e, s);
                else if (b < e)
                    parse(in, b, e, s);
                else
                    s.rec_at.assign(1, b);
            }
            catch (std::exception const&)
            {
                // the gzip stream is damaged somewhere in this slab's reach: nothing of the slab is kept; the sequential
                // reader (zlib) takes over at the slab's first record and produces records and message its own way.  The
                // search for that record may itself have run into the damage (it reads ahead in 4 MiB pieces, so whether it
                // does depends on how those fall): then the record is where the slab before this one ended -- its end IS this
                // slab's first record -- and next(), which hands slabs out in file order, fills that in.
                s.ids.clear();
                s.id_off.assign(1, 0);
                s.bases.clear();
                s.off.assign(1, 0);
                s.rec_at.clear();
                s.text.clear();
                s.irregular = true;
                s.resume_at = have_b ? b : 0;
                s.resume_after_previous = !have_b;
            }
            std::lock_guard<std::mutex> lk(m);
            if (last && i + 1 < n_slabs)
                n_slabs = i + 1;
            ready.emplace(i, std::move(s));
            cv.notify_all();
        }
    }
};

ParallelFastq::ParallelFastq(Impl* i) : impl_(i) {}

std::unique_ptr<ParallelFastq> ParallelFastq::open(const std::string& path, unsigned threads, size_t slab_bytes, size_t min_bytes,
                                                   bool mate_room, bool raw)
{
    std::string base = path;
    bool        gz_name = false;
    for (const char* z : { ".gz" })
        if (ends_with(base, z))
        {
            base    = base.substr(0, base.size() - std::strlen(z));
            gz_name = true;
        }
    bool fasta = false;
    for (const char* e : { ".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn", ".fas" })
        fasta = fasta || ends_with(base, e);
    if (!(fasta || ends_with(base, ".fq") || ends_with(base, ".fastq")) || threads == 0)
        return nullptr;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0)
        return nullptr;
    struct stat st;
    unsigned char magic[2] = { 0, 0 };
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < std::max<size_t>(min_bytes, 2) || pread(fd, magic, 2, 0) != 2)
    {
        ::close(fd);
        return nullptr;
    }
    const bool is_gzip = magic[0] == 0x1F && magic[1] == 0x8B;
    std::unique_ptr<ParallelGzip> gz;
    // (text for the device only from plain files.  An ordinary gzip file's decompressed stream can go the same way -- read_text and the
    //  border search below take it -- but measured it is slower: 15-18 against 23-30 Mreads/s; the run is bound by the inflate threads,
    //  which the parse does not hold up, and the raw mode's short window makes them pause)
    if (raw && is_gzip)
    {
        ::close(fd);
        return nullptr;
    }
    if (is_gzip)
    {
        // An ordinary gzip file (blocked gzip has a reader of its own, BgzfSource): inflated by several threads (pgzip.hpp),
        // and the slab parsers work on the decompressed stream as on a file.  The name must say .gz: the sequential reader that
        // takes over at irregular records opens the file by name.
        if (!gz_name || tun().is_set(Knob::no_pgzip) || BgzfSource::open(path))
        {
            ::close(fd);
            return nullptr;
        }
        const bool     e  = tun().is_set(Knob::inflate_threads);
        const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
        unsigned       it = e ? (unsigned)std::max<size_t>(1, tun().size(Knob::inflate_threads, 1)) : std::min(hw, 64u);
        // (hardware_concurrency ignores a cgroup quota; the affinity mask and cpu.max are what classify.cpp's usable_cores reads --
        //  here: a fixed share of what the caller gave its parsers, which is derived from that)
        if (!e)
            it = std::max(2u, 2 * threads); // (the parsers mostly wait for the inflate: measured 12.4 / 17.4 Mreads/s with 12 / 16 threads)
        gz = ParallelGzip::open(path, it, 0, tun().size(Knob::inflate_chunk, 0));
        if (!gz)
        {
            ::close(fd);
            return nullptr;
        }
    }
    Impl* im       = new Impl;
    im->size       = (size_t)st.st_size;
    im->fd         = fd;
    im->slab_bytes = std::max<size_t>(slab_bytes, 1 << 16);
    im->n_slabs    = gz ? ~(size_t)0 : (im->size + im->slab_bytes - 1) / im->slab_bytes;
    im->window     = raw ? threads + 2 : 2 * threads + 2; // (a raw slab is one pread: the readers need little lead, and every slab in flight is page-locked memory)
    im->mate_room  = mate_room;
    im->fasta      = fasta;
    im->raw        = raw;
    if (gz)
    {
        gz->set_retain_limit((uint64_t)(im->window + 3) * im->slab_bytes);
        im->gz = std::move(gz);
    }
    std::unique_ptr<ParallelFastq> pf(new ParallelFastq(im));
    for (unsigned t = 0; t < threads; ++t)
        im->workers.emplace_back([im] { im->work(); });
    return pf;
}

ParallelFastq::~ParallelFastq()
{
    Impl& s = *impl_;
    {
        std::lock_guard<std::mutex> lk(s.m);
        s.stop = true;
        s.cv.notify_all();
    }
    for (auto& t : s.workers)
        t.join();
    if (s.map)
        ::munmap(const_cast<char*>(s.map), s.size);
    ::close(s.fd);
}

bool ParallelFastq::fasta() const { return impl_->fasta; }
void ParallelFastq::recycle(Slab&& used)
{
    std::lock_guard<std::mutex> lk(impl_->m);
    if (impl_->free_slabs.size() < impl_->window + 2)
        impl_->free_slabs.push_back(std::move(used));
}

bool ParallelFastq::next(Slab& out)
{
    Impl&                        s = *impl_;
    std::unique_lock<std::mutex> lk(s.m);
    if (s.ended || s.next_to_take >= s.n_slabs)
        return false;
    s.cv.wait(lk, [&] { return s.ready.count(s.next_to_take) != 0 || s.next_to_take >= s.n_slabs; });
    if (s.next_to_take >= s.n_slabs) // (a gzip stream ended before this slab)
        return false;
    auto it = s.ready.find(s.next_to_take);
    out     = std::move(it->second);
    s.ready.erase(it);
    ++s.next_to_take;
    if (out.irregular && out.resume_after_previous)
        out.resume_at = s.delivered_end; // (0 for the first slab: the file's first record)
    else if (!out.irregular && out.error.empty())
        s.delivered_end = !out.text.empty() ? out.text_at + out.text.size() : (out.rec_at.empty() ? s.delivered_end : out.rec_at.back());
    if (!out.error.empty() || out.irregular)
    {
        s.ended = true; // what the other workers parsed beyond this point is dropped
        s.stop  = true;
    }
    if (s.gz && s.next_to_take * s.slab_bytes > 0) // the slabs still in work begin their search one byte before their range
        s.gz->release_below((uint64_t)s.next_to_take * s.slab_bytes - 1);
    s.cv.notify_all();
    return true;
}

// ---- lines of a plain text file --------------------------------------------------------------------------------------------
uint64_t count_newlines(const uint8_t* p, size_t n)
{
    const __m128i nl = _mm_set1_epi8('\n');
    uint64_t      c  = 0;
    size_t        i  = 0;
    for (; i + 64 <= n; i += 64)
    {
        const unsigned m0 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i)), nl));
        const unsigned m1 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 16)), nl));
        const unsigned m2 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 32)), nl));
        const unsigned m3 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 48)), nl));
        c += (uint64_t)__builtin_popcountll((uint64_t)m0 | (uint64_t)m1 << 16 | (uint64_t)m2 << 32 | (uint64_t)m3 << 48);
    }
    for (; i < n; ++i)
        c += p[i] == '\n';
    return c;
}

// offset (inside [p, p + n)) of the k-th newline, k >= 1; n when there are fewer
static size_t nth_newline(const uint8_t* p, size_t n, uint64_t k)
{
    const __m128i nl = _mm_set1_epi8('\n');
    size_t        i  = 0;
    for (; i + 16 <= n; i += 16)
    {
        unsigned       m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i)), nl));
        const unsigned c = (unsigned)__builtin_popcount(m);
        if (c < k)
        {
            k -= c;
            continue;
        }
        for (;; m &= m - 1)
            if (--k == 0)
                return i + (size_t)__builtin_ctz(m);
    }
    for (; i < n; ++i)
        if (p[i] == '\n' && --k == 0)
            return i;
    return n;
}

struct LineIndex::Impl
{
    static constexpr size_t kChunk = 4u << 20;
    int                     fd = -1;
    uint64_t                size = 0;
    size_t                  n_chunks = 0;
    std::vector<uint64_t>   count;      // newlines of chunk c
    std::vector<uint8_t>    have;       // chunk c is counted
    std::vector<uint64_t>   before;     // newlines before chunk c, for c <= known
    size_t                  known = 0;  // chunks [0, known) are counted and summed
    size_t                  next = 0;   // next chunk a thread takes
    size_t                  asked = 0;  // the chunk the caller's last question fell into (the counters stay a window ahead of it)
    size_t                  window = 512; // 2 GiB
    bool                    stop = false;
    std::mutex              m;
    std::condition_variable cv;
    std::vector<std::thread> threads;
    std::vector<uint8_t>    scratch;    // line_begin's chunk (one caller)
    const uint8_t*          map = nullptr; // the file mapped read-only: counted and copied from the page cache in place

    void work()
    {
        std::vector<uint8_t> buf(kChunk);
        for (;;)
        {
            size_t c;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || next >= n_chunks || next < asked + window; });
                if (stop || next >= n_chunks)
                    return;
                c = next++;
            }
            const uint64_t at  = (uint64_t)c * kChunk;
            const size_t   len = (size_t)std::min<uint64_t>(kChunk, size - at);
            size_t         got = 0;
            while (!map && got < len)
            {
                const ssize_t k = ::pread(fd, buf.data() + got, len - got, (off_t)(at + got));
                if (k <= 0)
                    break; // (the file shrank under us: what is there is counted)
                got += (size_t)k;
            }
            const uint64_t lines = map ? count_newlines(map + at, len) : count_newlines(buf.data(), got);
            std::lock_guard<std::mutex> lk(m);
            count[c] = lines;
            have[c]  = 1;
            while (known < n_chunks && have[known])
            {
                before[known + 1] = before[known] + count[known];
                ++known;
            }
            cv.notify_all();
        }
    }
};

LineIndex::LineIndex(Impl* i) : impl_(i) {}

std::unique_ptr<LineIndex> LineIndex::open(const std::string& path, unsigned threads, size_t min_bytes)
{
    bool by_name = false;
    for (const char* e : { ".fq", ".fastq", ".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn", ".fas" })
        by_name = by_name || ends_with(path, e);
    if (!by_name || threads == 0)
        return nullptr;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0)
        return nullptr;
    struct stat   st;
    unsigned char magic[2] = { 0, 0 };
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < std::max<size_t>(min_bytes, 2) || pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1F && magic[1] == 0x8B)
        || (magic[0] == 'B' && magic[1] == 'Z'))
    {
        ::close(fd);
        return nullptr;
    }
    Impl* im     = new Impl;
    im->fd       = fd;
    im->size     = (uint64_t)st.st_size;
    im->n_chunks = (size_t)((im->size + Impl::kChunk - 1) / Impl::kChunk);
    im->count.assign(im->n_chunks, 0);
    im->have.assign(im->n_chunks, 0);
    im->before.assign(im->n_chunks + 1, 0);
    im->scratch.resize(Impl::kChunk);
    {
        // (the counters read the page cache in place: nothing is copied to count a file's lines)
        void* m = ::mmap(nullptr, im->size, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED)
        {
            ::madvise(m, im->size, MADV_SEQUENTIAL);
            im->map = static_cast<const uint8_t*>(m);
        }
    }
    for (unsigned t = 0; t < threads; ++t)
        im->threads.emplace_back([im] {
            im->work();
            g_cpu.parse.add_this_thread();
        });
    return std::unique_ptr<LineIndex>(new LineIndex(im));
}

LineIndex::~LineIndex()
{
    Impl& s = *impl_;
    {
        std::lock_guard<std::mutex> lk(s.m);
        s.stop = true;
    }
    s.cv.notify_all();
    for (auto& t : s.threads)
        t.join();
    if (s.map)
        ::munmap(const_cast<uint8_t*>(s.map), s.size);
    ::close(s.fd);
}

uint64_t LineIndex::size() const { return impl_->size; }

uint64_t LineIndex::line_begin(uint64_t line)
{
    Impl& s = *impl_;
    if (line == 0)
        return 0;
    size_t   c = 0;
    uint64_t k = 0; // the line begins behind the k-th newline of chunk c
    {
        std::unique_lock<std::mutex> lk(s.m);
        // the chunk that holds newline number `line` (1-based): the first c with before[c + 1] >= line
        for (;;)
        {
            if (s.before[s.known] >= line)
                break;
            if (s.known >= s.n_chunks)
                return kNoSuchLine;
            s.asked = s.known; // (keeps the counters going)
            s.cv.notify_all();
            const size_t was = s.known;
            s.cv.wait(lk, [&] { return s.known > was; });
        }
        c = (size_t)(std::lower_bound(s.before.begin() + 1, s.before.begin() + s.known + 1, line) - (s.before.begin() + 1));
        k = line - s.before[c];
        if (c > s.asked)
        {
            s.asked = c;
            s.cv.notify_all();
        }
    }
    const uint64_t at  = (uint64_t)c * Impl::kChunk;
    const size_t   len = (size_t)std::min<uint64_t>(Impl::kChunk, s.size - at);
    size_t         got = s.map ? len : 0;
    while (got < len)
    {
        const ssize_t r = ::pread(s.fd, s.scratch.data() + got, len - got, (off_t)(at + got));
        if (r <= 0)
            break;
        got += (size_t)r;
    }
    const size_t p = nth_newline(s.map ? s.map + at : s.scratch.data(), got, k);
    return p >= got ? kNoSuchLine : at + p + 1;
}

bool LineIndex::read(uint64_t begin, uint64_t end, ByteBuf& dst, size_t reserve) const
{
    dst.reserve(std::max<size_t>(reserve, (size_t)(end - begin)));
    dst.resize((size_t)(end - begin));
    size_t got = 0;
    while (got < dst.size())
    {
        const ssize_t k = ::pread(impl_->fd, dst.data() + got, dst.size() - got, (off_t)(begin + got));
        if (k <= 0)
            return false;
        got += (size_t)k;
    }
    return true;
}

} // namespace gnhost
