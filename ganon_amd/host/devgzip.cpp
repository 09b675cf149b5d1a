// devgzip.cpp -- a gzip-compressed FASTQ / FASTA file as pieces of text in device memory (DeviceTextSource, backend.hpp).
//
// The reference reads `reads.fq.gz` through one zlib stream (seqan3::sequence_file_input in parse_reads,
// /root/reference/src/ganon-classify/GanonClassify.cpp:1220-1287,1433).  Here the file's bytes go to the device as they are -- a
// quarter of the text --, libganon_hip's gn_inflate_* (csrc/gn_inflate.hip) inflates them there, gn_inflate_cuts says where records
// begin, and the pieces between cuts are handed to the workers as device pointers: the text never crosses the link, the host sees
// the records' header lines only (gn_stream_fastq_headers).
//
//   reader threads: pread() the file in 8 MiB blocks into page-locked buffers, ahead of the feeder
//   feeder thread:  gn_inflate_feed in file order (the whole compressed file becomes resident)
//   stepper thread: gn_inflate_step when a step's bytes are there, gn_inflate_cuts, pieces into a queue; the record the step's end
//                   cuts is carried into the next step's text (gn_inflate_set_carry); a step's buffer is written again two steps
//                   later, so the stepper waits until every piece of step k is released before it runs step k + 2
// The mate file of a pair is opened `by_lines`: no stepper thread; the caller asks for "the next n lines" (the lines the first file's
// piece holds: parse_reads pairs records by number, GanonClassify.cpp:1240-1252) and the step that has to run for it runs in that call.
// Several devices: every distinct device of the run gets an inflater of the file (each is fed the whole compressed file over its own
// link) and they take the file's steps in turn -- gn_inflate_set_turns / gn_inflate_handoff: a step's decode depends on nothing before
// it, the step before hands over 32 KiB of window, a position, a CRC and the carried record.  The text of step k lies on device k mod N,
// where that device's worker classifies it.  With ONE inflating device a .gz file ran at 100-117 Mreads/s whatever the number of GPUs
// (DESIGN 7).  The mate file of a pair (by_lines) is inflated the same way; a pair's two pieces may then lie on two devices
// (gn_stream_upload_text_pair_devices).
// Anything the device path refuses (GN_ERANGE: damaged data, a wrong ISIZE, expansion beyond its buffers, ...) ends the source with
// an error text; the caller's sequential zlib reader continues at delivered() and produces the records and the message from there.
#include "backend.hpp"
#include "cpu_tally.hpp"
#include "tunables.hpp"

#include "../../include/ganon_hip.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gnhost
{
namespace
{

bool ends_with(const std::string& s, const char* e)
{
    const size_t n = std::strlen(e);
    return s.size() >= n && s.compare(s.size() - n, n, e) == 0;
}

class DeviceGzip final : public DeviceTextSource
{
public:
    DeviceGzip() = default;
    ~DeviceGzip() override
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : readers_)
            if (t.joinable())
                t.join();
        for (auto& t : feeders_)
            if (t.joinable())
                t.join();
        if (stepper_.joinable())
            stepper_.join();
        // pieces still out there keep their hold; the device buffers go with the inflater: wait for them
        {
            std::deque<DeviceTextPiece> mine;
            {
                std::lock_guard<std::mutex> lk(m_);
                mine.swap(q_);
            }
        }
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] {
                for (auto const& h : held_)
                    if (h[0] || h[1])
                        return false;
                return true;
            });
        }
        for (gn_inflate* z : zs_)
            gn_inflate_destroy(z);
        for (void* p : blocks_)
            gn_pinned_free(p);
        if (fd_ >= 0)
            ::close(fd_);
    }

    bool start(const std::string& path, const std::vector<int>& devices, size_t piece_bytes, size_t min_bytes, bool by_lines)
    {
        if (devices.empty())
            return false;
        const int device = devices.front();
        by_lines_ = by_lines;
        std::string base = path;
        if (!ends_with(base, ".gz"))
            return false;
        base = base.substr(0, base.size() - 3);
        for (const char* e : { ".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn", ".fas" })
            fasta_ = fasta_ || ends_with(base, e);
        if (!(fasta_ || ends_with(base, ".fq") || ends_with(base, ".fastq")))
            return false;
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0)
            return false;
        struct stat st;
        uint8_t     h[18];
        if (fstat(fd_, &st) != 0 || (size_t)st.st_size < std::max<size_t>(min_bytes, 64) || pread(fd_, h, 18, 0) != 18)
            return false;
        if (h[0] != 0x1F || h[1] != 0x8B || h[2] != 8)
            return false;
        // (blocked gzip -- BGZF, members of at most 64 KiB -- goes the same way: the search finds member headers as well as block headers)
        size_     = (uint64_t)st.st_size;
        device_   = device;
        piece_    = std::max<size_t>(piece_bytes, 1 << 16);
        const std::string* sb = tun().str(Knob::device_inflate_step);
        const std::string* cb = tun().str(Knob::device_inflate_chunk);
        const auto t_open = std::chrono::steady_clock::now();
        t0_ = t_open;
        l_step_ = sb ? (uint64_t)std::atoll(sb->c_str()) : (size_ < (1536ull << 20) ? 128ull << 20 : 256ull << 20);
        // how many inflaters: one per distinct device (a mate file: one), no more than the file has steps; $GANON_HOST_DEVICE_INFLATE_TURNS
        // says otherwise (tests put several on the one GPU of a box)
        size_t n_inf = tun().size(Knob::device_inflate_turns, devices.size());
        n_inf        = std::max<size_t>(1, std::min<size_t>(n_inf, (size_t)((size_ + l_step_ - 1) / l_step_)));
        {
            // (creating an inflater clears gigabytes of device memory, ~0.1 s: every device does that at the same time)
            std::vector<gn_inflate*> made(n_inf, nullptr);
            std::vector<std::thread> th;
            const uint32_t           chunk_b = cb ? (uint32_t)std::atoll(cb->c_str()) : 0;
            const uint64_t           step_b  = sb ? (uint64_t)std::atoll(sb->c_str()) : 0;
            for (size_t i = 1; i < n_inf; ++i)
                th.emplace_back([&, i] { (void)gn_inflate_create(devices[i % devices.size()], size_, chunk_b, step_b, &made[i]); });
            (void)gn_inflate_create(devices[0], size_, chunk_b, step_b, &made[0]);
            for (auto& t : th)
                t.join();
            for (size_t i = 0; i < n_inf; ++i)
            {
                if (!made[i]) // (no room on that device beside the filters: the ones before it share the file)
                {
                    for (size_t j = i + 1; j < n_inf; ++j)
                        if (made[j])
                            gn_inflate_destroy(made[j]);
                    break;
                }
                zs_.push_back(made[i]);
                devs_.push_back(devices[i % devices.size()]);
            }
        }
        if (zs_.empty())
            return false; // (no room for the file and the step buffers beside the filters: the host inflater takes it)
        for (size_t i = 0; i < zs_.size() && zs_.size() > 1; ++i)
            if (gn_inflate_set_turns(zs_[i], (uint32_t)zs_.size(), (uint32_t)i) != GN_OK)
                return false;
        z_ = zs_.front();
        held_.assign(zs_.size(), { 0, 0 });
        fed_blocks_of_.assign(zs_.size(), 0);
        fed_bytes_of_.assign(zs_.size(), 0);
        sec_create_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        n_blocks_ = (size_ + kBlock - 1) / kBlock;
        for (unsigned i = 0; i < kRing; ++i)
        {
            void* p = nullptr;
            if (gn_pinned_alloc(kBlock, &p) != GN_OK)
                return false;
            blocks_.push_back(p);
        }
        sec_pinned_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count() - sec_create_;
        const unsigned n_readers = (unsigned)std::min<uint64_t>(3, n_blocks_);
        for (unsigned t = 0; t < n_readers; ++t)
            readers_.emplace_back([this] { read_loop(); g_cpu.inflate.add_this_thread(); });
        for (size_t i = 0; i < zs_.size(); ++i)
            feeders_.emplace_back([this, i] { feed_loop(i); g_cpu.inflate.add_this_thread(); });
        if (!by_lines_)
            stepper_ = std::thread([this] { step_loop(); g_cpu.inflate.add_this_thread(); });
        return true;
    }

    bool next(DeviceTextPiece& out, std::string& err) override
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty() || finished_ || stop_; });
        if (q_.empty())
        {
            err = error_;
            return false;
        }
        out = std::move(q_.front());
        q_.pop_front();
        delivered_ = out.at + out.bytes;
        cv_.notify_all();
        return true;
    }
    // by_lines: the step that is needed runs here, on the caller's thread
    bool next_lines(uint64_t lines, DeviceTextPiece& out, std::string& err) override
    {
        if (!by_lines_)
        {
            err = "not a source by lines";
            return false;
        }
        for (;;)
        {
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!error_.empty())
                {
                    err = error_;
                    return false;
                }
            }
            if (l_eof_)
                return false;
            if (!l_have_)
            {
                l_want_       = std::min<uint64_t>(size_, l_want_ + l_step_ + (4ull << 20));
                // the inflaters of the file take its steps in turn (as in step_loop)
                const size_t who = l_step_no_ % zs_.size();
                if (l_own_steps_.size() != zs_.size())
                    l_own_steps_.assign(zs_.size(), 0);
                const int buf = (int)(l_own_steps_[who] & 1u);
                {
                    std::unique_lock<std::mutex> lk(m_);
                    while (!cv_.wait_for(lk, std::chrono::milliseconds(1), [&] { return stop_ || finished_ || (may_run() && fed_bytes_of_[who] >= l_want_ && held_[who][buf] == 0); }))
                        ;
                    if (stop_ || finished_)
                    {
                        err = error_;
                        return false;
                    }
                }
                if (zs_.size() > 1 && l_step_no_ && gn_inflate_handoff(zs_[(l_step_no_ - 1) % zs_.size()], zs_[who]) != GN_OK)
                {
                    std::lock_guard<std::mutex> lk(m_);
                    fail_locked(gn_last_error());
                    err = error_;
                    return false;
                }
                l_z_ = zs_[who];
                l_who_ = who;
                ++l_own_steps_[who];
                uint64_t n_text = 0;
                int      done   = 0;
                if (gn_inflate_step(l_z_, &n_text, &done) != GN_OK)
                {
                    std::lock_guard<std::mutex> lk(m_);
                    fail_locked(gn_last_error());
                    err = error_;
                    return false;
                }
                if (sec_first_ == 0)
                    sec_first_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
                uint64_t dn = 0;
                gn_inflate_text_device(l_z_, &l_text_, &dn);
                l_n_      = n_text;
                l_served_ = 0;
                l_lines_  = 0;
                l_done_   = done != 0;
                l_buf_    = buf;
                l_have_   = true;
                ++l_step_no_;
            }
            uint64_t off = 0, total = 0;
            const uint64_t ask = lines == ~0ull ? ~0ull : l_lines_ + lines;
            if (gn_inflate_cut_at_lines(l_z_, &ask, lines == ~0ull ? 0u : 1u, &off, &total) != GN_OK)
            {
                std::lock_guard<std::mutex> lk(m_);
                fail_locked(gn_last_error());
                err = error_;
                return false;
            }
            const bool whole = lines != ~0ull && off != ~0ull;
            if (!whole && !l_done_ && lines != ~0ull)
            {
                // the text at hand ends before the lines do: what is left of it begins the next step's text
                if (gn_inflate_set_carry(l_z_, l_n_ - l_served_) != GN_OK)
                {
                    std::lock_guard<std::mutex> lk(m_);
                    fail_locked(gn_last_error());
                    err = error_;
                    return false;
                }
                l_at_ += l_served_;
                l_have_ = false;
                continue;
            }
            const uint64_t end = whole ? off : l_n_;
            out.dev    = l_text_ + l_served_;
            out.bytes  = end - l_served_;
            out.at     = l_at_ + l_served_;
            out.device = devs_[l_who_];
            out.lines  = whole ? lines : total - l_lines_;
            {
                std::lock_guard<std::mutex> lk(m_);
                const int    buf = l_buf_;
                const size_t who = l_who_;
                ++held_[who][buf];
                out.hold = std::shared_ptr<void>(nullptr, [this, who, buf](void*) {
                    std::lock_guard<std::mutex> lk2(m_);
                    --held_[who][buf];
                    cv_.notify_all();
                });
            }
            l_lines_ += whole ? lines : total - l_lines_;
            l_served_ = end;
            delivered_ = out.at + out.bytes;
            if (!whole)
            {
                // the stream's end -- or the caller asked for "whatever is left" (lines == ~0: the rest of the CURRENT step's text, see
                // backend.hpp) while the stream has further steps: those stay available, the next call inflates on with nothing carried
                if (l_done_)
                    l_eof_ = true;
                else
                {
                    if (gn_inflate_set_carry(l_z_, 0) != GN_OK)
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        fail_locked(gn_last_error());
                        err = error_;
                        return false;
                    }
                    l_at_ += l_served_;
                    l_have_ = false;
                }
            }
            return out.bytes != 0 || whole;
        }
    }
    void go(const std::atomic<bool>* run) override
    {
        std::lock_guard<std::mutex> lk(m_);
        go_  = true;
        run_ = run;
        cv_.notify_all();
    }
    bool may_run() const { return !run_ || run_->load(); }
    uint64_t    delivered() const override { return delivered_; }
    bool        fasta() const override { return fasta_; }
    std::string report() const override
    {
        gn_inflate_stats st;
        std::memset(&st, 0, sizeof(st));
        for (gn_inflate* z : zs_)
        {
            gn_inflate_stats one;
            std::memset(&one, 0, sizeof(one));
            gn_inflate_get_stats(z, &one);
            st.steps += one.steps;
            st.chunks += one.chunks;
            st.fixups += one.fixups;
            st.markers += one.markers;
            st.members += one.members;
            st.text_bytes += one.text_bytes;
            st.ms_decode += one.ms_decode;
            st.ms_chain += one.ms_chain;
            st.ms_resolve += one.ms_resolve;
            st.ms_step_wall += one.ms_step_wall;
        }
        std::ostringstream o;
        o << "device inflate: ";
        if (zs_.size() > 1)
        {
            o << zs_.size() << " inflaters taking turns on devices";
            for (int d : devs_)
                o << ' ' << d;
            o << ", ";
        }
        o << st.steps << " steps, " << st.chunks << " chunks, " << st.fixups << " fix-ups, " << st.members << " members, "
          << st.text_bytes << " bytes of text; device ms: decode " << st.ms_decode << ", order " << st.ms_chain << ", windows+resolve " << st.ms_resolve
          << "; step calls " << st.ms_step_wall << " ms; opening: device buffers " << sec_create_ << " s, page-locked blocks " << sec_pinned_
          << " s, first text after " << sec_first_ << " s, all fed after " << sec_fed_ << " s";
        return o.str();
    }

private:
    static constexpr size_t   kBlock = 8u << 20;
    static constexpr unsigned kRing  = 8;

    void read_loop()
    {
        for (;;)
        {
            uint64_t b;
            {
                std::unique_lock<std::mutex> lk(m_);
                // block b uses ring slot b % kRing: free once block b - kRing is fed
                cv_.wait(lk, [&] { return stop_ || next_read_ >= n_blocks_ || next_read_ < fed_blocks_ + kRing; }); // (fed_blocks_: by EVERY inflater)
                if (stop_ || next_read_ >= n_blocks_)
                    return;
                b = next_read_++;
            }
            const uint64_t off = b * kBlock, n = std::min<uint64_t>(kBlock, size_ - off);
            uint8_t*       dst = static_cast<uint8_t*>(blocks_[b % kRing]);
            uint64_t       got = 0;
            bool           ok  = true;
            while (got < n)
            {
                const ssize_t r = pread(fd_, dst + got, n - got, (off_t)(off + got));
                if (r <= 0)
                {
                    ok = false;
                    break;
                }
                got += (uint64_t)r;
            }
            std::lock_guard<std::mutex> lk(m_);
            if (!ok)
                fail_locked("cannot read the file");
            read_done_[b] = true;
            cv_.notify_all();
        }
    }

    // one feeder per inflater: the whole compressed file goes to every device, each over its own link
    void feed_loop(size_t who)
    {
        for (uint64_t b = 0; b < n_blocks_; ++b)
        {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || finished_ || read_done_.count(b); });
                if (stop_ || finished_)
                    return;
            }
            const uint64_t off = b * kBlock, n = std::min<uint64_t>(kBlock, size_ - off);
            const int      rc  = gn_inflate_feed(zs_[who], static_cast<const uint8_t*>(blocks_[b % kRing]), n);
            std::lock_guard<std::mutex> lk(m_);
            if (rc != GN_OK)
            {
                fail_locked(gn_last_error());
                return;
            }
            fed_blocks_of_[who] = b + 1;
            fed_bytes_of_[who]  = off + n;
            // a block's ring slot is free once every inflater has it
            const uint64_t all_blocks = *std::min_element(fed_blocks_of_.begin(), fed_blocks_of_.end());
            while (fed_blocks_ < all_blocks)
                read_done_.erase(fed_blocks_++);
            fed_bytes_ = *std::min_element(fed_bytes_of_.begin(), fed_bytes_of_.end());
            if (fed_blocks_ == n_blocks_ && sec_fed_ == 0)
                sec_fed_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
            cv_.notify_all();
        }
    }

    void fail_locked(const std::string& why)
    {
        if (error_.empty())
            error_ = why;
        finished_ = true;
        cv_.notify_all();
    }

    void step_loop()
    {
        const uint32_t        lpr = fasta_ ? 2u : 4u;
        uint64_t              stream_at = 0; // decompressed offset of the next step's text[0] (the carried bytes included)
        std::vector<uint64_t> cuts, cut_lines;
        unsigned              step_no = 0;
        const std::string*    sb   = tun().str(Knob::device_inflate_step);
        const uint64_t        step = sb ? (uint64_t)std::atoll(sb->c_str()) : (size_ < (1536ull << 20) ? 128ull << 20 : 256ull << 20);
        uint64_t              want = 0; // compressed bytes the next step should find
        const size_t          n_inf = zs_.size();
        std::vector<unsigned> own_steps(n_inf, 0); // (an inflater writes its two text buffers in turn: the buffer of ITS step before last must be released)
        for (;;)
        {
            want = std::min<uint64_t>(size_, want + step + (4ull << 20));
            const size_t who = step_no % n_inf;
            gn_inflate*  z   = zs_[who];
            const int    buf = (int)(own_steps[who] & 1u);
            {
                std::unique_lock<std::mutex> lk(m_);
                while (!cv_.wait_for(lk, std::chrono::milliseconds(1), [&] { return stop_ || finished_ || (go_ && may_run() && fed_bytes_of_[who] >= want && held_[who][buf] == 0); }))
                    ;
                if (stop_ || finished_)
                    return;
            }
            // what this step needs of the one before it (another inflater's): position, member CRC, window, the carried record
            if (n_inf > 1 && step_no && gn_inflate_handoff(zs_[(step_no - 1) % n_inf], z) != GN_OK)
            {
                std::lock_guard<std::mutex> lk(m_);
                fail_locked(gn_last_error());
                return;
            }
            uint64_t n_text = 0;
            int      done   = 0;
            if (gn_inflate_step(z, &n_text, &done) != GN_OK)
            {
                std::lock_guard<std::mutex> lk(m_);
                fail_locked(gn_last_error());
                return;
            }
            ++own_steps[who];
            if (sec_first_ == 0)
                sec_first_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
            const uint8_t* dtext = nullptr;
            uint64_t       dn    = 0;
            gn_inflate_text_device(z, &dtext, &dn);
            uint32_t n_cuts = 0;
            cuts.resize((size_t)(n_text / piece_) + 2);
            cut_lines.resize(cuts.size());
            if (n_text && gn_inflate_cuts_lines(z, lpr, piece_, cuts.data(), cut_lines.data(), (uint32_t)cuts.size(), &n_cuts) != GN_OK)
            {
                std::lock_guard<std::mutex> lk(m_);
                fail_locked(gn_last_error());
                return;
            }
            cuts.resize(n_cuts);
            cut_lines.resize(n_cuts);
            uint64_t last = n_cuts ? cuts.back() : 0;
            if (done && last < n_text) // the file's last bytes are no whole record: the tokeniser says so, the sequential reader takes them
            {
                cuts.push_back(n_text);
                cut_lines.push_back(~0ull);
                last = n_text;
            }
            const uint64_t tail = n_text - last;
            if (!done && gn_inflate_set_carry(z, tail) != GN_OK)
            {
                // (a "record" larger than a quarter of a step's buffer: not four-line FASTQ; the sequential reader says what it is)
                std::lock_guard<std::mutex> lk(m_);
                fail_locked(gn_last_error());
                return;
            }
            {
                std::unique_lock<std::mutex> lk(m_);
                uint64_t                     from = 0, lines_before = 0;
                for (size_t ci = 0; ci < cuts.size(); ++ci)
                {
                    const uint64_t  c = cuts[ci];
                    DeviceTextPiece p;
                    p.dev    = dtext + from;
                    p.bytes  = c - from;
                    p.at     = stream_at + from;
                    p.device = devs_[who];
                    p.lines  = cut_lines[ci] == ~0ull ? ~0ull : cut_lines[ci] - lines_before;
                    lines_before = cut_lines[ci] == ~0ull ? lines_before : cut_lines[ci];
                    ++held_[who][buf];
                    p.hold = std::shared_ptr<void>(nullptr, [this, who, buf](void*) {
                        std::lock_guard<std::mutex> lk2(m_);
                        --held_[who][buf];
                        cv_.notify_all();
                    });
                    q_.push_back(std::move(p));
                    from = c;
                }
                stream_at += last;
                if (done)
                    finished_ = true;
                cv_.notify_all();
                if (done)
                    return;
            }
            ++step_no;
        }
    }

    int                   fd_ = -1;
    uint64_t              size_ = 0, n_blocks_ = 0;
    int                   device_ = 0;
    size_t                piece_ = 48u << 20;
    bool                  fasta_ = false;
    gn_inflate*           z_ = nullptr;     // the first inflater (the only one of a by_lines source)
    std::vector<gn_inflate*> zs_;           // every inflater of the file, in turn order
    std::vector<int>      devs_;            // ... and its device
    std::vector<void*>    blocks_;
    std::vector<std::thread> readers_;
    std::vector<std::thread> feeders_;
    std::thread           stepper_;
    mutable std::mutex    m_;
    std::condition_variable cv_;
    bool                  stop_ = false, finished_ = false, go_ = false;
    const std::atomic<bool>* run_ = nullptr;
    std::string           error_;
    uint64_t              next_read_ = 0, fed_blocks_ = 0, fed_bytes_ = 0;
    std::map<uint64_t, bool> read_done_;
    std::deque<DeviceTextPiece> q_;
    std::vector<std::array<uint64_t, 2>> held_;         // per inflater and text buffer: pieces out there
    std::vector<uint64_t> fed_blocks_of_, fed_bytes_of_; // per inflater; fed_blocks_ / fed_bytes_ = what EVERY inflater has
    std::atomic<uint64_t> delivered_{ 0 };
    std::chrono::steady_clock::time_point t0_;
    double sec_create_ = 0, sec_pinned_ = 0, sec_first_ = 0, sec_fed_ = 0;
    // by_lines state (the caller's thread only)
    bool           by_lines_ = false, l_have_ = false, l_done_ = false, l_eof_ = false;
    const uint8_t* l_text_ = nullptr;
    uint64_t       l_n_ = 0, l_served_ = 0, l_lines_ = 0, l_at_ = 0, l_want_ = 0;
    uint64_t       l_step_ = 256ull << 20;
    unsigned       l_step_no_ = 0;
    gn_inflate*    l_z_ = nullptr;          // the inflater whose step's text is at hand
    size_t         l_who_ = 0;
    std::vector<unsigned> l_own_steps_;
    int            l_buf_ = 0;
};

} // namespace

std::unique_ptr<DeviceTextSource> open_device_gzip(const std::string& path, const std::vector<int>& devices, size_t piece_bytes, size_t min_bytes, bool by_lines)
{
    std::unique_ptr<DeviceGzip> g(new DeviceGzip());
    if (!g->start(path, devices, piece_bytes, min_bytes, by_lines))
        return nullptr;
    return g;
}

} // namespace gnhost
