// reader.cpp -- the reader task of ganon-classify: files -> large numbered batches (split from classify.cpp in round 6; pipeline.hpp).
// The reference has ONE parser thread (GanonClassify.cpp:1220-1287, started at :1436-1441) that fills a queue of 400-read batches; here it
// hands out slabs of raw text (the device finds the records), parsed slabs, device-resident pieces of an inflated .gz, or sequentially
// parsed batches -- always numbered in input order, so that the ordered post stage writes what a --threads 1 run of the reference writes.
#include "pipeline.hpp"

#include "cpu_tally.hpp"
#include "seq_io.hpp"
#include "startup.hpp"
#include "tunables.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <sched.h>
#include <sys/stat.h>

namespace gnhost
{

size_t batch_reads()
{
    static const size_t n = std::max<size_t>(1, tun().size(Knob::batch_reads, 1u << 20));
    return n;
}

// appends the mates-2 region behind the mates-1 region and rebases its offsets
void finalize_batch(ReadBatch& rb, ByteBuf& bases2)
{
    if (!rb.paired)
        return;
    const uint64_t base = rb.bases.size();
    for (auto& o : rb.off2)
        o += base;
    rb.bases.insert(rb.bases.end(), bases2.begin(), bases2.end());
    bases2.clear();
}

namespace
{


// Second file of a pair, parsed on its own thread (for gzip input the reader is inflate-bound: two files, two inflate
// streams).  Only sequences are kept (ids come from file 1, :1243-1252); records arrive in blocks through a bounded
// queue; a parse error is delivered in place, after the records that precede it.
class MateStream
{
public:
    // the file is opened here, on the caller's thread: a file that cannot be opened fails before any record is read
    explicit MateStream(const std::string& path, uint64_t start_offset = 0)
      : q_(8), in_(new SeqReader(path, start_offset)), worker_([this] { run(); })
    {
    }
    ~MateStream()
    {
        stop_ = true;
        Block b;
        while (q_.pop(b)) {} // unblock the producer
        worker_.join();
    }
    // appends the next mate to `bases`; false at end of file; throws the file's ParseError where it occurred
    bool next(ByteBuf& bases)
    {
        while (pos_ == cur_.off.size() - 1)
        {
            if (cur_.last)
            {
                if (!cur_.error.empty())
                {
                    std::string e;
                    e.swap(cur_.error);
                    throw ParseError(e);
                }
                return false;
            }
            if (!q_.pop(cur_))
                return false;
            pos_ = 0;
        }
        bases.insert(bases.end(), cur_.bases.begin() + cur_.off[pos_], cur_.bases.begin() + cur_.off[pos_ + 1]);
        ++pos_;
        return true;
    }

private:
    struct Block
    {
        ByteBuf               bases;
        std::vector<uint64_t> off{ 0 };
        bool                  last = false;
        std::string           error; // with last: the ParseError that ended the file
    };
    void run()
    {
        Block b;
        try
        {
            std::string id;
            while (!stop_)
            {
                id.clear();
                if (!in_->next(id, b.bases))
                    break;
                b.off.push_back(b.bases.size());
                if (b.off.size() > 65536 || b.bases.size() >= (16u << 20))
                {
                    q_.push(std::move(b));
                    b = Block();
                }
            }
        }
        catch (ParseError const& e)
        {
            b.error = e.what();
        }
        b.last = true;
        q_.push(std::move(b));
        q_.done();
    }
    BoundedQueue<Block>        q_;
    std::atomic<bool>          stop_{ false };
    Block                      cur_;
    size_t                     pos_ = 0;
    std::unique_ptr<SeqReader> in_;
    std::thread                worker_; // last member: everything above exists when it starts
};

} // namespace

// cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a container that shows 256 CPUs
// and grants 16 runs 16 threads' worth of work, however many threads there are)
unsigned usable_cores()
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        n = (unsigned)CPU_COUNT(&set);
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string   quota;
    double        period = 0;
    if (f >> quota >> period && quota != "max" && period > 0)
    {
        const double q = std::atof(quota.c_str()) / period;
        if (q >= 1 && q < n)
            n = (unsigned)q;
    }
    return n ? n : 1;
}

namespace
{

// Second mates of a batch of pairs: the reader thread only notes which records of file 2's slabs belong to the batch; a few
// threads of this pool append them behind the first mates (one copy, straight into the batch's page-locked buffer), fill in
// the mate offsets and hand the batch to the device workers.  Batches carry their input-order number and reach
// the queue through deliver(), which keeps that order.
class MateCopier
{
public:
    struct Part
    {
        std::shared_ptr<ParallelFastq::Slab> slab;
        size_t                               first, count;
    };
    MateCopier(BatchQueue& out, unsigned threads) : out_(out)
    {
        for (unsigned t = 0; t < threads; ++t)
            th_.emplace_back([this] { run(); });
    }
    ~MateCopier()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_)
            t.join();
    }
    void submit(ReadBatch&& rb, std::vector<Part>&& parts)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.size() + busy_ < 2 * th_.size() + 2; });
        q_.emplace_back(std::move(rb), std::move(parts));
        loaders_.emplace_back();
        cv_.notify_all();
    }
    // a batch whose second half is fetched by `load` on one of the threads (the mate file's piece of a pair that travels as text)
    void submit(ReadBatch&& rb, std::function<void(ReadBatch&)> load)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.size() + busy_ < 2 * th_.size() + 2; });
        q_.emplace_back(std::move(rb), std::vector<Part>());
        loaders_.push_back(std::move(load));
        cv_.notify_all();
    }
    void drain() // every submitted batch is in the queue
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.empty() && busy_ == 0; });
    }
    // Batches enter the queue in input order, whichever thread finished them (the reader's own batches come through here
    // too): a device worker that held batch k+5 while batch k was still being copied would wait for its turn for ever.
    void deliver(ReadBatch&& rb)
    {
        std::lock_guard<std::mutex> lk(order_m_);
        const uint64_t seq = rb.seq;
        held_.emplace(seq, std::move(rb));
        for (auto it = held_.find(next_seq_); it != held_.end(); it = held_.find(next_seq_))
        {
            out_.push(std::move(it->second));
            held_.erase(it);
            ++next_seq_;
        }
    }
    // appends the parts' bases behind dst and their end offsets (absolute positions in dst) to off2
    static void materialise(const std::vector<Part>& parts, ByteBuf& dst, std::vector<uint64_t>& off2)
    {
        for (const Part& p : parts)
        {
            const auto&    sl   = *p.slab;
            const uint64_t from = sl.off[p.first], at = dst.size();
            dst.insert(dst.end(), sl.bases.begin() + from, sl.bases.begin() + sl.off[p.first + p.count]);
            for (size_t j = 1; j <= p.count; ++j)
                off2.push_back(at + (sl.off[p.first + j] - from));
        }
    }

private:
    void run()
    {
        for (;;)
        {
            std::pair<ReadBatch, std::vector<Part>> job;
            std::function<void(ReadBatch&)>         load;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty() || stop_; });
                if (q_.empty())
                {
                    g_cpu.mate.add_this_thread();
                    return;
                }
                job = std::move(q_.front());
                q_.pop_front();
                load = std::move(loaders_.front());
                loaders_.pop_front();
                ++busy_;
            }
            ReadBatch& rb = job.first;
            if (load)
                load(rb);
            else
            {
                rb.off2.assign(1, rb.bases.size()); // mates follow the first mates in the same buffer (finalize_batch's layout)
                rb.off2.reserve(rb.size() + 1);
                materialise(job.second, rb.bases, rb.off2);
                job.second.clear(); // (releases the slabs: the last user hands a slab back to its parser)
            }
            deliver(std::move(rb));
            {
                std::lock_guard<std::mutex> lk(m_);
                --busy_;
            }
            cv_.notify_all();
        }
    }
    BatchQueue&                                         out_;
    std::mutex                                          order_m_;
    std::map<uint64_t, ReadBatch>                       held_;
    uint64_t                                            next_seq_ = 0;
    std::mutex                                          m_;
    std::condition_variable                             cv_;
    std::deque<std::pair<ReadBatch, std::vector<Part>>> q_;
    std::deque<std::function<void(ReadBatch&)>>         loaders_; // (one per entry of q_; empty: the parts are copied)
    size_t                                              busy_ = 0;
    bool                                                stop_ = false;
    std::vector<std::thread>                            th_;
};

} // namespace

// the reader thread (:1220-1287): files -> large batches, numbered in input order.
// Uncompressed four-line FASTQ is parsed by several threads (ParallelFastq, seq_io.hpp): for single-end files a parsed
// slab IS the batch (no copy); for pairs the slabs of file 1 become batches and the mates are copied next to them from
// the slabs of file 2.  Everything else -- compressed input, FASTA, wrapped records, and whatever follows the first
// record the parallel parser does not take -- goes through the sequential reader, from the byte where the slabs stopped.
Cleanups g_cleanups;
void Cleanups::add(std::thread t)
{
    std::lock_guard<std::mutex> lk(m);
    th.push_back(std::move(t));
}
void Cleanups::join_all()
{
    std::vector<std::thread> mine;
    {
        std::lock_guard<std::mutex> lk(m);
        mine.swap(th);
    }
    for (auto& t : mine)
        if (fast_exit()) // (the process ends with _Exit in a moment: nobody needs the memory they are still freeing)
            t.detach();
        else
            t.join();
}


DeviceGate g_devices_ready;
std::atomic<unsigned> g_distinct_devices{ 1 };

void parse_reads(BatchQueue& queue, RunReport& report, std::mutex& report_mutex, const ReadPlan& plan, bool raw_fastq, Backend* device_text,
                 bool further_levels)
{
    uint64_t       seq = 0;
    MateCopier     copier(queue, (unsigned)tun().size(Knob::mate_threads, 3));
    // slab parsers: half of the cores this process may use, between 4 and 12 (the other half: reader, mate copier, device
    // workers, post pool); 8 on the 16-core quota of the boxes the numbers in DESIGN.md come from
    // ... per GPU: a node's worth of GPUs gets a node's worth of readers, within half of the usable cores (round 5 stopped at 12 parsers / 6
    // raw-slab readers whatever the node: ~130 Mreads/s of plain FASTQ for ANY number of GPUs, profiles/r06_host_ceiling_48m.json)
    const unsigned n_dev       = std::max(1u, g_distinct_devices.load());
    const unsigned par_threads = (unsigned)tun().size(Knob::parse_threads, std::min(12u * n_dev, std::max(4u, usable_cores() / 2)));
    const unsigned raw_threads = std::min(par_threads, 6u * n_dev);
    const size_t   slab_bytes  = tun().size(Knob::slab_bytes, 48u << 20);
    const size_t   par_min     = tun().size(Knob::parallel_min, 32u << 20);
    for (auto const& [prefix, files] : plan)
    {
        for (auto const& pair : files)
        {
            const bool           paired = pair.paired();
            ReadBatch            rb;
            ByteBuf              bases2; // mates 2 of the current batch
            auto                 fresh = [&]() {
                if (queue.take_free(rb))
                {
                    rb.id_buf.clear();
                    rb.id_off.assign(1, 0);
                    rb.bases.clear();
                    rb.raw = false;
                    rb.raw_fasta = false;
                    rb.text.clear();
                    rb.text2.clear();
                    rb.seq_at2.clear();
                    rb.seq_len2.clear();
                    rb.rec_at.clear();
                    rb.seq_at.clear();
                    rb.seq_len.clear();
                    rb.ticket.reset();
                    rb.dev_text = nullptr;
                    rb.dev_bytes = 0;
                    rb.dev_hold.reset();
                    rb.dev_device2 = -1;
                    rb.dev_text2 = nullptr;
                    rb.dev_bytes2 = 0;
                    rb.dev_hold2.reset();
                    rb.dev_need_letters = rb.dev_letters = false;
                }
                else
                    rb = ReadBatch();
                rb.paired = paired;
                rb.prefix = prefix;
                rb.off1.assign(1, 0);
                if (paired)
                    rb.off2.assign(1, 0);
                else
                    rb.off2.clear();
            };
            auto flush = [&]() {
                if (rb.size() == 0)
                    return;
                {
                    std::lock_guard<std::mutex> lk(report_mutex);
                    report.count_input(prefix, rb.size()); // :1253,1272
                }
                finalize_batch(rb, bases2);
                rb.seq = seq++;
                copier.deliver(std::move(rb));
                fresh();
            };
            auto report_error = [&](const std::string& what) { // :1278-1283: report, keep what was read, go on with the next file
                std::cerr << "Error parsing file(s) [" << pair.mate1 << ", " << pair.mate2 << "]" << what << std::endl;
            };
            fresh();

            // ---- parallel slabs ----------------------------------------------------------------------------------
            uint64_t resume1 = 0, resume2 = 0; // where the sequential reader takes over (0 = from the start)
            bool     file_done = false, fallback = false;
            // ---- a gzip file inflated on the device: pieces of text that never leave it (single-end; devgzip.cpp) ---------------------
            if (raw_fastq && !paired && device_text)
            {
                if (auto src = device_text->open_gzip_text(pair.mate1, slab_bytes, tun().size(Knob::device_inflate_min, 1u << 20)))
                {
                    src->go(&g_devices_ready.run);
                    auto            tracker = std::make_shared<RawFileTracker>();
                    size_t          pieces  = 0;
                    DeviceTextPiece pc;
                    std::string     why;
                    while (!tracker->stopped() && src->next(pc, why)) // (a piece that is no records ends the device's part of the file: no point in inflating on)
                    {
                        rb.raw        = true;
                        rb.raw_fasta  = src->fasta();
                        rb.text.clear();
                        rb.dev_text   = pc.dev;
                        rb.dev_bytes  = pc.bytes;
                        rb.dev_device = pc.device;
                        rb.dev_hold   = std::move(pc.hold);
                        rb.dev_need_letters = further_levels;
                        rb.text_at    = pc.at;
                        rb.raw_keep   = 0;
                        rb.ticket.reset(new RawTicket{ tracker, pieces++ });
                        rb.seq = seq++;
                        copier.deliver(std::move(rb));
                        fresh();
                    }
                    if (tun().is_set(Knob::timing) || !why.empty())
                        std::cerr << "[host input] " << pair.mate1 << ": " << src->report() << (why.empty() ? std::string() : "; given up: " + why) << std::endl;
                    uint64_t at = 0;
                    if (!tracker->wait_all(pieces, at))
                    {
                        if (at == UINT64_MAX) // (the pipeline is going down)
                            file_done = true;
                        else
                        {
                            resume1  = at;
                            fallback = true;
                        }
                    }
                    else if (!why.empty()) // the device path gave the file up behind the pieces it delivered: zlib goes on from there
                    {
                        resume1  = src->delivered();
                        fallback = true;
                    }
                    else
                        file_done = true;
                    // (the source's end -- threads joined, gigabytes of device buffers freed -- takes some ten milliseconds the last batches
                    //  need not wait for: it waits for its own pieces' holders anyway)
                    g_cleanups.add(std::thread([s = std::shared_ptr<DeviceTextSource>(std::move(src))]() mutable { s.reset(); }));
                }
            }
            // ---- both gzip files of a pair inflated on the device: file 1 in pieces, file 2 cut where the same records end ---------------
            // (GanonClassify.cpp:1240-1252: ids from file 1, file 2 consumed with take(n_reads) -- pairs go by record number)
            if (raw_fastq && paired && device_text)
            {
                const size_t piece = std::max<size_t>(slab_bytes / 2, 1 << 16), dmin = tun().size(Knob::device_inflate_min, 1u << 20);
                auto         src1  = device_text->open_gzip_text(pair.mate1, piece, dmin);
                auto         src2  = src1 ? device_text->open_gzip_text(pair.mate2, piece, 0, true) : nullptr;
                if (src1 && src2)
                {
                    src1->go(&g_devices_ready.run);
                    src2->go(&g_devices_ready.run);
                    auto            tracker = std::make_shared<RawFileTracker>();
                    size_t          pieces  = 0;
                    DeviceTextPiece p1, p2;
                    std::string     why;
                    bool            stopped = false; // a piece of file 1 was taken but could not be paired: the sequential readers start there
                    uint64_t        stop1 = 0, stop2 = 0;
                    while (!tracker->stopped() && src1->next(p1, why))
                    {
                        std::string why2;
                        p2 = DeviceTextPiece();
                        if (!src2->next_lines(p1.lines, p2, why2) && !why2.empty())
                        {
                            why     = "mate file: " + why2;
                            stopped = true;
                            stop1   = p1.at;
                            stop2   = src2->delivered();
                            break;
                        }
                        if (p2.bytes == 0 && why2.empty() && p2.hold == nullptr) // file 2 has ended: the mates stay empty, the sequential reader's business
                        {
                            stopped = true;
                            stop1   = p1.at;
                            stop2   = UINT64_MAX;
                            break;
                        }
                        rb.raw        = true;
                        rb.raw_fasta  = src1->fasta();
                        rb.text.clear();
                        rb.text2.clear();
                        rb.dev_text   = p1.dev;
                        rb.dev_bytes  = p1.bytes;
                        rb.dev_device = p1.device;
                        rb.dev_hold   = std::move(p1.hold);
                        rb.dev_device2 = p2.device;
                        rb.dev_text2  = p2.dev;
                        rb.dev_bytes2 = p2.bytes;
                        rb.dev_hold2  = std::move(p2.hold);
                        rb.dev_need_letters = further_levels;
                        rb.text_at    = p1.at;
                        rb.text2_at   = p2.at;
                        rb.raw_keep   = 0;
                        rb.ticket.reset(new RawTicket{ tracker, pieces++ });
                        rb.seq = seq++;
                        copier.deliver(std::move(rb));
                        fresh();
                    }
                    p1 = DeviceTextPiece();
                    p2 = DeviceTextPiece();
                    if (tun().is_set(Knob::timing) || !why.empty())
                        std::cerr << "[host input] " << pair.mate1 << ": " << src1->report() << "; " << pair.mate2 << ": " << src2->report()
                                  << (why.empty() ? std::string() : "; given up: " + why) << std::endl;
                    uint64_t at = 0, at2 = 0;
                    if (!tracker->wait_all(pieces, at, nullptr, &at2))
                    {
                        if (at == UINT64_MAX) // (the pipeline is going down)
                            file_done = true;
                        else
                        {
                            resume1  = at;
                            resume2  = at2;
                            fallback = true;
                        }
                    }
                    else if (stopped)
                    {
                        resume1  = stop1;
                        resume2  = stop2;
                        fallback = true;
                    }
                    else if (!why.empty()) // file 1's device path gave up behind the pieces it delivered
                    {
                        resume1  = src1->delivered();
                        resume2  = src2->delivered();
                        fallback = true;
                    }
                    else
                        file_done = true; // (whatever file 2 holds beyond file 1's records is not input: take(n_reads))
                }
                if (src1)
                    g_cleanups.add(std::thread([s = std::shared_ptr<DeviceTextSource>(std::move(src1))]() mutable { s.reset(); }));
                if (src2)
                    g_cleanups.add(std::thread([s = std::shared_ptr<DeviceTextSource>(std::move(src2))]() mutable { s.reset(); }));
            }
            // ---- raw pieces: the backend finds the records (single-end, uncompressed four-line FASTQ) ------------------------
            if (raw_fastq && !paired && !file_done && !fallback)
            {
                if (auto pfr = ParallelFastq::open(pair.mate1, raw_threads, slab_bytes, par_min, false, true))
                {
                    auto                tracker = std::make_shared<RawFileTracker>();
                    size_t              pieces  = 0;
                    ParallelFastq::Slab a;
                    bool                gave_up = false; // the slab readers do not take the piece at gave_up_at (and nothing behind it)
                    uint64_t            gave_up_at = 0;
                    while (pfr->next(a))
                    {
                        if (!a.text.empty())
                        {
                            rb.raw = true;
                            rb.raw_fasta = pfr->fasta();
                            rb.text.swap(a.text); // (a recycled batch's buffer goes back to the slab readers)
                            rb.text_at  = a.text_at;
                            rb.raw_keep = 0;
                            rb.ticket.reset(new RawTicket{ tracker, pieces++ });
                            rb.seq = seq++;
                            copier.deliver(std::move(rb)); // (input is counted by the worker, once the records are known)
                            fresh();
                        }
                        else if (a.irregular) // (the slab readers do not take this piece at all)
                        {
                            gave_up    = true;
                            gave_up_at = a.resume_at;
                        }
                        pfr->recycle(std::move(a));
                        a = ParallelFastq::Slab();
                    }
                    // every piece is records from end to end: done.  Otherwise the sequential reader goes on at the first byte that is
                    // not (a wrapped or damaged record, a last line without its newline, ...) -- pieces behind it were dropped.
                    uint64_t at = 0;
                    size_t   stopped_by = 0;
                    if (!tracker->wait_all(pieces, at, &stopped_by))
                    {
                        if (at == UINT64_MAX) // (the pipeline is going down)
                            file_done = true;
                        else
                        {
                            resume1  = at;
                            fallback = true;
                        }
                    }
                    else if (gave_up)
                    {
                        resume1  = gave_up_at;
                        fallback = true;
                    }
                    else
                        file_done = true;
                }
            }
            // ---- raw pieces of a pair: both mate files travel as text ----------------------------------------------------------------
            // File 1 is cut into pieces as above (half as large: a batch holds two of them); a piece of n records -- its lines are
            // counted while it is read -- goes with the next 4 n lines (2 n for FASTA) of file 2, which a line index of that file
            // locates (GanonClassify.cpp:1240-1252: file 2 is consumed with take(n_reads)) and a helper thread reads.  The backend
            // finds the records of both texts; the batch is the pairs both hold, and the first piece that is not records from end
            // to end IN BOTH FILES stops the pair of files there: the sequential readers go on at those two bytes.
            // When: measured on 16 host cores, one GPU (profiles/r04_e2e_pair_text_*.json) the two ways are level at 32 M pairs -- 58-62 Mpairs/s,
            // both at what the link and the workers' lanes carry -- with a third of the user CPU time for the text (1.6 against 5.0 s; the
            // kernel's copies out of the page cache are what is left), but the text's pieces are twice the page-locked memory, which has to
            // be locked while the first batches run: below some ten million pairs the parsed way finishes first.  So: text from 4 GiB of
            // first mate file on ($GANON_HOST_PAIR_TEXT=1 / 0: always / never).
            struct stat st1;
            const bool  force     = tun().is_set(Knob::pair_text);
            const bool  pair_text = raw_fastq && paired && !file_done && !fallback
                                   && (force ? !tun().off(Knob::pair_text) : (::stat(pair.mate1.c_str(), &st1) == 0 && (uint64_t)st1.st_size >= (4ull << 30)));
            if (pair_text)
            {
                auto pfr = ParallelFastq::open(pair.mate1, raw_threads, std::max<size_t>(slab_bytes / 2, 1 << 16), par_min, false, true);
                std::shared_ptr<LineIndex> idx2(pfr ? LineIndex::open(pair.mate2, 3, 0).release() : nullptr);
                if (pfr && idx2)
                {
                    const uint64_t      lpr     = pfr->fasta() ? 2 : 4;
                    auto                tracker = std::make_shared<RawFileTracker>();
                    size_t              pieces  = 0;
                    uint64_t            records = 0; // of the pieces handed out so far
                    ParallelFastq::Slab a;
                    bool                gave_up = false;
                    uint64_t            gave_up_at = 0, gave_up_at2 = 0;
                    while (pfr->next(a))
                    {
                        if (a.text.empty() || a.text_lines < lpr)
                        {
                            // a piece the slab readers do not deliver as text (a record of gigabytes), or one without a whole record:
                            // the sequential readers take it, from its first byte and from the mate file's matching line
                            if (a.text.empty() && !a.irregular)
                                break; // (an empty last slab: the file ended with the piece before)
                            gave_up     = true;
                            gave_up_at  = a.text.empty() ? a.resume_at : a.text_at;
                            gave_up_at2 = idx2->line_begin(records * lpr);
                            break;
                        }
                        const uint64_t n  = a.text_lines / lpr;
                        const uint64_t b0 = idx2->line_begin(records * lpr);
                        uint64_t       b1 = b0 == LineIndex::kNoSuchLine ? b0 : idx2->line_begin((records + n) * lpr);
                        if (b0 == LineIndex::kNoSuchLine)
                        {
                            // file 2 has ended: from here on the mates are empty, which is the sequential reader's business
                            gave_up     = true;
                            gave_up_at  = a.text_at;
                            gave_up_at2 = UINT64_MAX;
                            break;
                        }
                        if (b1 == LineIndex::kNoSuchLine)
                            b1 = idx2->size(); // fewer mates than records: the batch ends where they do, the piece is not whole
                        rb.raw       = true;
                        rb.raw_fasta = pfr->fasta();
                        rb.text.swap(a.text);
                        rb.text_at  = a.text_at;
                        rb.text2_at = b0;
                        rb.raw_keep = 0;
                        rb.ticket.reset(new RawTicket{ tracker, pieces++ });
                        rb.seq = seq++;
                        records += n;
                        const size_t room = std::max<size_t>(slab_bytes / 2, 1 << 16) + slab_bytes / 32 + 65536;
                        copier.submit(std::move(rb), [idx2, b0, b1, room](ReadBatch& x) {
                            if (!idx2->read(b0, b1, x.text2, room))
                                x.text2.clear(); // (the file shrank under us: the batch comes out short, the piece is not whole)
                        });
                        fresh();
                        pfr->recycle(std::move(a));
                        a = ParallelFastq::Slab();
                    }
                    uint64_t at = 0, at2 = 0;
                    if (!tracker->wait_all(pieces, at, nullptr, &at2))
                    {
                        if (at == UINT64_MAX) // (the pipeline is going down)
                            file_done = true;
                        else
                        {
                            resume1  = at;
                            resume2  = at2;
                            fallback = true;
                        }
                    }
                    else if (gave_up)
                    {
                        resume1  = gave_up_at;
                        resume2  = gave_up_at2;
                        fallback = true;
                    }
                    else
                        file_done = true; // (whatever file 2 holds beyond file 1's records is not input: take(n_reads))
                }
            }
            if (!file_done && !fallback)
            {
                auto pf1 = ParallelFastq::open(pair.mate1, paired ? std::max(1u, par_threads / 2) : par_threads, slab_bytes, par_min, paired);
                std::shared_ptr<ParallelFastq> pf2(paired && pf1 ? ParallelFastq::open(pair.mate2, std::max(1u, par_threads / 2), slab_bytes, 0)
                                                                 : nullptr);
                if (paired && !pf2)
                    pf1.reset();
                // file 2's slabs are shared with the mate copier; whoever lets go of one last hands it back to its parser
                auto new_b = [&]() {
                    return pf2 ? std::shared_ptr<ParallelFastq::Slab>(new ParallelFastq::Slab(),
                                                                      [pf2](ParallelFastq::Slab* p) {
                                                                          pf2->recycle(std::move(*p));
                                                                          delete p;
                                                                      })
                               : std::make_shared<ParallelFastq::Slab>();
                };
                ParallelFastq::Slab                  a;
                std::shared_ptr<ParallelFastq::Slab> bcur = new_b();
                ParallelFastq::Slab*                 bp   = bcur.get();
                std::vector<MateCopier::Part>        parts; // mates of the current batch that the copier will fetch
                size_t              bpos = 0;       // next unread mate of slab b
                bool                b_open = true;  // file 2 may still deliver slabs
                bool                b_end = false;  // file 2 is exhausted (mates stay empty, like the sequential reader at EOF)
                std::string         b_error;        // file 2's ParseError, raised when the mate at bpos is asked for
                bool                b_irregular = false;
                // makes sure slab b has an unread mate; false when file 2 cannot deliver one (end / error / irregular)
                auto mate_ready = [&]() -> bool {
                    while (bpos >= bp->size())
                    {
                        if (!bp->error.empty())
                        {
                            b_error = bp->error;
                            return false;
                        }
                        if (bp->irregular)
                        {
                            b_irregular = true;
                            return false;
                        }
                        auto nb = new_b();
                        if (!b_open || !pf2->next(*nb))
                        {
                            b_open = false;
                            b_end  = true;
                            return false;
                        }
                        bcur = std::move(nb);
                        bp   = bcur.get();
                        bpos = 0;
                    }
                    return true;
                };
                while (pf1 && pf1->next(a))
                {
                    // records of slab a, cut into batches of at most batch_reads()
                    size_t r0 = 0;
                    bool   stop_file = false;
                    while (r0 < a.size() && !stop_file)
                    {
                        const size_t r1 = std::min(a.size(), r0 + batch_reads());
                        size_t       taken = r1 - r0;
                        if (r0 == 0 && r1 == a.size())
                        {
                            rb.id_buf.swap(a.ids); // the slab is the batch
                            rb.id_off.swap(a.id_off);
                            rb.bases.swap(a.bases);
                            rb.off1.swap(a.off);
                        }
                        else
                        {
                            rb.id_buf.assign(a.ids, a.id_off[r0], a.id_off[r1] - a.id_off[r0]);
                            rb.id_off.resize(taken + 1);
                            rb.off1.resize(taken + 1);
                            for (size_t i = 0; i <= taken; ++i)
                            {
                                rb.id_off[i] = a.id_off[r0 + i] - a.id_off[r0];
                                rb.off1[i]   = a.off[r0 + i] - a.off[r0];
                            }
                            rb.bases.assign(a.bases.begin() + a.off[r0], a.bases.begin() + a.off[r1]);
                        }
                        if (paired)
                        {
                            rb.off2.assign(1, 0);
                            rb.off2.reserve(taken + 1);
                            size_t i = 0;
                            // mates of a run of records that file 2's current slab holds: one copy, offsets rebased in a
                            // loop of additions (the record-by-record code below takes over where file 2 ends, fails or
                            // turns irregular)
                            parts.clear();
                            while (i < taken && !b_end && (bpos < bp->size() || mate_ready()))
                            {
                                const size_t run = std::min(taken - i, bp->size() - bpos);
                                parts.push_back(MateCopier::Part{ bcur, bpos, run });
                                bpos += run;
                                i += run;
                            }
                            if (i == taken && !parts.empty())
                            {
                                // the usual case: every mate is there.  The copier threads do the copying.
                                {
                                    std::lock_guard<std::mutex> lk(report_mutex);
                                    report.count_input(prefix, rb.size()); // :1253,1272
                                }
                                rb.seq = seq++;
                                copier.submit(std::move(rb), std::move(parts));
                                parts.clear();
                                fresh();
                                r0 += taken;
                                continue;
                            }
                            MateCopier::materialise(parts, bases2, rb.off2); // file 2 ended / failed: finish here, record by record
                            parts.clear();
                            for (; i < taken; ++i)
                            {
                                if (b_end || mate_ready())
                                {
                                    if (!b_end)
                                    {
                                        bases2.insert(bases2.end(), bp->bases.begin() + bp->off[bpos], bp->bases.begin() + bp->off[bpos + 1]);
                                        ++bpos;
                                    }
                                    rb.off2.push_back(bases2.size());
                                    continue;
                                }
                                if (b_end) // (set by mate_ready just now: this and all later mates are empty)
                                {
                                    rb.off2.push_back(bases2.size());
                                    continue;
                                }
                                // file 2 stopped at this mate: keep the records before it
                                if (!b_error.empty())
                                    rb.off2.push_back(bases2.size()); // the sequential reader keeps this read with an empty mate
                                const size_t keep = rb.off2.size() - 1;
                                rb.id_buf.resize(rb.id_off[keep]);
                                rb.id_off.resize(keep + 1);
                                rb.bases.resize(rb.off1[keep]);
                                rb.off1.resize(keep + 1);
                                if (b_irregular)
                                {
                                    resume1  = a.rec_at[r0 + keep];
                                    resume2  = bp->resume_at;
                                    fallback = true;
                                }
                                taken     = keep;
                                stop_file = true;
                                break;
                            }
                        }
                        flush();
                        r0 += taken;
                    }
                    if (stop_file)
                    {
                        if (!b_error.empty())
                        {
                            report_error(b_error);
                            file_done = true;
                        }
                        break;
                    }
                    if (!a.error.empty())
                    {
                        report_error(a.error);
                        file_done = true;
                        break;
                    }
                    if (a.irregular)
                    {
                        resume1  = a.resume_at;
                        fallback = true;
                        if (paired)
                        {
                            // the mate of the first unparsed record of file 1: the next unread record of file 2
                            if (bpos < bp->size() || mate_ready())
                                resume2 = bp->rec_at[bpos];
                            else if (b_irregular)
                                resume2 = bp->resume_at;
                            else if (!b_error.empty())
                            {
                                report_error(b_error); // (file 2 fails before file 1 continues)
                                file_done = true;
                            }
                            else
                                resume2 = UINT64_MAX; // file 2 is exhausted
                        }
                        break;
                    }
                    pf1->recycle(std::move(a));
                    a = ParallelFastq::Slab();
                }
                if (pf1 && !fallback)
                    file_done = true; // the slabs covered the whole file (or ended it with a parse error)
            }
            if (file_done)
                continue;

            // ---- sequential reader (whole file, or the rest of it) -------------------------------------------------
            try
            {
                SeqReader                   fin1(pair.mate1, resume1);
                std::unique_ptr<MateStream> fin2;
                if (paired && resume2 != UINT64_MAX)
                    fin2.reset(new MateStream(pair.mate2, resume2));
                while (fin1.next(rb.id_buf, rb.bases))
                {
                    rb.id_off.push_back(rb.id_buf.size());
                    rb.off1.push_back(rb.bases.size());
                    if (paired)
                    {
                        try
                        {
                            if (fin2)
                                fin2->next(bases2); // at EOF the mate stays empty
                        }
                        catch (ParseError const&)
                        {
                            rb.off2.push_back(bases2.size());
                            throw;
                        }
                        rb.off2.push_back(bases2.size());
                    }
                    if (rb.size() >= batch_reads() || rb.bases.size() + bases2.size() >= kBatchBases)
                        flush();
                }
                flush();
            }
            catch (ParseError const& ext)
            {
                flush();
                report_error(ext.what());
                continue;
            }
        }
    }
    copier.drain();
    queue.done();
    g_cpu.reader.add_this_thread();
}

} // namespace gnhost
