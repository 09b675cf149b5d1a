// backend.hpp -- the seam between the host pipeline and the device hot path.
//
// Loading: a filter file is parsed into its metadata (FilterMeta) and its bit matrices are STREAMED, chunk by chunk,
// into a FilterSink -- for the product that is HBM (gn_filter_write_rows from pinned staging, backend_hip.cpp), on
// every selected GPU at once; the host never holds a whole matrix (the reference deserialises the complete sdsl
// bit_vector into RAM first, GanonClassify.cpp:949-986).
// Classifying: the host hands a batch of reads to a Backend and gets, per filter, the sparse result of select_matches
// (GanonClassify.cpp:504-577): for every read the targets whose summed, capped count reached the read's cutoff.
// The product binary links exactly one implementation, backend_hip.cpp (libganon_hip.so through include/ganon_hip.h);
// there is no CPU implementation in ganon_amd/.  (tests/host_oracle/ links a checker backend built on the CPU oracle
// to exercise the host logic without a GPU.)
#pragma once

#include "filter_io.hpp"
#include "hostmem.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <vector>

namespace gnhost
{

// Uncompressed FASTQ whose records the BACKEND finds (raw batches, below): the pieces of one file are validated in file
// order although several workers hold them at once.  A piece whose text is not four-line records from its first to its last
// byte stops the file there: pieces behind it are void, and the reader hands the rest of the file to its sequential parser.
class RawFileTracker
{
public:
    // piece `idx` was tokenised: all of it is records (complete), or the first byte that is not part of one is resume_at
    void publish(size_t idx, bool complete, uint64_t resume_at, uint64_t resume_at2 = 0)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (state_.size() <= idx)
            state_.resize(idx + 1, 0);
        if (state_[idx])
            return;
        state_[idx] = complete ? 1 : 2;
        ++published_;
        while (known_below_ < state_.size() && state_[known_below_])
            ++known_below_;
        if (!complete && idx < first_stop_)
        {
            first_stop_ = idx;
            resume_at_  = resume_at;
            resume_at2_ = resume_at2;
        }
        cv_.notify_all();
    }
    // waits until every piece before idx is known; true when all of them were complete (this piece counts)
    bool wait_prefix(size_t idx)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return first_stop_ < idx || known_below_ >= idx; });
        return first_stop_ >= idx;
    }
    // has a piece stopped the file already?  (the reader need not cut further pieces then)
    bool stopped()
    {
        std::lock_guard<std::mutex> lk(m_);
        return first_stop_ != SIZE_MAX;
    }
    // the reader, after the file's last piece: waits for all `count` pieces; false when one stopped the file (then resume_at)
    // (resume_at2: the same place in the mate file of a pair whose pieces travel as text)
    bool wait_all(size_t count, uint64_t& resume_at, size_t* stopped_by = nullptr, uint64_t* resume_at2 = nullptr)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return published_ >= count; });
        resume_at = resume_at_;
        if (resume_at2)
            *resume_at2 = resume_at2_;
        if (stopped_by)
            *stopped_by = first_stop_;
        return first_stop_ == SIZE_MAX;
    }

private:
    std::mutex              m_;
    std::condition_variable cv_;
    std::vector<uint8_t>    state_;
    size_t                  published_ = 0, first_stop_ = SIZE_MAX, known_below_ = 0; // (every piece below known_below_ has been published)
    uint64_t                resume_at_ = 0, resume_at2_ = 0;
};

// a raw batch's place in its file; a batch that is dropped before anyone tokenised it stops the file (nobody waits for ever)
struct RawTicket
{
    std::shared_ptr<RawFileTracker> tracker;
    size_t                          idx  = 0;
    bool                            done = false;
    void publish(bool complete, uint64_t resume_at, uint64_t resume_at2 = 0)
    {
        if (tracker && !done)
            tracker->publish(idx, complete, resume_at, resume_at2);
        done = true;
    }
    ~RawTicket() { publish(false, UINT64_MAX); }
};

// A batch of reads in the layout the C ABI takes: ASCII bases, mate-1 block then mate-2 block.
struct ReadBatch
{
    bool                  paired = false;
    uint64_t              seq    = 0;  // position in the input stream (outputs are written in this order)
    std::string           prefix;
    std::string           id_buf;      // all read ids back to back
    std::vector<uint64_t> id_off{ 0 }; // n+1 offsets into id_buf
    ByteBuf               bases;       // mates 1 of all reads, then mates 2 of all reads (page-locked under the HIP backend)
    std::vector<uint64_t> off1;        // n+1
    std::vector<uint64_t> off2;        // n+1 when paired (offsets into `bases`)
    // Raw form (single-end uncompressed FASTQ under a backend that tokenises, Backend::tokenises_fastq): `text` is a piece of
    // the file that begins with a record; Backend::tokenise finds the records, Backend::classify fills rec_at / seq_at / seq_len
    // for the `raw_keep` records the batch consists of.  Ids and letters are read where they lie in `text`.
    bool                       raw = false;
    bool                       raw_fasta = false; // the text is two-line FASTA (>id / letters), not four-line FASTQ
    ByteBuf                    text;
    uint64_t                   text_at = 0;  // offset of text[0] in the file
    uint32_t                   raw_keep = 0; // records of the batch (set by the pipeline between tokenise and classify)
    U32Buf                     rec_at, seq_at, seq_len;
    // ... of a pair: the piece of the mate file that holds the same records by number (cut by the reader at the line count of `text`);
    // mate i is record i of text2, its letters are read where they lie
    ByteBuf                    text2;
    uint64_t                   text2_at = 0;
    uint64_t                   raw_parsed2 = 0; // set by tokenise: bytes of text2 the batch's mates cover
    U32Buf                     seq_at2, seq_len2;
    // ... or the text lies in device memory (a piece of a gzip file inflated there: DeviceTextSource): `text` is empty until the batch is
    // classified, then holds the header lines of its records back to back (rec_at / seq_at are rewritten to fit: ids are read as ever,
    // letters are not on the host at all -- such batches only exist on runs with one hierarchy level, where nothing needs them)
    const uint8_t*        dev_text   = nullptr;
    uint64_t              dev_bytes  = 0;
    int                   dev_device = -1;
    int                   dev_device2 = -1; // ... of dev_text2 (-1: where dev_text lies)
    std::shared_ptr<void> dev_hold;  // keeps the device buffer alive; dropped once the text is copied into the worker's stream
    const uint8_t*        dev_text2  = nullptr; // ... of a pair: the mate file's piece (the same records by number), same device
    uint64_t              dev_bytes2 = 0;
    std::shared_ptr<void> dev_hold2;
    bool                  dev_need_letters = false; // the run has further hierarchy levels: classify() also brings the letters (bases, off1, off2)
    bool                  dev_letters = false;      // ... and they are there
    uint64_t raw_bytes() const { return dev_text ? dev_bytes : text.size(); }
    std::unique_ptr<RawTicket> ticket;
    size_t size() const { return raw ? rec_at.size() : id_off.size() - 1; }
    std::string_view id(size_t i) const
    {
        if (!raw)
            return { id_buf.data() + id_off[i], size_t(id_off[i + 1] - id_off[i]) };
        const char* b = reinterpret_cast<const char*>(text.data()) + rec_at[i] + 1; // behind the '@'
        size_t      n = seq_at[i] - rec_at[i] - 2;                                   // up to the '\n' before the letters
        if (n && b[n - 1] == '\r')
            --n;
        return { b, n };
    }
    uint64_t       len1(size_t i) const { return raw ? seq_len[i] : off1[i + 1] - off1[i]; }
    uint64_t       len2(size_t i) const { return !paired ? 0 : raw ? seq_len2[i] : off2[i + 1] - off2[i]; }
    const uint8_t* seq1(size_t i) const { return raw && !dev_letters ? text.data() + seq_at[i] : bases.data() + off1[i]; }
    const uint8_t* seq2(size_t i) const { return raw && !dev_letters ? text2.data() + seq_at2[i] : bases.data() + off2[i]; }
};

struct Match
{
    uint32_t read, target, count; // target = index into FilterMeta::targets
};

struct FilterResult
{
    U64Buf                match_off; // n+1 (the per-read arrays are written by the device: page-locked under the HIP backend)
    std::vector<Match, ArenaAllocator<Match>> matches; // grouped by read, ascending target
    std::vector<uint8_t>  fpr_ok;    // empty, or per match: 1 = the backend already verified q <= fpr_query (see set_postfilter)
    // ... or that flag is bit 31 of Match::count (kMatchFprOk), as the device writes it: the records are used as they arrive
    bool                  flag_in_count = false;
    static constexpr uint32_t kMatchFprOk = 0x80000000u;
};

struct BatchResult
{
    U32Buf                    n_hashes; // per read
    ByteBuf                   status;   // 0 ok, 1 small, 2 big (GN_READ_*)
    std::vector<FilterResult> per_filter;
    // set when the backend already applied the pre-pass of filter_matches (see Backend::set_postfilter): the matches are
    // the survivors, max_count is every read's largest match count BEFORE filtering, the two totals are what was dropped
    bool                  prefiltered = false;
    U32Buf                max_count;
    uint64_t              dropped_rel_filter = 0, dropped_fpr_query = 0;
};

// filter_matches parameters of a hierarchy level (GanonClassify.cpp:579-613,755-761)
struct PostFilterSpec
{
    double                           rel_filter = 0.0;
    double                           fpr_query  = 1.0;
    std::vector<std::vector<double>> target_fpr; // per filter of the level: per FilterMeta target of that filter
    // several filters: no target name occurs in two of them (then the level's merge of matches is a plain union and the
    // rules can be applied per filter with the level's max/min)
    bool disjoint_targets = true;
    // per filter: target -> id of its name in the level (the same name in two filters has the same id).  Needed when the
    // targets are not disjoint: the backend then replays the level's merge (a target keeps its largest count, the earliest
    // filter's on ties, GanonClassify.cpp:531-537) and hands over the winners only.  A read it cannot do that for comes
    // back with all its matches and bit 31 of BatchResult::max_count set: the host runs merge and rules on it.
    std::vector<std::vector<uint32_t>> target_gid;
};

// Text of a compressed input file that is produced in device memory (backend_hip: a gzip file inflated by csrc/gn_inflate.hip), piece by
// piece; every piece begins with a record and ends behind one (the file's last piece ends where the stream does).
struct DeviceTextPiece
{
    const uint8_t*        dev = nullptr;
    uint64_t              bytes = 0;
    uint64_t              at = 0; // offset of the piece's first byte in the decompressed stream
    uint64_t              lines = 0; // lines (newlines) of the piece; ~0: the file's last bytes, not a whole record
    int                   device = -1;
    std::shared_ptr<void> hold;
};
class DeviceTextSource
{
public:
    virtual ~DeviceTextSource() = default;
    // the next piece; false at the end of the stream (err empty) or when the device path gives the file up (err says why): a host
    // reader then continues at decompressed offset delivered()
    virtual bool        next(DeviceTextPiece& out, std::string& err) = 0;
    // a source opened by_lines (the mate file of a pair): the next `lines` lines as one piece -- fewer (out.lines says) where the stream
    // ends; ~0: what is left of the CURRENT step's text (not of the stream: while further steps exist the source stays open and a later
    // call goes on behind it)
    virtual bool        next_lines(uint64_t /*lines*/, DeviceTextPiece& /*out*/, std::string& err)
    {
        err = "not a source by lines";
        return false;
    }
    // Opening a source allocates and feeds; nothing is decoded before go(), and no new step is begun while *run is false (the caller
    // says when the device may take that work on: device allocations of others crawl beside it)
    virtual void        go(const std::atomic<bool>* /*run*/ = nullptr) {}
    virtual uint64_t    delivered() const = 0;
    virtual bool        fasta() const = 0;
    virtual std::string report() const { return std::string(); }
};

// devgzip.cpp (binaries linked with libganon_hip.so only): nullptr when the file is not one for the device inflater
// `devices`: where the file's steps are inflated in turn (gn_inflate_set_turns): the text of step k lies on devices[k mod n], next to the
// worker that classifies it; a source opened by_lines (a mate file) stays on devices[0]
std::unique_ptr<DeviceTextSource> open_device_gzip(const std::string& path, const std::vector<int>& devices, size_t piece_bytes, size_t min_bytes, bool by_lines);

// One device (or the test checker): receives filters, classifies batches.  Not thread-safe; one host thread each.
class Backend : public FilterSink
{
public:
    virtual void clear_filters() = 0;
    // classify `batch` against every loaded filter with minimiser shape (k, w); rel_cutoff[i] belongs to filter i
    // (a raw batch -- ReadBatch::raw -- has been through tokenise(); its first raw_keep records are classified and described)
    virtual bool classify(ReadBatch& batch, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out,
                          std::string& err) = 0;
    // Optional.  Does this backend find the records of uncompressed four-line FASTQ itself?  Then the reader hands such files
    // over as raw batches (pieces of the file in page-locked memory) instead of parsing them.
    virtual bool tokenises_fastq() const { return false; }
    // Optional.  A gzip-compressed FASTQ / FASTA file as text pieces in this backend's device memory; nullptr: not a file for that (not
    // gzip, blocked gzip, too small, no room) or not a backend that does it.  piece_bytes = text per piece, about.
    // Optional.  Free memory of the device open_gzip_text would use, now (0: not a backend with device memory).
    virtual uint64_t free_device_bytes() const { return 0; }
    // one_device: the text must lie on this backend's device only (the first file of a pair: its pieces travel with the mate file's)
    virtual std::unique_ptr<DeviceTextSource> open_gzip_text(const std::string& /*path*/, size_t /*piece_bytes*/, size_t /*min_bytes*/, bool /*by_lines*/ = false,
                                                             bool /*one_device*/ = false)
    {
        return nullptr;
    }
    // Raw batches: takes batch.text, finds the records.  n_reads = records before the first that is not a plain four-line
    // record (or the end of the text inside one); parsed_bytes = where that one begins (== text.size(): all of it is records).
    // A paired raw batch (batch.paired, batch.text2 = the mate file's piece with the same records by number): the pairs both
    // texts hold before either's first non-record; parsed_bytes is text's, batch.raw_parsed2 is set to text2's.
    virtual bool tokenise(ReadBatch& /*batch*/, uint32_t& /*n_reads*/, uint64_t& /*parsed_bytes*/, std::string& err)
    {
        err = "this backend does not tokenise";
        return false;
    }
    virtual std::string describe() const = 0;
    // Optional.  The two calls above in halves, for a caller that keeps two backends busy from one thread (it starts a batch on one
    // while the other one's kernels run): *_begin queues the work and returns, *_end waits for it and delivers.  The defaults do
    // everything in *_end.
    virtual bool tokenise_begin(ReadBatch& /*batch*/, std::string& /*err*/) { return true; }
    virtual bool tokenise_end(ReadBatch& batch, uint32_t& n_reads, uint64_t& parsed_bytes, std::string& err)
    {
        return tokenise(batch, n_reads, parsed_bytes, err);
    }
    virtual bool classify_begin(ReadBatch& /*batch*/, uint32_t /*k*/, uint32_t /*w*/, const std::vector<double>& /*rel_cutoff*/, std::string& /*err*/)
    {
        return true;
    }
    virtual bool classify_end(ReadBatch& batch, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out, std::string& err)
    {
        return classify(batch, k, w, rel_cutoff, out, err);
    }
    // Optional.  Ask for the --rel-filter rule (exactly) and the --fpr-query rule (conservatively: only matches that are
    // above the limit by a safe margin) to be applied where the matches are produced, so that only survivors travel to the
    // host, which then applies the exact --fpr-query rule to them (except to those the backend marks as surely passing,
    // FilterResult::fpr_ok).  nullptr switches it off.  Returns whether the backend
    // will do it for the filters it currently holds (filters that see whole reads and report disjoint targets);
    // BatchResult::prefiltered says so per batch.  The default does nothing: the host then runs filter_matches on everything.
    virtual bool set_postfilter(const PostFilterSpec* /*spec*/) { return false; }
    // Optional.  Reads with more than 65535 minimisers: classified like every other read (the reference's -DLONGREADS build)
    // instead of coming back with status 2.  Returns whether the backend can do that.
    virtual bool set_long_reads(bool /*on*/) { return false; }
    // Optional.  After a level's filters are loaded: does this worker take batches on this level?  (A level with a filter
    // that is partitioned over the devices keeps all of them busy with every batch: only a few workers run then.)
    virtual bool active() const { return true; }
    // Optional.  Called by a worker before its first batch of a level: batches will hold at most this many reads / bases.
    // (Device streams are created here, while the reader is still parsing its first slabs, instead of with the first batch.)
    virtual void prepare(size_t /*max_reads*/, size_t /*max_bases*/) {}
    // Optional.  After prepare(): run something small through the whole path so that one-time costs (code loading, first-use
    // allocations) are paid before the first real batch.  Results are discarded.
    virtual void warm_up(uint32_t /*k*/, uint32_t /*w*/, const std::vector<double>& /*rel_cutoff*/) {}
    // Optional.  A second worker context on the same device and the same filters (own streams, nothing loaded twice), for a
    // caller that keeps two batches in flight from one thread.  It takes clear_filters / set_postfilter / set_long_reads /
    // prepare / tokenise / classify like the backend it comes from, never filter data.  nullptr: there is none.
    virtual std::unique_ptr<Backend> twin() { return nullptr; }
    // Optional.  Where the level's filters were put (replicated / partitioned, which columns on which device), for --verbose.
    virtual std::string placement() const { return std::string(); }
    // Optional.  What travelled between devices so far (a bin-range partitioned filter's sparse matches on their way to the batch's
    // home device) and how: one line per device pair, for --verbose.
    virtual std::string exchange_report() const { return std::string(); }
};

// devices: indices, or empty = every visible device ("all").  backend_hip.cpp (or the test checker).
std::vector<std::unique_ptr<Backend>> make_backends(const std::vector<int>& devices, std::string& err);

// Forwards one filter stream to several backends (the filter is replicated into every GPU's HBM); the staging
// buffers of the first backend are shared (pinned host memory is visible to every device).
class ReplicatingSink final : public FilterSink
{
public:
    explicit ReplicatingSink(std::vector<std::unique_ptr<Backend>>& backends) : b_(backends) {}
    bool begin(const FilterMeta& meta, std::string& err) override
    {
        for (auto& b : b_)
            if (!b->begin(meta, err))
                return false;
        return true;
    }
    uint64_t* staging(int which, size_t bytes) override { return b_.front()->staging(which, bytes); }
    bool      rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err) override
    {
        for (auto& b : b_)
            if (!b->rows(ibf, row_begin, n_rows, src, err))
                return false;
        return true;
    }
    bool drain(std::string& err) override
    {
        for (auto& b : b_)
            if (!b->drain(err))
                return false;
        return true;
    }
    bool end(std::string& err) override
    {
        for (auto& b : b_)
            if (!b->end(err))
                return false;
        return true;
    }

private:
    std::vector<std::unique_ptr<Backend>>& b_;
};

} // namespace gnhost
