// backend_hip.cpp -- the only Backend of the product: libganon_hip.so through the C ABI (include/ganon_hip.h), one
// instance per GPU.  Fails loudly when no MI355X/HIP device is usable; there is no CPU fallback.
#include "backend.hpp"
#include "pinned_pool.hpp"
#include "placement.hpp"
#include "tunables.hpp"

#include <ganon_hip.h>

#include <algorithm>
#include <chrono>
#include <iostream>
#include <atomic>
#include <map>
#include <memory>
#include <vector>
#include <unordered_map>
#include <thread>
#include <mutex>
#include <cstdlib>
#include <sstream>

namespace gnhost
{

namespace
{

// One worker: a host thread's streams on the filters of the DeviceSet.  Replicated filters are classified on the worker's
// home device; the parts of a partitioned filter on whichever devices hold them, all against the same batch, and put back
// together on the home device (gn_gather).
class HipBackend final : public Backend
{
public:
    HipBackend(std::shared_ptr<DeviceSet> set, size_t index) : set_(std::move(set)), index_(index), device_(set_->entries()[index]) {}
    ~HipBackend() override
    {
        PinnedPool::get().settle();
        if (tun().is_set(Knob::timing) && index_ == 0 && !twin_)
            std::cerr << "[pinned pool] " << PinnedPool::get().tally() << std::endl;
        if (tun().is_set(Knob::timing) && n_create_)
            std::cerr << "[backend timing] device " << device_ << ": stream (re)creation " << sec_create_ << " s (" << n_create_
                      << "x), submit calls " << sec_submit_ << " s, waiting for + fetching the results " << sec_fetch_ << " s, FASTQ text: upload calls "
                      << sec_tok_enqueue_ << " s + waiting for the records " << sec_tok_wait_ << " s"
                      << (gathered_bytes_ ? ", moved between devices " + std::to_string(gathered_bytes_ >> 20) + " MiB" : std::string()) << std::endl;
        clear_filters();
    }

    // ---- FilterSink: the first worker hands the filter to the DeviceSet; the others find it there ---------------
    bool begin(const FilterMeta& f, std::string& err) override { return index_ ? true : set_->begin(f, err); }
    uint64_t* staging(int which, size_t bytes) override { return set_->staging(which, bytes); }
    bool rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err) override
    {
        return index_ ? true : set_->rows(ibf, row_begin, n_rows, src, err);
    }
    bool drain(std::string& err) override { return index_ ? true : set_->drain(err); }
    bool end(std::string& err) override { return index_ ? true : set_->end(err); }

    void clear_filters() override
    {
        drop_streams();
        if (index_ == 0 && !twin_)
            set_->clear();
    }

    std::unique_ptr<Backend> twin() override
    {
        auto t   = std::make_unique<HipBackend>(set_, index_);
        t->twin_ = true;
        return t;
    }

    // A level with a partitioned filter keeps every device busy with every batch: a few workers (upload and fetch of one
    // batch behind the kernels of another) are all it can use.
    bool active() const override
    {
        if (!set_->any_spread())
            return true;
        const size_t k = std::max<size_t>(1, tun().size(Knob::partition_workers, 2));
        return index_ < k;
    }

    std::string placement() const override { return index_ ? std::string() : set_->placement(); }
    std::string exchange_report() const override
    {
        if (index_ || twin_)
            return std::string();
        int n = 0;
        if (gn_device_count(&n) != GN_OK)
            return std::string();
        std::string out;
        for (int dst = 0; dst < n; ++dst)
            for (int src = 0; src < n; ++src)
            {
                int      state = 0;
                uint64_t bytes = 0;
                if (gn_peer_stats(dst, src, &state, &bytes) != GN_OK || (state == 0 && bytes == 0))
                    continue;
                out += "device " + std::to_string(src) + " -> " + std::to_string(dst) + ": "
                       + (dst == src ? "one device (copy path forced by the library switch gather_copy)"
                                     : state == 1 ? "peer access enabled (direct device-to-device copies)" : "no peer access (copies staged by the runtime)")
                       + ", "
                       + std::to_string(bytes >> 20) + " MiB of matches gathered\n";
            }
        return out;
    }

    // Uncompressed FASTQ as text, records found on the device (csrc/gn_fastq.hip): it takes the parse -- the largest share of the
    // host's CPU seconds -- off the host and costs 2.1x the bytes over the link.  Measured on one MI355X behind a 16-core host
    // (DESIGN 7-5): the link-bound device side alone delivers 151-157 Mreads/s, the whole binary 125-130 against 109-116 with the
    // host's slab parser on its default 8 threads (134 on 12), for 5 instead of 7.5 CPU seconds per 64 M reads.  Several GPUs have
    // a link each but share the host's cores: there it matters more.  $GANON_HOST_DEVICE_FASTQ=0 keeps the parse on the host.
    bool tokenises_fastq() const override
    {
        return !tun().off(Knob::device_fastq);
    }

    // Raw batch: the text goes to the first stream of every device the level's filters live on (the stream that takes a parsed
    // batch's upload, see classify()); the records are found there (csrc/gn_fastq.hip).  Every device finds the same ones.
    bool tokenise(ReadBatch& b, uint32_t& n_reads, uint64_t& parsed_bytes, std::string& err) override
    {
        return tokenise_begin(b, err) && tokenise_end(b, n_reads, parsed_bytes, err);
    }

    uint64_t free_device_bytes() const override
    {
        uint64_t fr = 0, tot = 0;
        return gn_device_memory(device_, &fr, &tot) == GN_OK ? fr : 0;
    }

    std::unique_ptr<DeviceTextSource> open_gzip_text(const std::string& path, size_t piece_bytes, size_t min_bytes, bool by_lines, bool one_device) override
    {
        if (one_device)
            return open_device_gzip(path, std::vector<int>{ device_ }, piece_bytes, min_bytes, by_lines);
        // every distinct device of the run takes its share of the file's steps, this worker's own device first
        std::vector<int> devs{ device_ };
        for (int d : set_->unique_devices())
            if (d != device_)
                devs.push_back(d);
        return open_device_gzip(path, devs, piece_bytes, min_bytes, by_lines);
    }

    bool tokenise_begin(ReadBatch& b, std::string& err) override
    {
        if (!resolve(err))
            return false;
        auto t = std::chrono::steady_clock::now();
        const bool     pair  = b.paired && b.raw;
        const uint64_t nb    = pair ? ((b.raw_bytes() + 15) & ~15ull) + (b.dev_text ? b.dev_bytes2 : b.text2.size()) : b.raw_bytes();
        const uint64_t reads = std::max<uint64_t>(hint_reads_, b.raw_bytes() / 40); // (records shorter than 40 bytes on average: the rest goes the slow way)
        // (prepare() sized the streams by the same rule: no re-creation unless a piece is larger than the reader said)
        std::vector<gn_stream*>& sources = tok_sources_;
        sources.clear();
        std::map<int, bool>     seen;
        for (auto& lf : filters_)
            for (auto& part : lf.parts)
            {
                if (seen.count(part.dp->device))
                    continue;
                seen[part.dp->device] = true;
                if (!part.s || part.stream_reads < reads || part.stream_bases < nb + 64)
                {
                    if (!make_stream(part, reads, std::max<uint64_t>(nb + nb / 16 + 65536, hint_bases_), err))
                        return false;
                    const auto now = std::chrono::steady_clock::now();
                    sec_create_ += std::chrono::duration<double>(now - t).count();
                    t = now;
                }
                const int fmt = b.raw_fasta ? GN_TEXT_FASTA : GN_TEXT_FASTQ;
                if ((b.dev_text && pair ? gn_stream_upload_text_pair_devices(part.s, b.dev_text, b.dev_bytes, b.dev_device, b.dev_text2, b.dev_bytes2,
                                                                           b.dev_device2 >= 0 ? b.dev_device2 : b.dev_device, fmt)
                     : b.dev_text      ? gn_stream_upload_text_device(part.s, b.dev_text, nb, fmt, b.dev_device)
                     : pair     ? gn_stream_upload_text_pair(part.s, b.text.data(), b.text.size(), b.text2.data(), b.text2.size(), fmt)
                                : gn_stream_upload_text(part.s, b.text.data(), nb, fmt))
                    != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                sources.push_back(part.s);
            }
        sec_tok_enqueue_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
        return !sources.empty();
    }

    bool tokenise_end(ReadBatch& b, uint32_t& n_reads, uint64_t& parsed_bytes, std::string& err) override
    {
        auto                     t       = std::chrono::steady_clock::now();
        std::vector<gn_stream*>& sources = tok_sources_;
        const bool               pair    = b.paired && b.raw;
        for (size_t i = 0; i < sources.size(); ++i)
        {
            uint32_t n  = 0;
            uint64_t pb = 0, pb2 = 0;
            if ((pair ? gn_stream_text_pair_index(sources[i], &n, &pb, &pb2) : gn_stream_fastq_index(sources[i], &n, nullptr, &pb)) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
            if (i == 0)
            {
                n_reads       = n;
                parsed_bytes  = pb;
                b.raw_parsed2 = pb2;
            }
            else if (n != n_reads || pb != parsed_bytes || pb2 != b.raw_parsed2)
            {
                err = "devices disagree about the records of a FASTQ piece";
                return false;
            }
        }
        b.dev_hold.reset(); // (the text is in every stream now: the inflater may write that buffer again)
        b.dev_hold2.reset();
        sec_tok_wait_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
        return !sources.empty();
    }

    bool classify(ReadBatch& b, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out,
                  std::string& err) override
    {
        return classify_begin(b, k, w, rel_cutoff, err) && classify_end(b, k, w, rel_cutoff, out, err);
    }

    // queues upload (a parsed batch), kernels and the pre-pass of every filter
    bool classify_begin(ReadBatch& b, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, std::string& err) override
    {
        if (!resolve(err))
            return false;
        if (warmed_)
            PinnedPool::get().mark_running();
        const uint32_t n = b.raw ? b.raw_keep : (uint32_t)b.size();
        if (b.raw && n == 0)
            return true;
        auto           t = std::chrono::steady_clock::now();
        auto           lap = [&](double& acc) {
            const auto now = std::chrono::steady_clock::now();
            acc += std::chrono::duration<double>(now - t).count();
            t = now;
        };

        // submit to every stream first (asynchronous), then fetch
        const uint64_t nb = std::max<uint64_t>(b.bases.size(), 1);
        std::map<int, gn_stream*> source_of; // device -> the stream that holds this batch there
        const bool                share_hashes = !tun().is_set(Knob::no_shared_hashes);
        for (size_t i = 0; i < filters_.size(); ++i)
            for (size_t g = 0; g < filters_[i].parts.size(); ++g)
            {
                Part& part = filters_[i].parts[g];
                if (!part.s || part.stream_reads < n || part.stream_bases < nb)
                {
                    if (b.raw && !source_of.count(part.dp->device))
                    {
                        err = "the stream that holds a tokenised batch is too small for it"; // (cannot happen: tokenise() sized it)
                        return false;
                    }
                    if (!make_stream(part, std::max<uint64_t>(n, hint_reads_), std::max<uint64_t>(nb, hint_bases_), err))
                        return false;
                    lap(sec_create_);
                }
                if (part.pf_generation != pf_generation_)
                {
                    gn_postfilter pf{ pf_spec_.rel_filter, pf_spec_.fpr_query, pf_active_ ? pf_fpr_[i][g].data() : nullptr,
                                      pf_merge_ ? 2 : (pf_joint_ ? 1 : 0), pf_active_ && pf_merge_ ? pf_gid_[i][g].data() : nullptr };
                    if (gn_stream_set_postfilter(part.s, pf_active_ ? &pf : nullptr) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    part.pf_generation = pf_generation_;
                }
                // the first stream of a device takes the batch (upload + minimisers); the level's other streams on that device --
                // further filters, column parts -- count the same hashes (gn_stream_classify_shared)
                auto src = share_hashes || b.raw ? source_of.find(part.dp->device) : source_of.end(); // (a raw batch lives in one stream per device)
                int rc;
                if (src != source_of.end())
                    rc = gn_stream_classify_shared(part.s, src->second, rel_cutoff[i]);
                else if (b.raw) // the batch is resident already (tokenise())
                {
                    rc = gn_stream_fastq_keep(part.s, n);
                    if (rc == GN_OK)
                        rc = gn_stream_classify(part.s, k, w, rel_cutoff[i]);
                }
                else
                    rc = gn_submit_batch(part.s, b.bases.data(), b.bases.size(), b.off1.data(), b.paired ? b.off2.data() : nullptr, n, k, w,
                                         rel_cutoff[i]);
                if (rc != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                if (src == source_of.end())
                    source_of[part.dp->device] = part.s;
            }
        lap(sec_submit_);
        return true;
    }

    // waits for the batch and fetches: per filter the matches grouped by read, the reads' hash counts, the pre-pass's tallies
    bool classify_end(ReadBatch& b, uint32_t, uint32_t, const std::vector<double>&, BatchResult& out, std::string& err) override
    {
        const uint32_t n = b.raw ? b.raw_keep : (uint32_t)b.size();
        auto           t = std::chrono::steady_clock::now();
        auto           lap = [&](double& acc) {
            const auto now = std::chrono::steady_clock::now();
            acc += std::chrono::duration<double>(now - t).count();
            t = now;
        };
        if (warmed_ && !sets_reserved_ && n)
        {
            // the first real batch says how large the per-read arrays really are: what warm_up()'s guess did not cover is locked now,
            // in one go, instead of one by one under the next batches
            sets_reserved_ = true;
            reserve_result_sets(n, true);
        }
        out.n_hashes.assign(n, 0);
        out.status.assign(n, 0);
        out.per_filter.resize(filters_.size());
        out.prefiltered = false;
        out.max_count.clear();
        out.dropped_rel_filter = out.dropped_fpr_query = 0;
        if (b.raw)
        {
            b.rec_at.resize(n);
            b.seq_at.resize(n);
            b.seq_len.resize(n);
            if (b.paired)
            {
                b.seq_at2.resize(n);
                b.seq_len2.resize(n);
            }
            if (n == 0) // a piece behind the one that stopped its file: nothing of it is input
            {
                for (auto& fr : out.per_filter)
                {
                    fr.match_off.assign(1, 0);
                    fr.matches.clear();
                    fr.fpr_ok.clear();
                }
                return true;
            }
        }
        if (pf_active_ && pf_joint_) // several device filters: the rules need the level's max/min per read
        {
            std::vector<gn_stream*> level;
            for (auto& lf : filters_)
                for (auto& part : lf.parts)
                    level.push_back(part.s);
            if (gn_streams_postfilter_joint(level.data(), (uint32_t)level.size()) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
        }
        bool have_reads = false; // n_hashes / status are the same on every stream: taken once
        for (size_t i = 0; i < filters_.size(); ++i)
        {
            Logical&      lf = filters_[i];
            FilterResult& fr = out.per_filter[i];
            fr.match_off.assign((size_t)n + 1, 0);
            fr.matches.clear();
            fr.fpr_ok.clear();
            if (lf.parts.size() == 1)
            {
                if (!fetch_part(lf.parts[0], n, out, !have_reads, fr, err))
                    return false;
                have_reads = true;
            }
            else
            {
                // several column parts, on this device or on others: the matches of a read are its parts' matches behind each
                // other (targets ascend with the bins, so the concatenation is in target order); put together on the device
                if (!lf.gather)
                {
                    std::vector<const uint32_t*> maps;
                    std::vector<uint32_t>        sizes;
                    for (auto& part : lf.parts)
                    {
                        maps.push_back(part.dp->to_target.empty() ? nullptr : part.dp->to_target.data());
                        sizes.push_back((uint32_t)part.dp->to_target.size());
                    }
                    if (gn_gather_create(device_, (uint32_t)lf.parts.size(), maps.data(), sizes.data(), &lf.gather) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                }
                std::vector<gn_stream*> ss;
                for (auto& part : lf.parts)
                    ss.push_back(part.s);
                uint64_t need = 0;
                if (gn_gather_run(lf.gather, ss.data(), (uint32_t)ss.size()) != GN_OK
                    || gn_gather_fetch(lf.gather, fr.match_off.data(), nullptr, 0, &need) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                fr.matches.resize(need ? need : 1);
                if (gn_gather_fetch(lf.gather, nullptr, reinterpret_cast<gn_match*>(fr.matches.data()), fr.matches.size(), &need) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                fr.matches.resize(need);
                fr.flag_in_count = pf_active_; // (the gather rewrote part-local target ids on the device)
                uint64_t moved = 0;
                if (gn_gather_device_matches(lf.gather, nullptr, nullptr, nullptr, &moved) == GN_OK)
                    gathered_bytes_ += moved;
                if (!have_reads)
                {
                    Part& home = lf.parts[lf.home];
                    if (gn_fetch_batch(home.s, out.n_hashes.data(), out.status.data(), nullptr, nullptr, 0, &need) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    have_reads = true;
                }
            }
            if (pf_active_)
                for (auto& part : lf.parts)
                {
                    const bool first = out.max_count.empty();
                    if (first)
                        out.max_count.assign(n, 0);
                    uint64_t a = 0, b2 = 0;
                    // (in a joint pass every stream holds the level's maximum: the first one's copy is taken)
                    if (gn_fetch_postfilter(part.s, first ? out.max_count.data() : nullptr, &a, &b2) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    out.prefiltered = true;
                    out.dropped_rel_filter += a;
                    out.dropped_fpr_query += b2;
                }
        }
        if (b.raw) // where the records lie in the batch's text (ids and letters are read there)
        {
            gn_stream* first = filters_.front().parts.front().s;
            if (gn_stream_fastq_records(first, b.rec_at.data(), b.seq_at.data(), b.seq_len.data()) != GN_OK
                || (b.paired && gn_stream_text_pair_records2(first, nullptr, b.seq_at2.data(), b.seq_len2.data()) != GN_OK))
            {
                err = gn_last_error();
                return false;
            }
            if (b.dev_text) // the text stayed on the device: the header lines come over, ids are read from them (ReadBatch::id)
            {
                hdr_off_.resize((size_t)n + 1);
                uint64_t hb = 0;
                if (b.text.size() < (size_t)n * 64 + 4096)
                    b.text.resize((size_t)n * 64 + 4096);
                int rc = gn_stream_fastq_headers(first, b.text.data(), b.text.size(), hdr_off_.data(), &hb);
                if (rc == GN_EOVERFLOW)
                {
                    b.text.resize((size_t)hb + 4096);
                    rc = gn_stream_fastq_headers(first, b.text.data(), b.text.size(), hdr_off_.data(), &hb);
                }
                if (rc != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                for (size_t i = 0; i < n; ++i)
                {
                    b.rec_at[i] = hdr_off_[i];
                    b.seq_at[i] = hdr_off_[i + 1];
                }
                b.dev_letters = false;
                if (b.dev_need_letters) // reads this level leaves unclassified go on with their letters (:811-820)
                {
                    b.off1.resize((size_t)n + 1);
                    if (b.paired)
                        b.off2.resize((size_t)n + 1);
                    uint64_t lb = 0;
                    if (b.bases.size() < (size_t)n * 160)
                        b.bases.resize((size_t)n * 160 + 4096);
                    rc = gn_stream_fetch_letters(first, b.bases.data(), b.bases.size(), b.off1.data(), b.paired ? b.off2.data() : nullptr, &lb);
                    if (rc == GN_EOVERFLOW)
                    {
                        b.bases.resize((size_t)lb + 4096);
                        rc = gn_stream_fetch_letters(first, b.bases.data(), b.bases.size(), b.off1.data(), b.paired ? b.off2.data() : nullptr, &lb);
                    }
                    if (rc != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    b.dev_letters = true;
                }
            }
        }
        lap(sec_fetch_);
        return true;
    }

    void prepare(size_t max_reads, size_t max_bases) override
    {
        // streams sized for the largest batch the reader makes, created before the first batch arrives (a stream is ~30 device
        // buffers: tens of milliseconds, and every later re-creation frees memory, which stalls the whole device)
        hint_bases_ = max_bases + max_bases / 16 + 65536; // (a piece of FASTQ text ends at a record's end, a little behind the slab's)
        hint_reads_ = tokenises_fastq() ? std::max<size_t>(max_reads, hint_bases_ / 40) : max_reads;
        std::string err;
        if (!resolve(err))
            return;
        const auto t0 = std::chrono::steady_clock::now();
        for (auto& lf : filters_)
            for (auto& part : lf.parts)
                if (!part.s && !make_stream(part, hint_reads_, hint_bases_, err))
                    return; // (the first batch will report what is wrong)
        sec_create_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // page-locked blocks for three result sets of batches of n reads (n_hashes, max_count, match_off, matches, rec_at, seq_at, seq_len, status)
    static void reserve_result_sets(size_t n, bool only_what_is_short)
    {
        const size_t m = n + n / 8;
        for (size_t bytes : { m * 4, m * 4, m * 8 + 8, m * 12, m * 4, m * 4, m * 4, m })
            if (only_what_is_short)
                PinnedPool::get().ensure_free(bytes, 3);
            else
                PinnedPool::get().reserve(bytes, 3);
    }

    void warm_up(uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff) override
    {
        if (tun().is_set(Knob::no_warm_up))
            return;
        // eight reads of 2 w letters, as a parsed batch or as text -- whichever way the run's batches will come
        ReadBatch   b;
        BatchResult out;
        std::string err, text;
        const std::string letters(std::max<size_t>(2 * w, 8), 'A');
        b.off1.assign(1, 0);
        for (int i = 0; i < 8; ++i)
        {
            text += "@w\n" + letters + "\n+\n" + std::string(letters.size(), 'I') + "\n";
            b.id_buf += "w";
            b.id_off.push_back(b.id_buf.size());
            b.bases.insert(b.bases.end(), letters.begin(), letters.end());
            b.off1.push_back(b.bases.size());
        }
        if (tokenises_fastq())
        {
            ReadBatch r;
            r.raw = true;
            r.text.assign(text.begin(), text.end());
            uint32_t n = 0;
            uint64_t parsed = 0;
            if (tokenise(r, n, parsed, err))
            {
                r.raw_keep = n;
                classify(r, k, w, rel_cutoff, out, err);
            }
        }
        classify(b, k, w, rel_cutoff, out, err); // (errors: the first real batch will report them)
        // page-locked blocks for the batches' text / bases
        {
            // ... and for the batches' text / bases: the reader fills its own window and the batch queue while the filters load; what
            // is in flight behind the queue (device contexts, post pool, writer) needs blocks of its own.  Locking a block while
            // batches are on the device stalls ALL device work for its duration (50 ms holes in the device timeline of the first
            // hundred milliseconds, profiles/r03_e2e_timeline.txt): every block is locked before the first batch.
            PinnedPool::get().reserve(tokenises_fastq() ? hint_bases_ : hint_bases_ / 2 + (1u << 20), 3);
        }
        // ... and for the per-read arrays of the result sets that will be in flight between this context, the post pool and the writer,
        // by a guess: reads of 150 letters with a short id line, ~300 bytes of FASTQ each.  (The first real batch checks the guess:
        // classify_end locks what is short then -- under running batches, which is what this is here to avoid.)
        {
            reserve_result_sets(hint_bases_ / 300, false); // (hint_bases_: the reader's slab size, whichever way its slabs arrive)
        }
        warmed_ = true;
        sec_create_ += sec_submit_ + sec_fetch_ + sec_tok_enqueue_ + sec_tok_wait_;
        sec_submit_ = sec_fetch_ = sec_tok_enqueue_ = sec_tok_wait_ = 0;
    }

    std::string describe() const override
    {
        std::ostringstream os;
        os << "HIP device " << device_ << " (libganon_hip, gfx950)";
        return os.str();
    }

    bool set_long_reads(bool on) override
    {
        long_reads_ = on;
        for (auto& lf : filters_) // (streams that exist already; new ones get it when they are created)
            for (auto& part : lf.parts)
                if (part.s)
                    gn_stream_set_long_reads(part.s, on ? 1 : 0);
        return true;
    }

    bool set_postfilter(const PostFilterSpec* spec) override
    {
        pf_active_ = false;
        ++pf_generation_;
        std::string err;
        if (!spec || !resolve(err) || filters_.empty() || spec->target_fpr.size() != filters_.size())
            return false;
        // filters that share target names: the device replays the level's merge, which needs every name's level-wide id
        const bool merge = filters_.size() > 1 && !spec->disjoint_targets;
        if (merge)
        {
            if (spec->target_gid.size() != filters_.size())
                return false;
            for (size_t i = 0; i < filters_.size(); ++i)
            {
                if (spec->target_gid[i].size() != spec->target_fpr[i].size())
                    return false;
                std::vector<uint32_t> ids = spec->target_gid[i]; // (a filter must name a target once, or there is no merge rule)
                std::sort(ids.begin(), ids.end());
                if (std::adjacent_find(ids.begin(), ids.end()) != ids.end() || (!ids.empty() && ids.back() >= (1u << 28)))
                    return false;
            }
        }
        // one device filter (a whole filter, or a column part of a wide one) per stream; column parts are cut at target
        // boundaries, so their targets are disjoint; all of a level's streams take part in one joint pass, which spans the
        // devices of a partitioned filter (up to 16 streams per device, 16 devices); a merging pass stays on one device
        std::map<int, size_t> per_device;
        size_t                n_streams = 0;
        for (auto const& lf : filters_)
            for (auto const& part : lf.parts)
            {
                ++per_device[part.dp->device];
                ++n_streams;
            }
        for (auto const& [dev, cnt] : per_device)
            if (cnt > 16) // GN_PF_MAX_JOINT
                return false;
        if (per_device.size() > 16 || (merge && (per_device.size() > 1 || n_streams > 16)))
            return false;
        pf_fpr_.assign(filters_.size(), {});
        pf_gid_.assign(filters_.size(), {});
        for (size_t i = 0; i < filters_.size(); ++i)
        {
            const std::vector<double>& fpr = spec->target_fpr[i];
            std::vector<uint8_t>       seen(fpr.size(), 0);
            pf_fpr_[i].resize(filters_[i].parts.size());
            pf_gid_[i].resize(filters_[i].parts.size());
            for (size_t g = 0; g < filters_[i].parts.size(); ++g)
            {
                const DevPart& part = *filters_[i].parts[g].dp;
                if (part.to_target.empty())
                {
                    if (filters_[i].parts.size() != 1)
                        return false;
                    pf_fpr_[i][g] = fpr;
                    if (merge)
                        pf_gid_[i][g] = spec->target_gid[i];
                    continue;
                }
                // device target ids must map one-to-one onto the filter's targets, or a read's matches are not what the host sees
                pf_fpr_[i][g].resize(part.to_target.size(), 0.0);
                if (merge)
                    pf_gid_[i][g].resize(part.to_target.size(), 0u);
                for (size_t d = 0; d < part.to_target.size(); ++d)
                {
                    const uint32_t t = part.to_target[d];
                    if (t >= seen.size() || seen[t])
                        return false;
                    seen[t]          = 1;
                    pf_fpr_[i][g][d] = fpr[t];
                    if (merge)
                        pf_gid_[i][g][d] = spec->target_gid[i][t];
                }
            }
        }
        pf_spec_   = *spec;
        pf_joint_  = n_streams > 1;
        pf_merge_  = merge;
        pf_active_ = true;
        return true;
    }

private:
    struct Part
    {
        const DevPart* dp = nullptr;
        gn_stream*     s  = nullptr;
        uint32_t       stream_reads = 0;
        uint64_t       stream_bases = 0;
        uint64_t       pf_generation = 0; // Backend::set_postfilter call this stream was last configured for
    };
    struct Logical
    {
        bool              is_hibf = false;
        std::vector<Part> parts;  // one, or the column parts of a wide / partitioned flat filter
        size_t            home = 0; // a part on this worker's device (n_hashes / status are fetched from its stream)
        gn_gather*        gather = nullptr;
    };

    bool make_stream(Part& part, uint64_t reads, uint64_t bases, std::string& err)
    {
        if (part.s)
            gn_stream_destroy(part.s);
        part.s            = nullptr;
        const uint32_t cr = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(reads, 1u << 16), 0xFFFFFFF0ull);
        const uint64_t cb = std::max<uint64_t>(bases, 1ull << 24);
        if (gn_stream_create(part.dp->f, cr, cb, 0, &part.s) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        part.stream_reads  = cr;
        part.stream_bases  = cb;
        part.pf_generation = 0;
        ++n_create_;
        if (long_reads_ && gn_stream_set_long_reads(part.s, 1) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        return true;
    }

    // this worker's view of the level's filters (after the DeviceSet has received them all)
    bool resolve(std::string& err)
    {
        auto const& shared = set_->filters();
        if (filters_.size() == shared.size())
            return true;
        drop_streams();
        for (auto const& sf : shared)
        {
            Logical lf;
            lf.is_hibf = sf.is_hibf;
            auto const& copy = sf.spread ? sf.copies.at(0) : sf.copies.at(set_->unique_index(device_));
            for (size_t g = 0; g < copy.size(); ++g)
            {
                Part part;
                part.dp = &copy[g];
                lf.parts.push_back(part);
            }
            for (size_t g = 0; g < lf.parts.size(); ++g)
                if (lf.parts[g].dp->device == device_)
                {
                    lf.home = g;
                    break;
                }
            if (lf.parts.empty())
            {
                err = "a filter without device parts";
                return false;
            }
            filters_.push_back(std::move(lf));
        }
        return true;
    }

    void drop_streams()
    {
        for (auto& lf : filters_)
        {
            for (auto& part : lf.parts)
                if (part.s)
                    gn_stream_destroy(part.s); // (never touches the filter)
            if (lf.gather)
                gn_gather_destroy(lf.gather);
        }
        filters_.clear();
    }

    using Matches = std::vector<Match, ArenaAllocator<Match>>;
    static_assert(sizeof(Match) == sizeof(gn_match) && offsetof(Match, target) == offsetof(gn_match, target) && offsetof(Match, count) == offsetof(gn_match, count),
                  "the device's match records are used as they arrive");

    // records as they came from the device -> the caller's: target ids translated (HIBF: user bin -> target; a part of a partitioned
    // filter: part-local -> the filter's); true when some belong to no target.  The "surely passes --fpr-query" flag stays where the
    // device put it, bit 31 of the count (FilterResult::flag_in_count).
    static bool translate(Matches& matches, const std::vector<uint32_t>* to_target)
    {
        bool drop = false;
        if (to_target)
            for (auto& m : matches)
            {
                m.target = (*to_target)[m.target];
                drop |= m.target == 0xFFFFFFFFu;
            }
        return drop;
    }

    // match offsets and matches of one device filter (and, if wanted, n_hashes / status of the batch), straight into the result's
    // page-locked arrays
    bool fetch_part(Part& part, uint32_t n, BatchResult& out, bool with_reads, FilterResult& fr, std::string& err)
    {
        uint64_t need = 0;
        if (gn_fetch_batch(part.s, with_reads ? out.n_hashes.data() : nullptr, with_reads ? out.status.data() : nullptr, fr.match_off.data(),
                           nullptr, 0, &need)
            != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        fr.matches.resize(need ? need : 1);
        if (gn_fetch_batch(part.s, nullptr, nullptr, nullptr, reinterpret_cast<gn_match*>(fr.matches.data()), fr.matches.size(), &need) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        fr.matches.resize(need);
        fr.flag_in_count = pf_active_;
        if (translate(fr.matches, part.dp->to_target.empty() ? nullptr : &part.dp->to_target))
            drop_unassigned(n, fr);
        return true;
    }

    // user bins that belong to no target (cannot happen with raptor indices) are dropped
    static void drop_unassigned(uint32_t n, FilterResult& fr)
    {
        Matches keep;
        U64Buf  off((size_t)n + 1, 0);
        for (auto const& m : fr.matches)
            if (m.target != 0xFFFFFFFFu)
            {
                keep.push_back(m);
                off[m.read + 1]++;
            }
        for (uint32_t r = 0; r < n; ++r)
            off[r + 1] += off[r];
        fr.matches.swap(keep);
        fr.match_off.swap(off);
    }

    std::shared_ptr<DeviceSet> set_;
    size_t                index_;
    bool                  sets_reserved_ = false;
    bool                  warmed_ = false; // warm_up() has run: the next batch is a real one
    bool                  twin_ = false; // a second set of streams beside the worker with the same index (never receives filters)
    int                   device_;
    double                sec_create_ = 0, sec_submit_ = 0, sec_fetch_ = 0, sec_tok_enqueue_ = 0, sec_tok_wait_ = 0; // $GANON_HOST_TIMING: where classify() spends its time
    unsigned              n_create_ = 0;
    uint64_t              gathered_bytes_ = 0;
    uint64_t              hint_reads_ = 0, hint_bases_ = 0; // largest batch the reader makes (prepare())
    bool                  long_reads_ = false;
    std::vector<Logical>  filters_;
    std::vector<uint32_t>   hdr_off_;     // gn_stream_fastq_headers: offsets of a device-text batch's header lines
    std::vector<gn_stream*> tok_sources_; // streams that hold the FASTQ text of the batch being tokenised (one per device)
    PostFilterSpec        pf_spec_;
    std::vector<std::vector<std::vector<double>>> pf_fpr_; // [filter][part]: per device target of that part
    bool                  pf_joint_ = false;
    bool                  pf_merge_ = false; // the level's filters share targets: the joint pass merges (gn_postfilter.joint = 2)
    std::vector<std::vector<std::vector<uint32_t>>> pf_gid_; // [filter][part][device target] -> level-wide target id
    bool                  pf_active_ = false;
    uint64_t              pf_generation_ = 1;
};

} // namespace

std::vector<std::unique_ptr<Backend>> make_backends(const std::vector<int>& devices, std::string& err)
{
    std::vector<std::unique_ptr<Backend>> out;
    int                                   n = 0;
    // worker threads BLOCK while they wait for the device instead of spinning (HIP's default): the cores are needed by the
    // parsers and the post pool (2-4 device workers on a 16-core quota: +2 .. +17 % end to end, profiles/r03_e2e_ab_sync.txt)
    // (the library's switches: $GANON_HIP_ABLATE, read once when it was loaded -- a list that names no sync mode gets "sync=block")
    {
        const std::string* e  = tun().str(Knob::hip_ablate);
        std::string        sw = e ? *e : "";
        if (sw.find("sync=") == std::string::npos)
        {
            sw += (sw.empty() ? "" : ",") + std::string("sync=block");
            (void)gn_ablate(sw.c_str()); // before the first device call (gn_device_count applies the mode)
        }
    }
    if (gn_device_count(&n) != GN_OK || n <= 0)
    {
        err = std::string("no usable MI355X/HIP device (") + gn_last_error() + "); ganon-classify has no CPU fallback";
        return out;
    }
    std::vector<int> use = devices;
    if (use.empty()) // "all"
        for (int d = 0; d < n; ++d)
            use.push_back(d);
    for (int d : use)
        if (d < 0 || d >= n)
        {
            err = "device index " + std::to_string(d) + " out of range (" + std::to_string(n) + " device(s) visible)";
            return out;
        }
    // one worker per listed device; the filters live in the DeviceSet they share (a device listed again gets a further
    // worker on the same copy)
    auto set = std::make_shared<DeviceSet>(use);
    for (size_t i = 0; i < use.size(); ++i)
        out.emplace_back(new HipBackend(set, i));
    // device-bound host buffers (read batches) come from page-locked memory from now on (hostmem.hpp)
    if (!tun().is_set(Knob::pageable))
    {
        g_host_arena.alloc   = [](size_t n) -> void* { return PinnedPool::get().take(n); };
        g_host_arena.release = [](void* p) { PinnedPool::get().give(p); };
        // (uncompressed FASTQ / FASTA travels as text: a 48 MiB slab plus the slack prepare() adds, i.e. the 56 MiB class)
        const size_t slab = 48u << 20;
        PinnedPool::get().warm_up(out.front()->tokenises_fastq() ? slab + slab / 16 + 65536 : (32u << 20));
    }
    return out;
}

} // namespace gnhost
