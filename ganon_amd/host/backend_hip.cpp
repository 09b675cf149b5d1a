// backend_hip.cpp -- the only Backend of the product: libganon_hip.so through the C ABI (include/ganon_hip.h), one
// instance per GPU.  Fails loudly when no MI355X/HIP device is usable; there is no CPU fallback.
#include "backend.hpp"

#include <ganon_hip.h>

#include <algorithm>
#include <chrono>
#include <iostream>
#include <atomic>
#include <map>
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

// Rows wider than this many 64-bin words are cut into column parts, each a flat IBF of its own on the device (the flat
// count kernels give a row at most 16 wave slices; the reference has no such limit, and ganon-build databases with more
// than 131 072 technical bins exist).  A part is what ganon_amd/partition.py gives one GPU of a bin-range partitioned
// filter, only that all parts live in the same HBM: cut at 64-bin word boundaries between targets, rows re-laid-out per
// part by the strided copy of gn_filter_write_rows, per-target cutoff applied inside the part (a target's bins never
// straddle a cut), matches concatenated per read in part order.
constexpr uint64_t kPartWords = 512; // 32 768 bins: four wave slices of 16-byte lanes (the row shape of BASELINE config 4)


// Page-locked blocks for the read batches, kept for the life of the process.  Locking pages costs ~0.26 s per GiB
// (profiles/r02_pinned_probe.json) and a pipeline's worth of batch buffers is half a GiB or more, so (a) a block, once
// locked, is handed out again instead of being unlocked, and (b) a thread starts locking the first blocks while the filters
// are still being loaded.  Sizes are rounded up to a power of two (>= 1 MiB) so that freed blocks fit later requests.
class PinnedPool
{
public:
    static PinnedPool& get()
    {
        static PinnedPool* p = new PinnedPool(); // (never destroyed: blocks may be released during static teardown)
        return *p;
    }
    void* take(size_t n)
    {
        const size_t cls = size_class(n);
        {
            std::lock_guard<std::mutex> lk(m_);
            auto& fl = free_[cls];
            if (!fl.empty())
            {
                void* p = fl.back();
                fl.pop_back();
                return p;
            }
        }
        void* p = nullptr;
        if (gn_pinned_alloc(cls, &p) != GN_OK)
            return std::malloc(n ? n : 1); // (no more lockable memory: an ordinary buffer still works, its copies are just staged)
        std::lock_guard<std::mutex> lk(m_);
        size_of_[p] = cls;
        return p;
    }
    void give(void* p)
    {
        if (!p)
            return;
        {
            std::lock_guard<std::mutex> lk(m_);
            auto it = size_of_.find(p);
            if (it != size_of_.end())
            {
                free_[it->second].push_back(p);
                return;
            }
        }
        std::free(p); // (the malloc fallback of take())
    }
    // blocks for the first slabs of the reader (32 MiB holds the bases of a 48 MiB FASTQ slab), locked in the background
    void warm_up()
    {
        const char* e = std::getenv("GANON_HOST_PRELOCK_MIB");
        const size_t total = (e ? (size_t)std::atol(e) : 512) << 20, block = 32u << 20;
        if (total == 0)
            return;
        warm_ = std::thread([this, total, block] {
            for (size_t done = 0; done < total && !stop_; done += block)
            {
                void* p = nullptr;
                if (gn_pinned_alloc(block, &p) != GN_OK)
                    return;
                std::lock_guard<std::mutex> lk(m_);
                size_of_[p] = block;
                free_[block].push_back(p);
            }
        });
    }
    void settle() // before the process lets go of the device: the warm-up thread is not in the middle of a call
    {
        stop_ = true;
        if (warm_.joinable())
            warm_.join();
    }

private:
    static size_t size_class(size_t n)
    {
        size_t c = 1u << 20;
        while (c < n)
            c <<= 1;
        return c;
    }
    std::mutex                            m_;
    std::map<size_t, std::vector<void*>>  free_;
    std::unordered_map<void*, size_t>     size_of_;
    std::thread                           warm_;
    std::atomic<bool>                     stop_{ false };
};

class HipBackend final : public Backend
{
public:
    // primary != nullptr: a further worker on the same device.  It classifies against the primary's filters (one copy of the
    // bits in that GPU's HBM, read-only while batches run) with streams of its own.
    explicit HipBackend(int device, HipBackend* primary = nullptr) : device_(device), primary_(primary) {}
    ~HipBackend() override
    {
        PinnedPool::get().settle();
        if (std::getenv("GANON_HOST_TIMING") && n_create_)
            std::cerr << "[backend timing] device " << device_ << ": stream (re)creation " << sec_create_ << " s (" << n_create_
                      << "x), submit calls " << sec_submit_ << " s, waiting for + fetching the results " << sec_fetch_ << " s" << std::endl;
        clear_filters();
        for (auto& s : stage_)
            if (s.ptr)
                gn_pinned_free(s.ptr);
    }

    // ---- FilterSink: the filter is created empty on the device, its rows arrive in chunks ----------------------
    bool begin(const FilterMeta& f, std::string& err) override
    {
        if (primary_)
            return true; // (the primary, earlier in the sink's list, receives the filter)
        Logical lf;
        lf.row_words.clear();
        for (auto const& m : f.shapes)
            lf.row_words.push_back(m.bin_words);
        if (!f.is_hibf)
        {
            const IbfShape&       m = f.shapes.at(0);
            std::vector<uint32_t> bin2target(m.bins, 0xFFFFFFFFu);
            for (size_t t = 0; t < f.targets.size(); ++t)
                for (uint64_t b : f.target_bins[t])
                    bin2target[b] = (uint32_t)t;
            // column parts: [word_lo, word_hi) with cuts moved left to the nearest boundary between two targets
            std::vector<uint64_t> cuts{ 0 };
            while (m.bin_words - cuts.back() > kPartWords)
            {
                // the nearest legal cut at or below the full part width; an even number of words is preferred (16-byte lanes)
                auto legal = [&](uint64_t c) {
                    const uint32_t left = bin2target[c * 64 - 1], right = bin2target[c * 64];
                    return left != right || left == 0xFFFFFFFFu;
                };
                uint64_t c = cuts.back() + kPartWords, odd = 0;
                while (c > cuts.back() + 1 && !(legal(c) && ((c - cuts.back()) & 1u) == 0))
                {
                    if (!odd && legal(c))
                        odd = c;
                    --c;
                }
                if (!(legal(c) && ((c - cuts.back()) & 1u) == 0) && odd)
                    c = odd;
                if (!legal(c))
                {
                    err = "a target owns more than " + std::to_string(kPartWords * 64) + " consecutive technical bins: the filter cannot be cut into column parts";
                    return false;
                }
                cuts.push_back(c);
            }
            cuts.push_back(m.bin_words);
            for (size_t g = 0; g + 1 < cuts.size(); ++g)
            {
                Part part;
                part.word_lo = cuts[g];
                part.words   = cuts[g + 1] - cuts[g];
                const uint64_t bin_lo = part.word_lo * 64, bins = std::min<uint64_t>(m.bins, cuts[g + 1] * 64) - bin_lo;
                std::vector<uint32_t> local(bins, 0xFFFFFFFFu);
                if (cuts.size() == 2)
                    local = bin2target; // the whole filter: target ids are the caller's
                else
                {
                    // local target ids in order of appearance (targets ascend with bins, filter_io.cpp)
                    for (uint64_t b = 0; b < bins; ++b)
                    {
                        const uint32_t t = bin2target[bin_lo + b];
                        if (t == 0xFFFFFFFFu)
                            continue;
                        if (part.to_target.empty() || part.to_target.back() != t)
                        {
                            // (a target seen before can only come back if its bins are not contiguous)
                            auto it = std::find(part.to_target.begin(), part.to_target.end(), t);
                            if (it != part.to_target.end())
                            {
                                local[b] = (uint32_t)(it - part.to_target.begin());
                                continue;
                            }
                            part.to_target.push_back(t);
                        }
                        local[b] = (uint32_t)part.to_target.size() - 1;
                    }
                    for (uint32_t t : part.to_target) // every bin of an owned target must be inside the part
                        for (uint64_t b : f.target_bins[t])
                            if (b < bin_lo || b >= bin_lo + bins)
                            {
                                err = "target '" + f.targets[t] + "' has technical bins on both sides of a column cut (its bins are not "
                                      "contiguous): this filter is too wide for one row group";
                                for (auto& q : lf.parts)
                                    gn_filter_free(q.f);
                                return false;
                            }
                }
                gn_ibf_desc d{ nullptr, m.bin_size, part.words, bins, (uint32_t)m.hash_funs, (uint32_t)m.hash_shift };
                const uint32_t nt = cuts.size() == 2 ? (uint32_t)f.targets.size() : (uint32_t)std::max<size_t>(part.to_target.size(), 1);
                if (gn_filter_upload_ibf(device_, &d, local.data(), nt, &part.f) != GN_OK)
                {
                    err = gn_last_error();
                    for (auto& q : lf.parts)
                        gn_filter_free(q.f);
                    return false;
                }
                lf.parts.push_back(std::move(part));
            }
        }
        else
        {
            std::vector<gn_ibf_desc>    descs;
            std::vector<const int64_t*> nx, bu;
            for (size_t i = 0; i < f.shapes.size(); ++i)
            {
                auto& m = f.shapes[i];
                descs.push_back(gn_ibf_desc{ nullptr, m.bin_size, m.bin_words, m.bins, (uint32_t)m.hash_funs, (uint32_t)m.hash_shift });
                nx.push_back(f.next_ibf_id[i].data());
                bu.push_back(f.bin_to_user[i].data());
            }
            Part part;
            if (gn_filter_upload_hibf(device_, (uint32_t)descs.size(), descs.data(), nx.data(), bu.data(), f.n_user_bins, &part.f) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
            // user bin -> target index (select_matches(THIBF) reads counts[bins[0]], GanonClassify.cpp:556-558)
            part.to_target.assign(f.n_user_bins, 0xFFFFFFFFu);
            for (size_t t = 0; t < f.targets.size(); ++t)
                part.to_target[f.target_bins[t][0]] = (uint32_t)t;
            lf.is_hibf = true;
            lf.parts.push_back(std::move(part));
        }
        filters_.push_back(std::move(lf));
        return true;
    }

    uint64_t* staging(int which, size_t bytes) override
    {
        if (primary_)
            return primary_->staging(which, bytes);
        Stage& s = stage_[which & 1];
        if (s.bytes < bytes)
        {
            if (s.ptr)
                gn_pinned_free(s.ptr);
            s.ptr   = nullptr;
            s.bytes = 0;
            void* p = nullptr;
            if (gn_pinned_alloc(bytes, &p) != GN_OK)
                return nullptr; // the loader falls back to pageable memory
            s.ptr   = p;
            s.bytes = bytes;
        }
        return static_cast<uint64_t*>(s.ptr);
    }

    bool rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err) override
    {
        // asynchronous on the filter's load stream when src is pinned (gn_filter_write_rows); src holds whole rows, every
        // column part takes its words of them
        if (primary_)
            return true;
        Logical& lf = filters_.back();
        if (ibf >= lf.row_words.size())
        {
            err = "rows for an IBF the filter does not have";
            return false;
        }
        for (auto& part : lf.parts)
            if (gn_filter_write_rows(part.f, lf.is_hibf ? ibf : 0, row_begin, n_rows, src, lf.row_words[ibf], part.word_lo) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
        return true;
    }

    bool drain(std::string& err) override
    {
        if (!primary_ && !filters_.empty())
            for (auto& part : filters_.back().parts)
                if (gn_filter_write_sync(part.f) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
        return true;
    }

    bool end(std::string& err) override
    {
        if (primary_)
        {
            // the primary has just finished this filter: take its device filters, without streams
            Logical lf = primary_->filters_.at(filters_.size());
            for (auto& part : lf.parts)
            {
                part.s             = nullptr;
                part.stream_reads  = 0;
                part.stream_bases  = 0;
                part.pf_generation = 0;
            }
            filters_.push_back(std::move(lf));
            return true;
        }
        for (auto& part : filters_.back().parts)
            if (gn_filter_finalize(part.f) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
        return true;
    }

    void clear_filters() override
    {
        for (auto& lf : filters_)
            for (auto& part : lf.parts)
            {
                if (part.s)
                    gn_stream_destroy(part.s);
                if (!primary_) // (a further worker's streams do not need the filter to go away: gn_stream_destroy never touches it)
                    gn_filter_free(part.f);
            }
        filters_.clear();
    }

    bool classify(const ReadBatch& b, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out,
                  std::string& err) override
    {
        const uint32_t n = (uint32_t)b.size();
        auto           t = std::chrono::steady_clock::now();
        auto           lap = [&](double& acc) {
            const auto now = std::chrono::steady_clock::now();
            acc += std::chrono::duration<double>(now - t).count();
            t = now;
        };
        out.n_hashes.assign(n, 0);
        out.status.assign(n, 0);
        out.per_filter.resize(filters_.size());
        out.prefiltered = false;
        out.max_count.clear();
        out.dropped_rel_filter = out.dropped_fpr_query = 0;
        // submit to every stream first (asynchronous), then fetch
        const uint64_t nb = std::max<uint64_t>(b.bases.size(), 1);
        for (size_t i = 0; i < filters_.size(); ++i)
            for (size_t g = 0; g < filters_[i].parts.size(); ++g)
            {
                Part& part = filters_[i].parts[g];
                if (!part.s || part.stream_reads < n || part.stream_bases < nb)
                {
                    if (part.s)
                        gn_stream_destroy(part.s);
                    part.s            = nullptr;
                    const uint32_t cr = std::max<uint32_t>(n, 1u << 16);
                    const uint64_t cb = std::max<uint64_t>(nb, 1ull << 24);
                    if (gn_stream_create(part.f, cr, cb, 0, &part.s) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    part.stream_reads = cr;
                    part.stream_bases = cb;
                    part.pf_generation = 0;
                    lap(sec_create_);
                    ++n_create_;
                    if (long_reads_ && gn_stream_set_long_reads(part.s, 1) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
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
                if (gn_submit_batch(part.s, b.bases.data(), b.bases.size(), b.off1.data(), b.paired ? b.off2.data() : nullptr, n, k, w,
                                    rel_cutoff[i])
                    != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
            }
        lap(sec_submit_);
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
        for (size_t i = 0; i < filters_.size(); ++i)
        {
            Logical&      lf = filters_[i];
            FilterResult& fr = out.per_filter[i];
            fr.match_off.assign((size_t)n + 1, 0);
            fr.matches.clear();
            if (lf.parts.size() == 1)
            {
                fr.fpr_ok.clear();
                if (!fetch_part(lf.parts[0], n, out, fr.match_off, fr.matches, err, pf_active_ ? &fr.fpr_ok : nullptr))
                    return false;
                if (pf_active_)
                {
                    const bool first = out.max_count.empty();
                    if (first)
                        out.max_count.assign(n, 0);
                    uint64_t a = 0, b2 = 0;
                    // (in a joint pass every stream holds the level's maximum: the first one's copy is taken)
                    if (gn_fetch_postfilter(lf.parts[0].s, first ? out.max_count.data() : nullptr, &a, &b2) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    out.prefiltered = true;
                    out.dropped_rel_filter += a;
                    out.dropped_fpr_query += b2;
                }
                continue;
            }
            // several column parts: the matches of a read are its parts' matches behind each other (targets ascend with
            // the bins, so the concatenation is already in target order)
            std::vector<std::vector<uint64_t>> offs(lf.parts.size());
            std::vector<std::vector<Match>>    ms(lf.parts.size());
            std::vector<std::vector<uint8_t>>  oks(lf.parts.size());
            for (size_t g = 0; g < lf.parts.size(); ++g)
            {
                offs[g].assign((size_t)n + 1, 0);
                if (!fetch_part(lf.parts[g], n, out, offs[g], ms[g], err, pf_active_ ? &oks[g] : nullptr))
                    return false;
                if (pf_active_)
                {
                    const bool first = out.max_count.empty();
                    if (first)
                        out.max_count.assign(n, 0);
                    uint64_t a = 0, b2 = 0;
                    if (gn_fetch_postfilter(lf.parts[g].s, first ? out.max_count.data() : nullptr, &a, &b2) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    out.prefiltered = true;
                    out.dropped_rel_filter += a;
                    out.dropped_fpr_query += b2;
                }
                for (uint32_t r = 0; r < n; ++r)
                    fr.match_off[r + 1] += offs[g][r + 1] - offs[g][r];
            }
            for (uint32_t r = 0; r < n; ++r)
                fr.match_off[r + 1] += fr.match_off[r];
            fr.matches.resize(fr.match_off[n]);
            fr.fpr_ok.clear();
            if (pf_active_)
                fr.fpr_ok.resize(fr.match_off[n]);
            for (uint32_t r = 0; r < n; ++r)
            {
                uint64_t o = fr.match_off[r];
                for (size_t g = 0; g < lf.parts.size(); ++g)
                    for (uint64_t x = offs[g][r]; x < offs[g][r + 1]; ++x)
                    {
                        if (pf_active_)
                            fr.fpr_ok[o] = oks[g][x];
                        fr.matches[o++] = ms[g][x];
                    }
            }
        }
        lap(sec_fetch_);
        return true;
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
        if (!spec || filters_.empty() || spec->target_fpr.size() != filters_.size())
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
        // boundaries, so their targets are disjoint; all of a level's streams take part in one joint pass
        size_t n_streams = 0;
        for (auto const& lf : filters_)
            n_streams += lf.parts.size();
        if (n_streams > 16) // GN_PF_MAX_JOINT
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
                const Part& part = filters_[i].parts[g];
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
    struct Stage
    {
        void*  ptr   = nullptr;
        size_t bytes = 0;
    };
    struct Part
    {
        gn_filter*            f = nullptr;
        gn_stream*            s = nullptr;
        uint32_t              stream_reads = 0;
        uint64_t              stream_bases = 0;
        uint64_t              pf_generation = 0; // Backend::set_postfilter call this stream was last configured for
        uint64_t              word_lo = 0, words = 0;
        std::vector<uint32_t> to_target; // device target id -> index into FilterMeta::targets (empty: the same)
    };
    struct Logical
    {
        bool                  is_hibf = false;
        std::vector<Part>     parts;     // one, or the column parts of a wide flat filter
        std::vector<uint64_t> row_words; // bin_words of every IBF as stored in the file
    };

    // n_hashes / status (the same for every part), match offsets and matches of one device filter, target ids translated
    bool fetch_part(Part& part, uint32_t n, BatchResult& out, std::vector<uint64_t>& match_off, std::vector<Match>& matches,
                    std::string& err, std::vector<uint8_t>* fpr_ok = nullptr)
    {
        uint64_t need = 0;
        if (gn_fetch_batch(part.s, out.n_hashes.data(), out.status.data(), match_off.data(), nullptr, 0, &need) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        tmp_.resize(need ? need : 1);
        if (gn_fetch_batch(part.s, nullptr, nullptr, nullptr, tmp_.data(), tmp_.size(), &need) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        matches.resize(need);
        bool drop = false;
        for (uint64_t j = 0; j < need; ++j)
        {
            uint32_t t = tmp_[j].target;
            if (!part.to_target.empty())
            {
                t = part.to_target[t]; // HIBF: user bin -> target; column part: local -> global target
                drop |= t == 0xFFFFFFFFu;
            }
            matches[j] = Match{ tmp_[j].read, t, tmp_[j].count & ~GN_MATCH_FPR_OK };
        }
        if (fpr_ok)
        {
            fpr_ok->resize(need);
            for (uint64_t j = 0; j < need; ++j)
                (*fpr_ok)[j] = (tmp_[j].count & GN_MATCH_FPR_OK) ? 1 : 0;
        }
        if (drop)
        {
            // user bins that belong to no target (cannot happen with raptor indices) are dropped
            std::vector<Match>    keep;
            std::vector<uint64_t> off((size_t)n + 1, 0);
            for (auto const& m : matches)
                if (m.target != 0xFFFFFFFFu)
                {
                    keep.push_back(m);
                    off[m.read + 1]++;
                }
            for (uint32_t r = 0; r < n; ++r)
                off[r + 1] += off[r];
            matches.swap(keep);
            match_off.swap(off);
        }
        return true;
    }

    int                   device_;
    HipBackend*           primary_ = nullptr;
    double                sec_create_ = 0, sec_submit_ = 0, sec_fetch_ = 0; // $GANON_HOST_TIMING: where classify() spends its time
    unsigned              n_create_ = 0;
    bool                  long_reads_ = false;
    Stage                 stage_[2];
    std::vector<Logical>  filters_;
    std::vector<gn_match> tmp_;
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
    // one backend per listed device; a device listed again gets a further worker that shares the first one's filters
    std::map<int, HipBackend*> first;
    for (int d : use)
    {
        auto it = first.find(d);
        auto* b = new HipBackend(d, it == first.end() ? nullptr : it->second);
        if (it == first.end())
            first[d] = b;
        out.emplace_back(b);
    }
    // device-bound host buffers (read batches) come from page-locked memory from now on (hostmem.hpp)
    if (!std::getenv("GANON_HOST_PAGEABLE"))
    {
        g_host_arena.alloc   = [](size_t n) -> void* { return PinnedPool::get().take(n); };
        g_host_arena.release = [](void* p) { PinnedPool::get().give(p); };
        PinnedPool::get().warm_up();
    }
    return out;
}

} // namespace gnhost
