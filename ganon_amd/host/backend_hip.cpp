// backend_hip.cpp -- the only Backend of the product: libganon_hip.so through the C ABI (include/ganon_hip.h), one
// instance per GPU.  Fails loudly when no MI355X/HIP device is usable; there is no CPU fallback.
#include "backend.hpp"

#include <ganon_hip.h>

#include <algorithm>
#include <sstream>

namespace gnhost
{

namespace
{

class HipBackend final : public Backend
{
public:
    explicit HipBackend(int device) : device_(device) {}
    ~HipBackend() override
    {
        clear_filters();
        for (auto& s : stage_)
            if (s.ptr)
                gn_pinned_free(s.ptr);
    }

    // ---- FilterSink: the filter is created empty on the device, its rows arrive in chunks ----------------------
    bool begin(const FilterMeta& f, std::string& err) override
    {
        gn_filter* h  = nullptr;
        int        rc = 0;
        if (!f.is_hibf)
        {
            const IbfShape& m = f.shapes.at(0);
            gn_ibf_desc     d{ nullptr, m.bin_size, m.bin_words, m.bins, (uint32_t)m.hash_funs, (uint32_t)m.hash_shift };
            std::vector<uint32_t> bin2target(m.bins, 0xFFFFFFFFu);
            for (size_t t = 0; t < f.targets.size(); ++t)
                for (uint64_t b : f.target_bins[t])
                    bin2target[b] = (uint32_t)t;
            rc = gn_filter_upload_ibf(device_, &d, bin2target.data(), (uint32_t)f.targets.size(), &h);
            userbin_to_target_.emplace_back();
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
            rc = gn_filter_upload_hibf(device_, (uint32_t)descs.size(), descs.data(), nx.data(), bu.data(), f.n_user_bins, &h);
            // user bin -> target index (select_matches(THIBF) reads counts[bins[0]], GanonClassify.cpp:556-558)
            std::vector<uint32_t> ub2t(f.n_user_bins, 0xFFFFFFFFu);
            for (size_t t = 0; t < f.targets.size(); ++t)
                ub2t[f.target_bins[t][0]] = (uint32_t)t;
            userbin_to_target_.push_back(std::move(ub2t));
        }
        if (rc != GN_OK)
        {
            err = gn_last_error();
            userbin_to_target_.pop_back();
            return false;
        }
        loading_words_.clear();
        for (auto const& m : f.shapes)
            loading_words_.push_back(m.bin_words);
        filters_.push_back(h);
        streams_.push_back(nullptr);
        stream_reads_.push_back(0);
        stream_bases_.push_back(0);
        return true;
    }

    uint64_t* staging(int which, size_t bytes) override
    {
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
        // asynchronous on the filter's load stream when src is pinned (gn_filter_write_rows); src holds whole rows
        if (ibf >= loading_words_.size())
        {
            err = "rows for an IBF the filter does not have";
            return false;
        }
        if (gn_filter_write_rows(filters_.back(), ibf, row_begin, n_rows, src, loading_words_[ibf], 0) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        return true;
    }

    bool drain(std::string& err) override
    {
        if (!filters_.empty() && gn_filter_write_sync(filters_.back()) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        return true;
    }

    bool end(std::string& err) override
    {
        if (gn_filter_finalize(filters_.back()) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        return true;
    }

    void clear_filters() override
    {
        for (auto* s : streams_)
            if (s)
                gn_stream_destroy(s);
        for (auto* f : filters_)
            gn_filter_free(f);
        streams_.clear();
        filters_.clear();
        stream_reads_.clear();
        stream_bases_.clear();
        userbin_to_target_.clear();
    }

    bool classify(const ReadBatch& b, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out,
                  std::string& err) override
    {
        const uint32_t n = (uint32_t)b.size();
        out.n_hashes.assign(n, 0);
        out.status.assign(n, 0);
        out.per_filter.resize(filters_.size());
        // submit to every filter's stream first (asynchronous), then fetch
        for (size_t i = 0; i < filters_.size(); ++i)
        {
            const uint64_t nb = std::max<uint64_t>(b.bases.size(), 1);
            if (!streams_[i] || stream_reads_[i] < n || stream_bases_[i] < nb)
            {
                if (streams_[i])
                    gn_stream_destroy(streams_[i]);
                streams_[i]       = nullptr;
                const uint32_t cr = std::max<uint32_t>(n, 1u << 16);
                const uint64_t cb = std::max<uint64_t>(nb, 1ull << 24);
                if (gn_stream_create(filters_[i], cr, cb, 0, &streams_[i]) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
                stream_reads_[i] = cr;
                stream_bases_[i] = cb;
            }
            if (gn_submit_batch(streams_[i], b.bases.data(), b.bases.size(), b.off1.data(), b.paired ? b.off2.data() : nullptr, n,
                                k, w, rel_cutoff[i])
                != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
        }
        for (size_t i = 0; i < filters_.size(); ++i)
        {
            FilterResult& fr = out.per_filter[i];
            fr.match_off.assign((size_t)n + 1, 0);
            uint64_t need = 0;
            if (gn_fetch_batch(streams_[i], out.n_hashes.data(), out.status.data(), fr.match_off.data(), nullptr, 0, &need) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
            tmp_.resize(need ? need : 1);
            if (gn_fetch_batch(streams_[i], nullptr, nullptr, nullptr, tmp_.data(), tmp_.size(), &need) != GN_OK)
            {
                err = gn_last_error();
                return false;
            }
            fr.matches.resize(need);
            const auto& ub2t = userbin_to_target_[i];
            for (uint64_t j = 0; j < need; ++j)
            {
                uint32_t t = tmp_[j].target;
                if (!ub2t.empty())
                    t = ub2t[t]; // HIBF reports user bins
                fr.matches[j] = Match{ tmp_[j].read, t, tmp_[j].count };
            }
            if (!ub2t.empty())
            {
                // user bins that belong to no target (cannot happen with raptor indices) are dropped
                bool drop = false;
                for (auto const& m : fr.matches)
                    if (m.target == 0xFFFFFFFFu)
                        drop = true;
                if (drop)
                {
                    std::vector<Match>    keep;
                    std::vector<uint64_t> off((size_t)n + 1, 0);
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
            }
        }
        return true;
    }

    std::string describe() const override
    {
        std::ostringstream os;
        os << "HIP device " << device_ << " (libganon_hip, gfx950)";
        return os.str();
    }

private:
    struct Stage
    {
        void*  ptr   = nullptr;
        size_t bytes = 0;
    };
    int                                device_;
    Stage                              stage_[2];
    std::vector<gn_filter*>            filters_;
    std::vector<gn_stream*>            streams_;
    std::vector<uint32_t>              stream_reads_;
    std::vector<uint64_t>              stream_bases_;
    std::vector<std::vector<uint32_t>> userbin_to_target_;
    std::vector<gn_match>              tmp_;
    std::vector<uint64_t>              loading_words_; // bin_words of every IBF of the filter being loaded
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
    for (int d : use)
        out.emplace_back(new HipBackend(d));
    return out;
}

} // namespace gnhost
