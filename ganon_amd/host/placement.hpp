// placement.hpp -- where the filters of a hierarchy level live on the devices of a run (DeviceSet): replicated per GPU, or a flat
// IBF cut by technical-bin range into column parts on several GPUs.  Used by backend_hip.cpp only.
#pragma once

#include "backend.hpp"
#include "pinned_pool.hpp"

namespace gnhost
{

// Rows wider than this many 64-bin words are cut into column parts, each a flat IBF of its own on the device (the flat
// count kernels give a row at most 16 wave slices; the reference has no such limit, and ganon-build databases with more
// than 131 072 technical bins exist).  A part is what ganon_amd/partition.py gives one GPU of a bin-range partitioned
// filter, only that all parts live in the same HBM: cut at 64-bin word boundaries between targets, rows re-laid-out per
// part by the strided copy of gn_filter_write_rows, per-target cutoff applied inside the part (a target's bins never
// straddle a cut), matches concatenated per read in part order.
constexpr uint64_t kPartWords = 512; // 32 768 bins: four wave slices of 16-byte lanes (the row shape of BASELINE config 4)



// ---- where the filters of a hierarchy level live -------------------------------------------------------------------
// One DeviceSet per run, shared by all workers.  It receives every filter of the level once (FilterSink) and decides per
// filter, from the filter's size and what the devices have left:
//   * REPLICATED: a copy on every GPU in use (read-sharded classification, SURVEY 8e first bullet) -- entries of --device
//     that name the same GPU share its copy;
//   * PARTITIONED (flat IBF only): the reference loads a filter of any size (GanonClassify.cpp:949-986,1007-1039); one that
//     does not fit beside what a device already holds is cut by technical-bin range at target boundaries and its column
//     parts are placed on different devices (SURVEY 8e second bullet, BASELINE config 5).  Every part is at most kPartWords
//     wide, so a device may hold several.
// A "device" for placement is a GPU with the memory it has free at the start of the run (minus a reserve for the batch
// buffers), or -- with $GANON_DEVICE_BUDGET=<bytes, K/M/G/T suffix allowed> -- every --device ENTRY with that budget, so
// that `--device 0,0,0` with a small budget exercises the partitioned path on one GPU (tests).
struct DevPart
{
    gn_filter*            f      = nullptr;
    int                   device = 0;
    int                   vdev   = 0;
    uint64_t              word_lo = 0, words = 0;
    std::vector<uint32_t> to_target; // device target id -> index into FilterMeta::targets (empty: the same)
};

struct SharedFilter
{
    bool                              is_hibf = false;
    bool                              spread  = false; // partitioned over the placement devices
    std::vector<uint64_t>             row_words;       // bin_words of every IBF as stored in the file
    // replicated: copies[u] = the filter's parts on unique device u; partitioned: copies[0] = all parts, in column order
    std::vector<std::vector<DevPart>> copies;
};

inline uint64_t parse_bytes(const char* v)
{
    char*  end = nullptr;
    double x   = std::strtod(v, &end);
    if (end)
        switch (*end)
        {
            case 'k': case 'K': x *= 1024.0; break;
            case 'm': case 'M': x *= 1024.0 * 1024.0; break;
            case 'g': case 'G': x *= 1024.0 * 1024.0 * 1024.0; break;
            case 't': case 'T': x *= 1024.0 * 1024.0 * 1024.0 * 1024.0; break;
            default: break;
        }
    return x > 0 ? (uint64_t)x : 0;
}

class DeviceSet
{
public:
    explicit DeviceSet(const std::vector<int>& entries) : entries_(entries)
    {
        for (int d : entries)
            if (std::find(uniq_.begin(), uniq_.end(), d) == uniq_.end())
                uniq_.push_back(d);
        const std::string* e = tun().str(Knob::device_budget);
        if (e && !e->empty())
        {
            const uint64_t b = parse_bytes(e->c_str());
            for (int d : entries)
                vdev_.push_back(VDev{ d, b, 0 });
            virtual_ = true;
        }
        else
            for (int d : uniq_)
            {
                uint64_t fr = 0, tot = 0;
                if (gn_device_memory(d, &fr, &tot) != GN_OK)
                    fr = 0;
                // what the batch buffers of the workers need stays out of the filters' budget
                const uint64_t reserve = std::max<uint64_t>(fr / 8, std::min<uint64_t>(fr / 2, 16ull << 30));
                vdev_.push_back(VDev{ d, fr > reserve ? fr - reserve : 0, 0 });
            }
    }
    ~DeviceSet()
    {
        clear();
        for (auto& s : stage_)
            if (s.ptr)
                gn_pinned_free(s.ptr);
    }

    const std::vector<int>& entries() const { return entries_; }
    const std::vector<SharedFilter>& filters() const { return filters_; }
    bool any_spread() const
    {
        for (auto const& f : filters_)
            if (f.spread)
                return true;
        return false;
    }
    size_t n_devices() const { return uniq_.size(); }
    const std::vector<int>& unique_devices() const { return uniq_; }
    size_t unique_index(int device) const { return (size_t)(std::find(uniq_.begin(), uniq_.end(), device) - uniq_.begin()); }

    void clear()
    {
        for (auto& sf : filters_)
            for (auto& copy : sf.copies)
                for (auto& part : copy)
                    if (part.f)
                        gn_filter_free(part.f);
        filters_.clear();
        for (auto& v : vdev_)
            v.used = 0;
        log_.clear();
    }

    std::string placement() const { return log_; }

    bool begin(const FilterMeta& f, std::string& err)
    {
        SharedFilter sf;
        for (auto const& m : f.shapes)
            sf.row_words.push_back(m.bin_words);
        uint64_t bytes = 0;
        for (auto const& m : f.shapes) // (an HIBF's rows are padded to whole lines on the device: what it takes there, not the file's payload)
            bytes += f.is_hibf ? m.bin_size * gn_hibf_row_stride_words(m.bin_words) * 8 : m.payload_bytes();
        bool fits = true;
        for (auto const& v : vdev_)
            fits = fits && v.used + bytes <= v.budget;
        std::ostringstream note;
        note << "filter " << filters_.size() << " (" << (f.is_hibf ? "HIBF, " : "IBF, ") << bytes / double(1ull << 30) << " GiB): ";
        if (f.is_hibf)
        {
            if (!fits)
            {
                err = "the HIBF (" + std::to_string(bytes >> 20) + " MiB) does not fit into the memory of one device beside the filters "
                      "loaded before it; an HIBF cannot be partitioned by bin range (its levels are data dependent)";
                return false;
            }
            sf.is_hibf = true;
            for (int d : uniq_)
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
                DevPart part;
                part.device = d;
                if (gn_filter_upload_hibf(d, (uint32_t)descs.size(), descs.data(), nx.data(), bu.data(), f.n_user_bins, &part.f) != GN_OK)
                {
                    err = gn_last_error();
                    free_parts(sf);
                    return false;
                }
                // user bin -> target index (select_matches(THIBF) reads counts[bins[0]], GanonClassify.cpp:556-558)
                part.to_target.assign(f.n_user_bins, 0xFFFFFFFFu);
                for (size_t t = 0; t < f.targets.size(); ++t)
                    part.to_target[f.target_bins[t][0]] = (uint32_t)t;
                sf.copies.push_back({});
                sf.copies.back().push_back(std::move(part));
            }
            for (auto& v : vdev_)
                v.used += bytes;
            note << "replicated on " << uniq_.size() << " device(s)";
        }
        else
        {
            const IbfShape&       m = f.shapes.at(0);
            std::vector<uint32_t> bin2target(m.bins, 0xFFFFFFFFu);
            for (size_t t = 0; t < f.targets.size(); ++t)
                for (uint64_t b : f.target_bins[t])
                    bin2target[b] = (uint32_t)t;
            // a cut between words c-1 and c is legal where no target has bins on both sides
            auto legal = [&](uint64_t c) {
                const uint32_t left = bin2target[c * 64 - 1], right = bin2target[c * 64];
                return left != right || left == 0xFFFFFFFFu;
            };
            // the largest legal cut in (from, limit]; an even number of words is preferred (16-byte lanes); 0 = none
            auto next_cut = [&](uint64_t from, uint64_t limit) -> uint64_t {
                if (limit >= m.bin_words)
                    return m.bin_words;
                uint64_t c = limit, odd = 0;
                while (c > from + 1 && !(legal(c) && ((c - from) & 1u) == 0))
                {
                    if (!odd && legal(c))
                        odd = c;
                    --c;
                }
                if (!(legal(c) && ((c - from) & 1u) == 0) && odd)
                    c = odd;
                return c > from && legal(c) ? c : 0;
            };
            // [word_lo, word_hi) per placement device
            struct Range
            {
                size_t   vdev;
                uint64_t lo, hi;
            };
            std::vector<Range> ranges;
            if (fits)
                ranges.push_back(Range{ 0, 0, m.bin_words }); // (replicated: the same cuts on every device)
            else
            {
                sf.spread = true;
                const uint64_t row_group_bytes = m.bin_size * 8; // one 64-bin word of every row
                uint64_t       at = 0;
                for (size_t v = 0; v < vdev_.size() && at < m.bin_words; ++v)
                {
                    const uint64_t room  = vdev_[v].budget > vdev_[v].used ? (vdev_[v].budget - vdev_[v].used) / row_group_bytes : 0;
                    const uint64_t left  = m.bin_words - at, devs = vdev_.size() - v;
                    const uint64_t share = std::min<uint64_t>(room, (left + devs - 1) / devs);
                    if (share == 0)
                        continue;
                    const uint64_t c = next_cut(at, at + share);
                    if (c == 0)
                        continue; // (no boundary between targets inside this device's share: the next one may have more room)
                    ranges.push_back(Range{ v, at, c });
                    at = c;
                }
                if (at < m.bin_words)
                {
                    err = "the filter (" + std::to_string(bytes >> 20) + " MiB) does not fit into the " + std::to_string(vdev_.size())
                          + " device(s) given (" + std::to_string(budget_left() >> 20) + " MiB left for filters in total" +
                          (virtual_ ? ", $GANON_DEVICE_BUDGET" : "") + ")";
                    return false;
                }
            }
            auto make_parts = [&](const Range& rg, int device, int vdev, std::vector<DevPart>& out) -> bool {
                std::vector<uint64_t> cuts{ rg.lo };
                while (rg.hi - cuts.back() > kPartWords)
                {
                    const uint64_t c = next_cut(cuts.back(), cuts.back() + kPartWords);
                    if (c == 0)
                    {
                        err = "a target owns more than " + std::to_string(kPartWords * 64) + " consecutive technical bins: the filter cannot be cut into column parts";
                        return false;
                    }
                    cuts.push_back(c);
                }
                cuts.push_back(rg.hi);
                const bool whole = rg.lo == 0 && rg.hi == m.bin_words && cuts.size() == 2;
                for (size_t g = 0; g + 1 < cuts.size(); ++g)
                {
                    DevPart part;
                    part.device  = device;
                    part.vdev    = vdev;
                    part.word_lo = cuts[g];
                    part.words   = cuts[g + 1] - cuts[g];
                    const uint64_t bin_lo = part.word_lo * 64, bins = std::min<uint64_t>(m.bins, cuts[g + 1] * 64) - bin_lo;
                    std::vector<uint32_t> local(bins, 0xFFFFFFFFu);
                    if (whole)
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
                                    return false;
                                }
                    }
                    gn_ibf_desc d{ nullptr, m.bin_size, part.words, bins, (uint32_t)m.hash_funs, (uint32_t)m.hash_shift };
                    const uint32_t nt = whole ? (uint32_t)f.targets.size() : (uint32_t)std::max<size_t>(part.to_target.size(), 1);
                    if (gn_filter_upload_ibf(device, &d, local.data(), nt, &part.f) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
                    out.push_back(std::move(part));
                }
                return true;
            };
            if (!sf.spread)
            {
                for (int d : uniq_)
                {
                    sf.copies.push_back({});
                    if (!make_parts(ranges[0], d, -1, sf.copies.back()))
                    {
                        free_parts(sf);
                        return false;
                    }
                }
                for (auto& v : vdev_)
                    v.used += bytes;
                note << "replicated on " << uniq_.size() << " device(s)";
                if (sf.copies[0].size() > 1)
                    note << ", " << sf.copies[0].size() << " column parts each";
            }
            else
            {
                sf.copies.push_back({});
                note << "partitioned by bin range:";
                for (auto const& rg : ranges)
                {
                    const size_t before = sf.copies[0].size();
                    if (!make_parts(rg, vdev_[rg.vdev].device, (int)rg.vdev, sf.copies[0]))
                    {
                        free_parts(sf);
                        return false;
                    }
                    vdev_[rg.vdev].used += (rg.hi - rg.lo) * m.bin_size * 8;
                    note << " [words " << rg.lo << "-" << rg.hi << " -> device " << vdev_[rg.vdev].device << ", "
                         << sf.copies[0].size() - before << " part(s)]";
                }
                if (sf.copies[0].size() > 64) // GN_GATHER_MAX_PARTS
                {
                    err = "the filter would be cut into " + std::to_string(sf.copies[0].size()) + " column parts (at most 64)";
                    free_parts(sf);
                    return false;
                }
            }
        }
        log_ += note.str() + "\n";
        filters_.push_back(std::move(sf));
        return true;
    }

    uint64_t* staging(int which, size_t bytes)
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

    // asynchronous on every device filter's load stream when src is pinned (gn_filter_write_rows); src holds whole rows,
    // every column part -- on whichever device -- takes its words of them
    bool rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err)
    {
        SharedFilter& sf = filters_.back();
        if (ibf >= sf.row_words.size())
        {
            err = "rows for an IBF the filter does not have";
            return false;
        }
        for (auto& copy : sf.copies)
            for (auto& part : copy)
                if (gn_filter_write_rows(part.f, sf.is_hibf ? ibf : 0, row_begin, n_rows, src, sf.row_words[ibf], part.word_lo) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
        return true;
    }

    bool drain(std::string& err)
    {
        if (!filters_.empty())
            for (auto& copy : filters_.back().copies)
                for (auto& part : copy)
                    if (gn_filter_write_sync(part.f) != GN_OK)
                    {
                        err = gn_last_error();
                        return false;
                    }
        return true;
    }

    bool end(std::string& err)
    {
        for (auto& copy : filters_.back().copies)
            for (auto& part : copy)
                if (gn_filter_finalize(part.f) != GN_OK)
                {
                    err = gn_last_error();
                    return false;
                }
        return true;
    }

private:
    struct VDev
    {
        int      device;
        uint64_t budget, used;
    };
    struct Stage
    {
        void*  ptr   = nullptr;
        size_t bytes = 0;
    };
    uint64_t budget_left() const
    {
        uint64_t x = 0;
        for (auto const& v : vdev_)
            x += v.budget > v.used ? v.budget - v.used : 0;
        return x;
    }
    static void free_parts(SharedFilter& sf)
    {
        for (auto& copy : sf.copies)
            for (auto& part : copy)
                if (part.f)
                    gn_filter_free(part.f);
        sf.copies.clear();
    }

    std::vector<int>          entries_, uniq_;
    std::vector<VDev>         vdev_;
    bool                      virtual_ = false;
    std::vector<SharedFilter> filters_;
    Stage                     stage_[2];
    std::string               log_;
};

} // namespace gnhost
