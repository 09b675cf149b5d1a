// verify.cpp -- `ganon-classify --ibf F --verify-filter refs.tsv`: the membership check of the reference's own build test
// (validate_elements, /root/reference/tests/ganon-build/GanonBuild.test.cpp:53-98) against a filter FILE, on the device.
//
// refs.tsv has ganon-build's input format (`file [<tab> target]`, GanonBuild.cpp:88-140).  For every line the file's sequences
// are hashed on the device with the k / w the filter's IBFConfig states (canonical (k,w)-minimisers, raptor::adjust_seed), and
// every distinct hash has to be found -- all h bits set -- in at least one technical bin the file's bin_map gives the target
// (gn_filter_probe).  A filter that was built from these sequences passes whoever wrote it; one read with the wrong row, bin,
// seed or shift arithmetic fails at once, and the first false negative is printed with its hash, its h rows and the bits found
// there.  This is the first thing to run on a filter written by the reference's ganon-build (scripts/first_contact.sh).
#include "config.hpp"
#include "filter_io.hpp"
#include "hasher.hpp"
#include "hostmem.hpp"
#include "seq_io.hpp"

#include "ganon_hip.h"
#include "ganon_ibf_hash.h"

#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

namespace gnhost
{

namespace
{

// the filter's bits into HBM of one device through the streaming loader (what DeviceSet does for a replica, placement.hpp)
class OneDeviceSink final : public FilterSink
{
public:
    explicit OneDeviceSink(int device) : device_(device) {}
    ~OneDeviceSink() override
    {
        if (f_)
            gn_filter_free(f_);
        for (auto& s : stage_)
            if (s.ptr)
                gn_pinned_free(s.ptr);
    }
    bool begin(const FilterMeta& f, std::string& err) override
    {
        const IbfShape&       m = f.shapes.at(0);
        std::vector<uint32_t> bin2target(m.bins, 0xFFFFFFFFu);
        for (size_t t = 0; t < f.targets.size(); ++t)
            for (uint64_t b : f.target_bins[t])
                bin2target[b] = (uint32_t)t;
        gn_ibf_desc d{ nullptr, m.bin_size, m.bin_words, m.bins, (uint32_t)m.hash_funs, (uint32_t)m.hash_shift };
        words_ = m.bin_words;
        if (gn_filter_upload_ibf(device_, &d, bin2target.data(), (uint32_t)f.targets.size(), &f_) != GN_OK)
        {
            err = gn_last_error();
            return false;
        }
        return true;
    }
    uint64_t* staging(int which, size_t bytes) override
    {
        Stage& s = stage_[which & 1];
        if (s.bytes < bytes)
        {
            if (s.ptr)
                gn_pinned_free(s.ptr);
            s = Stage{};
            void* p = nullptr;
            if (gn_pinned_alloc(bytes, &p) != GN_OK)
                return nullptr;
            s.ptr = p, s.bytes = bytes;
        }
        return static_cast<uint64_t*>(s.ptr);
    }
    bool rows(uint32_t, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err) override
    {
        if (gn_filter_write_rows(f_, 0, row_begin, n_rows, src, words_, 0) == GN_OK)
            return true;
        err = gn_last_error();
        return false;
    }
    bool drain(std::string& err) override
    {
        if (gn_filter_write_sync(f_) == GN_OK)
            return true;
        err = gn_last_error();
        return false;
    }
    bool end(std::string& err) override
    {
        if (gn_filter_finalize(f_) == GN_OK)
            return true;
        err = gn_last_error();
        return false;
    }
    gn_filter* filter() const { return f_; }

private:
    struct Stage
    {
        void*  ptr   = nullptr;
        size_t bytes = 0;
    };
    int        device_;
    uint64_t   words_ = 0;
    gn_filter* f_     = nullptr;
    Stage      stage_[2];
};

// seqan3::interleaved_bloom_filter::hash_and_fit, for the report of a false negative only
uint64_t ibf_row(uint64_t v, unsigned i, const IbfShape& m)
{
    static const uint64_t seeds[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST; // include/ganon_ibf_hash.h
    uint64_t              x = v * seeds[i];
    x ^= x >> m.hash_shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint64_t)(((unsigned __int128)x * m.bin_size) >> 64);
}

} // namespace

bool verify_filter(const Config& config)
{
    if (config.hibf)
    {
        std::cerr << "--verify-filter checks a flat .ibf (the reference's build test has no HIBF counterpart); run --inspect-filter --hibf "
                     "on the file instead"
                  << std::endl;
        return false;
    }
    if (config.ibf.size() != 1)
    {
        std::cerr << "--verify-filter needs exactly one --ibf file" << std::endl;
        return false;
    }
    const int device = config.devices.empty() ? 0 : config.devices.front();
    try
    {
        FilterMeta    meta;
        OneDeviceSink sink(device);
        load_filter_file(config.ibf[0], false, meta, sink);
        const IbfShape& m = meta.shapes.at(0);
        const uint32_t  k = meta.ibf_config.kmer_size, w = meta.ibf_config.window_size;
        std::map<std::string, size_t> index;
        for (size_t t = 0; t < meta.targets.size(); ++t)
            index[meta.targets[t]] = t;
        std::cout << "filter\t" << config.ibf[0] << "\tk=" << k << " w=" << w << " h=" << m.hash_funs << " bins=" << m.bins << " rows=" << m.bin_size
                  << " targets=" << meta.targets.size() << "\n";
        std::cout << "#target\tfile\tbins\tdistinct_hashes\thits\tmissing\tverdict\n";

        Hasher        hasher(device, k, w);
        std::ifstream in(config.verify_filter);
        if (!in)
        {
            std::cerr << "cannot open " << config.verify_filter << std::endl;
            return false;
        }
        std::string line, ids;
        ByteBuf     seq;
        uint64_t    n_lines = 0, n_bad = 0, n_hashes_total = 0;
        while (std::getline(in, line, '\n'))
        {
            if (line.empty())
                continue;
            std::vector<std::string> fields;
            std::istringstream       ls(line);
            std::string              f;
            while (std::getline(ls, f, '\t'))
                fields.push_back(f);
            if (fields.empty() || fields.size() > 2)
                continue;
            const std::string& file = fields[0];
            std::string        target = fields.size() == 2 ? fields[1] : file.substr(file.find_last_of('/') == std::string::npos ? 0 : file.find_last_of('/') + 1);
            ++n_lines;
            auto it = index.find(target);
            if (it == index.end())
            {
                std::cout << target << "\t" << file << "\t0\t0\t0\t0\tFAIL: the filter's bin_map has no such target\n";
                ++n_bad;
                continue;
            }
            std::vector<uint64_t> hashes;
            unsigned              flushes = 0;
            {
                SeqReader reader(file);
                for (;;)
                {
                    ids.clear();
                    seq.clear();
                    if (!reader.next(ids, seq))
                        break;
                    hasher.add(seq.data(), seq.size(), hashes, flushes);
                }
                hasher.flush(hashes, flushes);
                hasher.flush_short(hashes, flushes);
            }
            std::sort(hashes.begin(), hashes.end());
            hashes.erase(std::unique(hashes.begin(), hashes.end()), hashes.end());
            std::vector<uint32_t> bins;
            for (uint64_t b : meta.target_bins[it->second])
                bins.push_back((uint32_t)b);
            uint64_t hits = 0, missing = 0, first = 0;
            if (gn_filter_probe(sink.filter(), hashes.data(), hashes.size(), bins.data(), (uint32_t)bins.size(), &hits, &missing, &first) != GN_OK)
                throw std::runtime_error(hip_error());
            n_hashes_total += hashes.size();
            std::cout << target << "\t" << file << "\t" << bins.size() << "\t" << hashes.size() << "\t" << hits << "\t" << missing << "\t"
                      << (missing ? "FAIL" : "ok") << "\n";
            if (missing)
            {
                ++n_bad;
                // the first false negative: its hash, its h rows, and what the file holds there for every bin of the target
                const uint64_t        v = hashes[first];
                std::vector<uint64_t> rows(m.hash_funs), words(m.hash_funs * m.bin_words);
                for (unsigned i = 0; i < m.hash_funs; ++i)
                    rows[i] = ibf_row(v, i, m);
                if (gn_filter_download_row_list(sink.filter(), 0, rows.data(), rows.size(), words.data()) != GN_OK)
                    throw std::runtime_error(hip_error());
                std::cout << "  first false negative: hash " << v << " (index " << first << " of the sorted distinct hashes); rows";
                for (auto r : rows)
                    std::cout << " " << r;
                std::cout << "; bits [bin: one per hash function]";
                for (size_t b = 0; b < bins.size() && b < 8; ++b)
                {
                    std::cout << " [" << bins[b] << ":";
                    for (unsigned i = 0; i < m.hash_funs; ++i)
                        std::cout << " " << ((words[i * m.bin_words + (bins[b] >> 6)] >> (bins[b] & 63)) & 1);
                    std::cout << "]";
                }
                std::cout << "\n";
            }
        }
        std::cout << "result\t" << (n_bad ? "FAIL" : "ok") << "\t" << n_lines << " line(s), " << n_bad << " failing, " << n_hashes_total
                  << " distinct minimisers looked up\n";
        return n_bad == 0 && n_lines > 0;
    }
    catch (const std::exception& e)
    {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return false;
    }
}

} // namespace gnhost
