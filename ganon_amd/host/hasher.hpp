// hasher.hpp -- the distinct minimiser hashes of whole sequences, computed on the device (count_hashes,
// /root/reference/src/ganon-build/GanonBuild.cpp:184-249).  Shared by ganon-build (which inserts them) and
// `ganon-classify --verify-filter` (which looks them up again, tests/ganon-build/GanonBuild.test.cpp:53-98).
#pragma once

#include "ganon_hip.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gnhost
{

inline std::string hip_error()
{
    return gn_last_error();
}

// One parser thread's device side: a stream on a placeholder filter (hashing does not look at the filter)
class Hasher
{
public:
    Hasher(int device, uint32_t k, uint32_t w) : k_(k), w_(w)
    {
        // pieces of `stride_` window starts; short enough for the lane-per-read minimiser kernel where it applies
        stride_ = w <= 128 ? 512 : 4096;
        gn_ibf_desc d{};
        d.bins = 64, d.bin_words = 1, d.bin_size = 64, d.hash_shift = 57, d.hash_funs = 1, d.rows = nullptr;
        std::vector<uint32_t> identity(64);
        for (uint32_t b = 0; b < 64; ++b)
            identity[b] = b;
        if (gn_filter_upload_ibf(device, &d, identity.data(), 64, &flt_) != GN_OK)
            throw std::runtime_error(hip_error());
        if (gn_stream_create(flt_, kMaxPieces, kMaxBases, 1, &st_) != GN_OK)
            throw std::runtime_error(hip_error());
        void* p = nullptr;
        if (gn_pinned_alloc(kMaxBases, &p) != GN_OK)
            throw std::runtime_error(hip_error());
        bases_ = static_cast<uint8_t*>(p);
        off_.reserve(kMaxPieces + 1);
        off_.push_back(0);
    }
    ~Hasher()
    {
        if (st_)
            gn_stream_destroy(st_);
        if (flt_)
            gn_filter_free(flt_);
        if (bases_)
            gn_pinned_free(bases_);
    }

    // add one sequence; `out` receives the distinct hashes of everything flushed so far for the current file
    void add(const uint8_t* seq, uint64_t len, std::vector<uint64_t>& out, unsigned& flushes)
    {
        if (len < k_)
            return;
        if (len < w_)
        {
            short_[(uint32_t)len].append(reinterpret_cast<const char*>(seq), len);
            return;
        }
        for (uint64_t at = 0; at + w_ <= len; at += stride_)
        {
            const uint64_t n = std::min<uint64_t>(len - at, (uint64_t)stride_ + w_ - 1);
            if (fill_ + n > kMaxBases || off_.size() > kMaxPieces)
                flush(out, flushes);
            std::memcpy(bases_ + fill_, seq + at, n);
            fill_ += n;
            off_.push_back(fill_);
        }
    }

    void flush(std::vector<uint64_t>& out, unsigned& flushes)
    {
        if (off_.size() > 1)
        {
            run(w_, out);
            ++flushes;
        }
        fill_ = 0;
        off_.assign(1, 0);
    }

    // sequences shorter than the window: one launch per length, the window being the whole sequence
    void flush_short(std::vector<uint64_t>& out, unsigned& flushes)
    {
        for (auto& [len, cat] : short_)
        {
            for (size_t at = 0; at < cat.size();)
            {
                const size_t n_seq = std::min<size_t>((cat.size() - at) / len, std::min<size_t>(kMaxPieces, kMaxBases / len));
                std::memcpy(bases_, cat.data() + at, n_seq * len);
                off_.assign(1, 0);
                for (size_t i = 1; i <= n_seq; ++i)
                    off_.push_back(i * len);
                fill_ = n_seq * len;
                run(len, out);
                ++flushes;
                at += n_seq * len;
            }
        }
        short_.clear();
        fill_ = 0;
        off_.assign(1, 0);
    }

private:
    // one device batch: enough for a bacterial genome in one go; every parser thread has its own (page-locked) copy, so the
    // size is also what start-up pays per thread
    static constexpr uint64_t kMaxBases  = 64ull << 20;
    static constexpr uint32_t kMaxPieces = 1u << 18;

    void run(uint32_t w, std::vector<uint64_t>& out)
    {
        const uint32_t n = (uint32_t)off_.size() - 1;
        if (gn_stream_upload_reads(st_, bases_, fill_, off_.data(), nullptr, n) != GN_OK || gn_stream_minimisers(st_, k_, w) != GN_OK)
            throw std::runtime_error(hip_error());
        uint64_t nd = 0;
        if (gn_stream_distinct_hashes(st_, nullptr, 0, &nd) != GN_OK)
            throw std::runtime_error(hip_error());
        const size_t at = out.size();
        out.resize(at + nd);
        if (nd && gn_stream_distinct_hashes(st_, out.data() + at, nd, &nd) != GN_OK)
            throw std::runtime_error(hip_error());
    }

    uint32_t                        k_, w_, stride_;
    gn_filter*                      flt_ = nullptr;
    gn_stream*                      st_  = nullptr;
    uint8_t*                        bases_ = nullptr;
    uint64_t                        fill_  = 0;
    std::vector<uint64_t>           off_;
    std::map<uint32_t, std::string> short_;
};

} // namespace gnhost
