// pinned_pool.hpp -- the page-locked host memory of the HIP backend (backend_hip.cpp): blocks by size class, locked once, handed
// out again.  hostmem.hpp's ByteBuf / U32Buf / U64Buf allocate through it once make_backends has pointed g_host_arena here.
#pragma once

#include <ganon_hip.h>

#include "startup.hpp"
#include "tunables.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace gnhost
{

// Page-locked blocks for the read batches, kept for the life of the process.  Locking pages costs ~0.26 s per GiB
// (profiles/r02_pinned_probe.json) and a pipeline's worth of batch buffers is half a GiB or more, so (a) a block, once
// locked, is handed out again instead of being unlocked, and (b) a thread starts locking the first blocks while the filters
// are still being loaded.  Sizes are rounded up (size_class) so that freed blocks fit later requests.
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
        void*      p  = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (gn_pinned_alloc(cls, &p) != GN_OK)
            return std::malloc(n ? n : 1); // (no more lockable memory: an ordinary buffer still works, its copies are just staged)
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(m_);
        size_of_[p] = cls;
        (running_ ? late_ : early_).add(cls, sec);
        return p;
    }
    // from here on batches are on the device: a block locked now stalls them ($GANON_HOST_TIMING reports how many were)
    void mark_running() { running_ = true; }
    std::string tally()
    {
        std::lock_guard<std::mutex> lk(m_);
        std::ostringstream os;
        os << "page-locked on demand before the first batch: " << early_.n << " blocks, " << (early_.bytes >> 20) << " MiB, " << early_.sec
           << " s; after it: " << late_.n << " blocks, " << (late_.bytes >> 20) << " MiB, " << late_.sec << " s; blocks per size (MiB: all / free now)";
        std::map<size_t, size_t> all;
        for (auto const& kv : size_of_)
            all[kv.second]++;
        for (auto const& kv : all)
        {
            auto it = free_.find(kv.first);
            os << ' ' << (kv.first >> 20) << ": " << kv.second << " / " << (it == free_.end() ? 0 : it->second.size());
        }
        return os.str();
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
    // block_bytes: what the reader will ask for (32 MiB holds the bases of a parsed 48 MiB slab; a slab that travels as text needs 56)
    void warm_up(size_t block_bytes = 32u << 20)
    {
        const size_t total = tun().size(Knob::prelock_mib, 512) << 20, block = size_class(block_bytes);
        if (total == 0)
            return;
        warm_ = std::thread([this, total, block] {
            const double t0 = StartupLog::now();
            size_t       done = 0;
            for (; done < total && !stop_; done += block)
            {
                void* p = nullptr;
                if (gn_pinned_alloc(block, &p) != GN_OK)
                    break;
                std::lock_guard<std::mutex> lk(m_);
                size_of_[p] = block;
                free_[block].push_back(p);
            }
            StartupLog::get().span("page-locking the batch pool (background thread)", t0, std::to_string(done >> 20) + " MiB in " + std::to_string(block >> 20) + " MiB blocks");
        });
    }
    // `count` more blocks that hold n bytes each go into the pool (locked now, handed out later)
    void reserve(size_t n, size_t count)
    {
        const size_t cls = size_class(n);
        for (size_t i = 0; i < count; ++i)
        {
            void* p = nullptr;
            if (gn_pinned_alloc(cls, &p) != GN_OK)
                return;
            std::lock_guard<std::mutex> lk(m_);
            size_of_[p] = cls;
            free_[cls].push_back(p);
        }
    }
    // blocks that hold n bytes: at least `count` of them are free in the pool right now (locks only what is short)
    void ensure_free(size_t n, size_t count)
    {
        const size_t cls = size_class(n);
        size_t       have;
        {
            std::lock_guard<std::mutex> lk(m_);
            have = free_[cls].size();
        }
        if (have < count)
            reserve(n, count - have);
    }
    void settle() // before the process lets go of the device: the warm-up thread is not in the middle of a call
    {
        stop_ = true;
        if (warm_.joinable())
            warm_.join();
    }

private:
    // powers of two up to 8 MiB, multiples of 8 MiB above (a 51 MiB piece of FASTQ text takes 56 MiB, not 64: locking is what costs)
    static size_t size_class(size_t n)
    {
        size_t c = 1u << 20;
        while (c < n && c < (8u << 20))
            c <<= 1;
        if (c >= n)
            return c;
        return (n + (8u << 20) - 1) / (8u << 20) * (8u << 20);
    }
    struct Tally
    {
        size_t n = 0, bytes = 0;
        double sec = 0;
        void   add(size_t b, double s)
        {
            ++n;
            bytes += b;
            sec += s;
        }
    };
    Tally                                 early_, late_;
    std::atomic<bool>                     running_{ false };
    std::mutex                            m_;
    std::map<size_t, std::vector<void*>>  free_;
    std::unordered_map<void*, size_t>     size_of_;
    std::thread                           warm_;
    std::atomic<bool>                     stop_{ false };
};

} // namespace gnhost
