// robin_order.hpp -- iteration order of the reference's hash maps, for `--reference-order`.
//
// ganon-classify writes `.all` lines in the order its per-read map of matches iterates (TMatches,
// /root/reference/src/ganon-classify/GanonClassify.cpp:53,583) and `.rep` rows in the order its report map iterates (TRep,
// :180,836, after sum_reports :475-490), and select_matches fills TMatches in the order the filter's target map iterates
// (TMap, :55,516,556).  All three are robin_hood::unordered_map, whose iteration order is its slot order: a function of the
// key's hash, the order of insertion and the table's growth history.  The library is an un-vendored submodule of the
// reference (libs/robin-hood-hashing, no pinned revision visible in the tree), so what follows restates robin_hood.h of the
// 3.11 series from its published source -- Table::keyToIdx, insertKeyPrepareEmptySpot, shiftUp, try_increase_info,
// increase_size / rehashPowerOfTwo / insert_move, hash_bytes -- and libstdc++'s std::hash<std::string> (_Hash_bytes, the
// Murmur-2 64-bit variant) that TRep's PairHash uses (:59-67).  NOT verified against a run of the reference (none can be
// built here); tests pin the hashes against independent implementations and the table against hand-worked sequences.
//
// Only the ORDER is simulated: keys are the caller's 32-bit ids with their 64-bit hash; no values are stored.
#pragma once

#include <cstdint>
#include <cstring>
#include <string_view>
#include <vector>

namespace gnhost
{

// robin_hood::hash_bytes (what robin_hood::hash<std::string> calls): Murmur-2-64 body, seed 0xe17a1465, without the final
// multiply (keyToIdx does its own mixing)
inline uint64_t rh_hash_bytes(const void* ptr, size_t len)
{
    constexpr uint64_t m = 0xc6a4a7935bd1e995ull, seed = 0xe17a1465ull;
    constexpr unsigned r = 47;
    const auto*        p = static_cast<const unsigned char*>(ptr);
    uint64_t           h = seed ^ (len * m);
    const size_t       n_blocks = len / 8;
    for (size_t i = 0; i < n_blocks; ++i)
    {
        uint64_t k;
        std::memcpy(&k, p + 8 * i, 8);
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    const unsigned char* tail = p + 8 * n_blocks;
    switch (len & 7u)
    {
        case 7: h ^= (uint64_t)tail[6] << 48; [[fallthrough]];
        case 6: h ^= (uint64_t)tail[5] << 40; [[fallthrough]];
        case 5: h ^= (uint64_t)tail[4] << 32; [[fallthrough]];
        case 4: h ^= (uint64_t)tail[3] << 24; [[fallthrough]];
        case 3: h ^= (uint64_t)tail[2] << 16; [[fallthrough]];
        case 2: h ^= (uint64_t)tail[1] << 8; [[fallthrough]];
        case 1:
            h ^= (uint64_t)tail[0];
            h *= m;
            break;
        default: break;
    }
    h ^= h >> r;
    return h;
}
inline uint64_t rh_hash(std::string_view s) { return rh_hash_bytes(s.data(), s.size()); }

// libstdc++ std::hash<std::string> on 64-bit targets: std::_Hash_bytes(ptr, len, 0xc70f6907)
inline uint64_t std_hash_bytes(const void* ptr, size_t len, uint64_t seed = 0xc70f6907ull)
{
    constexpr uint64_t mul = (0xc6a4a793ull << 32) + 0x5bd1e995ull;
    auto               shift_mix = [](uint64_t v) { return v ^ (v >> 47); };
    const auto*        p   = static_cast<const unsigned char*>(ptr);
    const size_t       aligned = len & ~(size_t)7;
    uint64_t           hash = seed ^ (len * mul);
    for (size_t i = 0; i < aligned; i += 8)
    {
        uint64_t k;
        std::memcpy(&k, p + i, 8);
        const uint64_t data = shift_mix(k * mul) * mul;
        hash ^= data;
        hash *= mul;
    }
    if (len & 7)
    {
        uint64_t data = 0;
        for (int n = (int)(len & 7) - 1; n >= 0; --n)
            data = (data << 8) + p[aligned + n];
        hash ^= data;
        hash *= mul;
    }
    hash = shift_mix(hash) * mul;
    hash = shift_mix(hash);
    return hash;
}
// PairHash (GanonClassify.cpp:59-67)
inline uint64_t pair_hash(std::string_view first, std::string_view second)
{
    return std_hash_bytes(first.data(), first.size()) ^ (std_hash_bytes(second.data(), second.size()) << 1);
}

// Slot bookkeeping of robin_hood::detail::Table (MaxLoadFactor100 = 80): which slot every key ends up in.
class RobinSlots
{
public:
    void clear()
    {
        info_.clear();
        slot_id_.clear();
        slot_hash_.clear();
        n_ = 0;
        mask_ = 0;
        max_allowed_ = 0;
        info_inc_ = kInitialInfoInc;
        info_hash_shift_ = 0;
        multiplier_ = 0xc4ceb9fe1a85ec53ull;
    }
    size_t size() const { return n_; }

    // operator[] / emplace of a key the caller knows by `id` (equal ids = equal keys).  Returns true if it was new.
    bool insert(uint32_t id, uint64_t hash)
    {
        for (int attempt = 0; attempt < 256; ++attempt)
        {
            size_t   idx  = 0;
            uint32_t info = 0;
            if (mask_ == 0 && info_.empty())
            {
                increase_size();
                continue;
            }
            key_to_idx(hash, idx, info);
            while (info < info_[idx])
                next(info, idx);
            while (info == info_[idx])
            {
                if (slot_id_[idx] == id)
                    return false;
                next(info, idx);
            }
            if (n_ >= max_allowed_)
            {
                increase_size();
                continue;
            }
            const size_t   insertion_idx  = idx;
            const uint32_t insertion_info = info;
            if (insertion_info + info_inc_ > 0xFF)
                max_allowed_ = 0;
            while (info_[idx] != 0)
                next(info, idx);
            if (idx != insertion_idx)
                shift_up(idx, insertion_idx);
            slot_id_[insertion_idx]   = id;
            slot_hash_[insertion_idx] = hash;
            info_[insertion_idx]      = (uint8_t)insertion_info;
            ++n_;
            return true;
        }
        return false; // (robin_hood throws an overflow error here)
    }

    // ids in iteration order (begin() .. end(): ascending slot)
    void order(std::vector<uint32_t>& out) const
    {
        out.clear();
        const size_t total = info_.empty() ? 0 : with_buffer(mask_ + 1);
        for (size_t i = 0; i < total; ++i)
            if (info_[i] != 0)
                out.push_back(slot_id_[i]);
    }

private:
    static constexpr uint32_t kInitialInfoNumBits = 5;
    static constexpr uint32_t kInitialInfoInc     = 1u << kInitialInfoNumBits;
    static constexpr uint64_t kInfoMask           = kInitialInfoInc - 1;

    static size_t calc_max_allowed(size_t max_elements) { return max_elements * 80 / 100; }
    static size_t with_buffer(size_t n)
    {
        const size_t a = calc_max_allowed(n);
        return n + (a < 0xFF ? a : (size_t)0xFF);
    }
    void key_to_idx(uint64_t h, size_t& idx, uint32_t& info) const
    {
        h *= multiplier_;
        h ^= h >> 33;
        info = info_inc_ + (uint32_t)((h & kInfoMask) >> info_hash_shift_);
        idx  = (size_t)(h >> kInitialInfoNumBits) & mask_;
    }
    void next(uint32_t& info, size_t& idx) const
    {
        ++idx;
        info += info_inc_;
    }
    void shift_up(size_t start, size_t insertion_idx)
    {
        size_t idx = start;
        while (idx != insertion_idx)
        {
            slot_id_[idx]   = slot_id_[idx - 1];
            slot_hash_[idx] = slot_hash_[idx - 1];
            --idx;
        }
        idx = start;
        while (idx != insertion_idx)
        {
            info_[idx] = (uint8_t)(info_[idx - 1] + info_inc_);
            if ((uint32_t)info_[idx] + info_inc_ > 0xFF)
                max_allowed_ = 0;
            --idx;
        }
    }
    void init_data(size_t max_elements)
    {
        n_           = 0;
        mask_        = max_elements - 1;
        max_allowed_ = calc_max_allowed(max_elements);
        const size_t total = with_buffer(max_elements);
        info_.assign(total + 8, 0); // (+ the sentinel and the 8-byte overread of the real layout)
        info_[total] = 1;
        slot_id_.assign(total, 0);
        slot_hash_.assign(total, 0);
        info_inc_        = kInitialInfoInc;
        info_hash_shift_ = 0;
    }
    bool try_increase_info()
    {
        if (info_inc_ <= 2)
            return false;
        info_inc_ = (uint8_t)(info_inc_ >> 1);
        ++info_hash_shift_;
        const size_t total = with_buffer(mask_ + 1);
        for (size_t i = 0; i < total; ++i) // (the real code shifts 8 info bytes at a time; per byte it is the same)
            info_[i] = (uint8_t)((info_[i] >> 1) & 0x7f);
        info_[total] = 1;
        max_allowed_ = calc_max_allowed(mask_ + 1);
        return true;
    }
    void increase_size()
    {
        if (mask_ == 0 && info_.empty())
        {
            init_data(8);
            return;
        }
        const size_t max_now = calc_max_allowed(mask_ + 1);
        if (n_ < max_now && try_increase_info())
            return;
        if (n_ * 2 < calc_max_allowed(mask_ + 1))
        {
            multiplier_ += 0xc4ceb9fe1a85ec54ull; // nextHashMultiplier: same size, another mixing constant
            rehash(mask_ + 1);
        }
        else
            rehash((mask_ + 1) * 2);
    }
    void rehash(size_t buckets)
    {
        std::vector<uint8_t>  old_info;
        std::vector<uint32_t> old_id;
        std::vector<uint64_t> old_hash;
        old_info.swap(info_);
        old_id.swap(slot_id_);
        old_hash.swap(slot_hash_);
        const size_t old_total = old_id.size();
        init_data(buckets);
        for (size_t i = 0; i < old_total; ++i)
            if (old_info[i] != 0)
                insert_move(old_id[i], old_hash[i]);
    }
    void insert_move(uint32_t id, uint64_t hash)
    {
        if (max_allowed_ == 0 && !try_increase_info())
            return; // (overflow error in the real thing)
        size_t   idx  = 0;
        uint32_t info = 0;
        key_to_idx(hash, idx, info);
        while (info <= info_[idx])
        {
            ++idx;
            info += info_inc_;
        }
        const size_t  insertion_idx  = idx;
        const uint8_t insertion_info = (uint8_t)info;
        if ((uint32_t)insertion_info + info_inc_ > 0xFF)
            max_allowed_ = 0;
        while (info_[idx] != 0)
            next(info, idx);
        if (idx != insertion_idx)
            shift_up(idx, insertion_idx);
        slot_id_[insertion_idx]   = id;
        slot_hash_[insertion_idx] = hash;
        info_[insertion_idx]      = insertion_info;
        ++n_;
    }

    std::vector<uint8_t>  info_;
    std::vector<uint32_t> slot_id_;
    std::vector<uint64_t> slot_hash_;
    size_t                n_ = 0, mask_ = 0, max_allowed_ = 0;
    uint32_t              info_inc_ = kInitialInfoInc, info_hash_shift_ = 0;
    uint64_t              multiplier_ = 0xc4ceb9fe1a85ec53ull;
};

} // namespace gnhost
