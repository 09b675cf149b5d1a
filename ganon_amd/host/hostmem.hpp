// hostmem.hpp -- where the bytes that go to the device live on the host.
// A read batch's bases are copied to the GPU once per batch; from ordinary (pageable) memory that copy is staged by the
// runtime through the CPU at a few GB/s and blocks the calling thread, from page-locked memory it is a DMA at PCIe speed
// that overlaps with everything else.  ByteBuf is a byte vector whose storage comes from a process-wide pair of
// functions; the HIP backend points them at gn_pinned_alloc / gn_pinned_free (backend_hip.cpp), everything else keeps
// malloc / free.  Buffers are recycled by the pipeline, so the (expensive) page-locking happens a handful of times.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

namespace gnhost
{

struct HostArena
{
    void* (*alloc)(size_t) = [](size_t n) -> void* { return std::malloc(n); };
    void (*release)(void*) = [](void* p) { std::free(p); };
};
inline HostArena g_host_arena;

template <typename T>
struct ArenaAllocator
{
    using value_type = T;
    ArenaAllocator() = default;
    template <typename U>
    ArenaAllocator(const ArenaAllocator<U>&) noexcept
    {
    }
    T* allocate(size_t n)
    {
        void* p = g_host_arena.alloc(n * sizeof(T));
        if (!p)
            throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) noexcept { g_host_arena.release(p); }
    // resize() leaves new elements as they are (these buffers are filled by pread / memcpy / the device right after)
    template <typename U, typename... Args>
    void construct(U* p, Args&&... args)
    {
        if constexpr (sizeof...(Args) == 0)
            ::new (static_cast<void*>(p)) U;
        else
            ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
    template <typename U>
    bool operator==(const ArenaAllocator<U>&) const noexcept
    {
        return true;
    }
    template <typename U>
    bool operator!=(const ArenaAllocator<U>&) const noexcept
    {
        return false;
    }
};

using ByteBuf = std::vector<uint8_t, ArenaAllocator<uint8_t>>;
using U32Buf  = std::vector<uint32_t, ArenaAllocator<uint32_t>>;
using U64Buf  = std::vector<uint64_t, ArenaAllocator<uint64_t>>; // per-read arrays the device writes (page-locked under the HIP backend)

} // namespace gnhost
