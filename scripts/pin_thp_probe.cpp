// pin_thp_probe.cpp -- what does page-locking cost, and is memory backed by transparent huge pages cheaper to lock?
//   A  hipHostMalloc(n)                                         (what gn_pinned_alloc did up to round 5)
//   B  mmap(n, 2 MiB aligned) + madvise(MADV_HUGEPAGE) + touch + hipHostRegister(n)
//   C  as B without the madvise (4 KiB pages)
// per block size: seconds to get a usable block, H2D copy rate from it, seconds to release it.
//   hipcc -O2 -o scripts/pin_thp_probe scripts/pin_thp_probe.cpp && scripts/pin_thp_probe     -> JSON lines
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sys/mman.h>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1)
    {
        std::printf("no device\n");
        return 1;
    }
    void* dev = nullptr;
    hipMalloc(&dev, 512u << 20);
    hipStream_t st;
    hipStreamCreate(&st);
    { void* w = nullptr; hipHostMalloc(&w, 1 << 20, hipHostMallocPortable); hipMemcpyAsync(dev, w, 1 << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); hipHostFree(w); } // warm
    for (size_t mib : { 56u, 512u })
        for (int rep = 0; rep < 2; ++rep)
            for (int mode = 0; mode < 3; ++mode)
            {
                const size_t n = mib << 20;
                void*        p = nullptr;
                void*        raw = nullptr;
                double       t0 = now(), t_touch = 0;
                hipError_t   e = hipSuccess;
                if (mode == 0)
                    e = hipHostMalloc(&p, n, hipHostMallocPortable);
                else
                {
                    raw = mmap(nullptr, n + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    p   = (void*)(((uintptr_t)raw + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
                    if (mode == 1)
                        madvise(p, n, MADV_HUGEPAGE);
                    const double a = now();
                    for (size_t i = 0; i < n; i += 4096)
                        ((volatile char*)p)[i] = 0;
                    t_touch = now() - a;
                    e       = hipHostRegister(p, n, hipHostRegisterPortable);
                }
                const double t_alloc = now() - t0;
                if (e != hipSuccess)
                {
                    std::printf("{\"mode\": %d, \"mib\": %zu, \"error\": \"%s\"}\n", mode, mib, hipGetErrorString(e));
                    continue;
                }
                std::memset(p, 1, n);
                double best = 0;
                for (int r = 0; r < 3; ++r)
                {
                    const double a = now();
                    hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, st);
                    hipStreamSynchronize(st);
                    const double s = now() - a;
                    if (n / s / 1e9 > best)
                        best = n / s / 1e9;
                }
                const double t1 = now();
                if (mode == 0)
                    hipHostFree(p);
                else
                {
                    hipHostUnregister(p);
                    munmap(raw, n + (2u << 20));
                }
                std::printf("{\"mode\": \"%s\", \"mib\": %zu, \"rep\": %d, \"alloc_s\": %.4f, \"touch_s\": %.4f, \"lock_GBps\": %.2f, \"h2d_GBps\": %.1f, \"free_s\": %.4f}\n",
                            mode == 0 ? "hipHostMalloc" : mode == 1 ? "mmap+THP+hipHostRegister" : "mmap+hipHostRegister", mib, rep, t_alloc, t_touch,
                            n / t_alloc / 1e9, best, now() - t1);
                std::fflush(stdout);
            }
    return 0;
}
