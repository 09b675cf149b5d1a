// pinned_probe.cpp -- how fast does a CPU thread read / write page-locked (gn_pinned_alloc) memory compared with malloc'ed
// memory on this host?  (decides which host buffers of the pipeline may be page-locked)   build: see scripts/pinned_probe.sh
#include "ganon_hip.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main()
{
    const size_t n = 512u << 20;
    void*        p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (gn_pinned_alloc(n, &p) != GN_OK) { std::printf("no device: %s\n", gn_last_error()); return 1; }
    double alloc_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    char* m = static_cast<char*>(std::malloc(n));
    char* d = static_cast<char*>(std::malloc(n));
    std::memset(m, 1, n); std::memset(d, 2, n); std::memset(p, 3, n);
    auto bw = [&](void* dst, const void* src) {
        double best = 0;
        for (int r = 0; r < 3; ++r) {
            auto a = std::chrono::steady_clock::now();
            std::memcpy(dst, src, n);
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
            best = std::max(best, n / s / 1e9);
        }
        return best;
    };
    std::printf("{\"pinned_alloc_512MiB_s\": %.3f, \"memcpy_GBps\": {\"malloc_to_malloc\": %.2f, \"pinned_to_malloc\": %.2f, \"malloc_to_pinned\": %.2f}",
                alloc_s, bw(d, m), bw(d, p), bw(p, m));
    unsigned long long sum = 0;
    auto a = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i += 64) sum += static_cast<unsigned char*>(p)[i];
    double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    a = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i += 64) sum += (unsigned char)m[i];
    double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    std::printf(", \"stride64_read_s\": {\"pinned\": %.3f, \"malloc\": %.3f}, \"checksum\": %llu}\n", s1, s2, sum);
    gn_pinned_free(p);
    return 0;
}
