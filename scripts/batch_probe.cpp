// batch_probe.cpp -- where a device worker's time per batch goes: the product's calls (gn_submit_batch + gn_fetch_batch) on 1 M-read
// batches against a 1 GiB filter, with the per-read arrays in pageable memory (what backend_hip.cpp did up to round 3) or in
// page-locked memory, one stream or two streams taking turns.
//   g++ -O2 -std=c++17 -pthread -I include -o scripts/batch_probe scripts/batch_probe.cpp -L ganon_amd/csrc -lganon_hip -Wl,-rpath,$PWD/ganon_amd/csrc
#include <ganon_hip.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { if ((x) != GN_OK) { fprintf(stderr, "%s: %s\n", #x, gn_last_error()); exit(1); } } while (0)
struct Bufs { uint8_t* bases; uint64_t* off; uint32_t* nh; uint8_t* st; uint64_t* mo; gn_match* m; };
static void* alloc(bool pinned, size_t n) { void* p = nullptr; if (pinned) { CK(gn_pinned_alloc(n, &p)); } else p = malloc(n); memset(p, 0, n); return p; }
int main(int argc, char** argv)
{
    const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 1u << 20, L = 150; const uint64_t rows = 1ull << 21; const int batches = argc > 1 ? atoi(argv[1]) : 24;
    gn_ibf_desc d{ nullptr, rows, 64, 4096, 4, (uint32_t)__builtin_clzll(rows) };
    std::vector<uint32_t> b2t(4096); for (uint32_t i = 0; i < 4096; ++i) b2t[i] = i;
    gn_filter* f; CK(gn_filter_upload_ibf(0, &d, b2t.data(), 4096, &f)); CK(gn_filter_fill_random(f, 0, 42, 1, 0, 64)); CK(gn_filter_finalize(f));
    for (int mode = 0; mode < 4; ++mode)
    {
        const bool pin_small = mode & 1; const int n_streams = mode & 2 ? 2 : 1;
        std::vector<gn_stream*> s(n_streams); std::vector<Bufs> bf(n_streams);
        for (int i = 0; i < n_streams; ++i)
        {
            CK(gn_stream_create(f, n, (uint64_t)n * L, 0, &s[i]));
            bf[i].bases = (uint8_t*)alloc(true, (size_t)n * L);
            uint64_t x = 88172645463325252ull + i;
            for (size_t j = 0; j < (size_t)n * L; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; bf[i].bases[j] = "ACGT"[(x >> 33) & 3]; }
            bf[i].off = (uint64_t*)alloc(pin_small, ((size_t)n + 1) * 8); for (uint32_t r = 0; r <= n; ++r) bf[i].off[r] = (uint64_t)r * L;
            bf[i].nh = (uint32_t*)alloc(pin_small, (size_t)n * 4); bf[i].st = (uint8_t*)alloc(pin_small, n); bf[i].mo = (uint64_t*)alloc(pin_small, ((size_t)n + 1) * 8);
            bf[i].m = (gn_match*)alloc(pin_small, (size_t)n * 4 * sizeof(gn_match));
        }
        auto one = [&](int i) {
            CK(gn_submit_batch(s[i], bf[i].bases, (uint64_t)n * L, bf[i].off, nullptr, n, 19, 31, 0.75));
        };
        auto fetch = [&](int i) { uint64_t need = 0; CK(gn_fetch_batch(s[i], bf[i].nh, bf[i].st, bf[i].mo, nullptr, 0, &need)); CK(gn_fetch_batch(s[i], nullptr, nullptr, nullptr, bf[i].m, (uint64_t)n * 4, &need)); };
        for (int i = 0; i < n_streams; ++i) { one(i); fetch(i); } // warm up
        double t_sub = 0, t_fetch = 0; const double t0 = now();
        if (n_streams == 1)
            for (int b = 0; b < batches; ++b) { double a = now(); one(0); double c = now(); fetch(0); t_sub += c - a; t_fetch += now() - c; }
        else
        {
            std::vector<std::thread> th;
            for (int i = 0; i < n_streams; ++i) th.emplace_back([&, i] { for (int b = i; b < batches; b += n_streams) { one(i); fetch(i); } });
            for (auto& t : th) t.join();
        }
        const double wall = now() - t0;
        printf("{\"reads_per_batch\": %u, \"per_read_arrays\": \"%s\", \"streams\": %d, \"batches\": %d, \"ms_per_batch\": %.2f, \"mreads_per_s\": %.1f, \"submit_ms\": %.2f, \"fetch_ms\": %.2f}\n",
               n, pin_small ? "page-locked" : "pageable", n_streams, batches, wall / batches * 1e3, (double)n * batches / wall / 1e6, t_sub / batches * 1e3, t_fetch / batches * 1e3);
        for (int i = 0; i < n_streams; ++i) gn_stream_destroy(s[i]);
    }
    gn_filter_free(f);
    return 0;
}
