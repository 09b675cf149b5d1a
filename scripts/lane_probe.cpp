// lane_probe.cpp -- the device side of the product's pipeline alone: T host threads, L batch contexts (gn_stream) each, raw FASTQ text of
// 160 k reads per batch from page-locked memory against a 1 GiB filter, the calls and their order as in host/classify.cpp's worker loop
// (kernels of the batch uploaded last round; next upload; results of the oldest batch) -- no file reading, no post-processing, no output.
//   g++ -O2 -std=c++17 -pthread -I include -o scripts/lane_probe scripts/lane_probe.cpp -L ganon_amd/csrc -lganon_hip -Wl,-rpath,$PWD/ganon_amd/csrc
#include <ganon_hip.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { if ((x) != GN_OK) { fprintf(stderr, "%s: %s\n", #x, gn_last_error()); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const uint32_t n = 160000, L = 150; const uint64_t rows = 1ull << 21; const int batches = argc > 1 ? atoi(argv[1]) : 400;
    { int nd = 0; CK(gn_device_count(&nd)); } // (applies $GANON_HIP_SYNC)
    // $PROBE_BG=lock: a thread page-locks and frees 64 MiB blocks all the time; =copy: four threads copy 48 MiB blocks between ordinary buffers
    std::atomic<bool> bg_stop{ false }; std::vector<std::thread> bg;
    if (const char* e = getenv("PROBE_BG"))
    {
        if (std::string(e) == "lock")
            bg.emplace_back([&] { while (!bg_stop) { void* p = nullptr; if (gn_pinned_alloc(64u << 20, &p) == GN_OK) gn_pinned_free(p); } });
        else if (std::string(e) == "pread" || std::string(e) == "pread_plain")
        {
            // eight threads read a 2 GiB tmpfs file, 48 MiB at a time, into page-locked (or ordinary) buffers: what the product's slab readers do
            const bool pinned = std::string(e) == "pread";
            { FILE* fp = fopen("/dev/shm/lane_probe.bin", "wb"); std::vector<char> z(1 << 24, 'A'); for (int i = 0; i < 128; ++i) fwrite(z.data(), 1, z.size(), fp); fclose(fp); }
            for (int i = 0; i < 8; ++i)
                bg.emplace_back([&, i, pinned] {
                    const int fd = open("/dev/shm/lane_probe.bin", O_RDONLY); void* p = nullptr;
                    if (pinned) { if (gn_pinned_alloc(48u << 20, &p) != GN_OK) return; } else p = malloc(48u << 20);
                    uint64_t off = (uint64_t)i * (48u << 20);
                    while (!bg_stop) { size_t got = 0; while (got < (48u << 20)) { ssize_t k = pread(fd, (char*)p + got, (48u << 20) - got, off + got); if (k <= 0) break; got += k; } off = (off + 8ull * (48u << 20)) % (2000ull << 20); }
                    close(fd);
                });
        }
        else
            for (int i = 0; i < 4; ++i)
                bg.emplace_back([&] { std::vector<char> a(48u << 20, 1), b(48u << 20); while (!bg_stop) { memcpy(b.data(), a.data(), a.size()); a[0]++; } });
    }
    gn_ibf_desc d{ nullptr, rows, 64, 4096, 4, (uint32_t)__builtin_clzll(rows) };
    std::vector<uint32_t> b2t(4096); for (uint32_t i = 0; i < 4096; ++i) b2t[i] = i;
    gn_filter* f; CK(gn_filter_upload_ibf(0, &d, b2t.data(), 4096, &f)); CK(gn_filter_fill_random(f, 0, 42, 1, 0, 64)); CK(gn_filter_finalize(f));
    // one piece of text (every context uploads the same bytes from its own page-locked copy)
    std::string text; text.reserve((size_t)n * 320);
    uint64_t x = 88172645463325252ull; char id[16];
    for (uint32_t r = 0; r < n; ++r)
    {
        snprintf(id, sizeof id, "@r%09u\n", r); text += id;
        for (uint32_t j = 0; j < L; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; text += "ACGT"[(x >> 33) & 3]; }
        text += "\n+\n"; text.append(L, 'I'); text += '\n';
    }
    const char* one = getenv("PROBE_ONE"); // "TxL": just that configuration
    for (int T : { 2, 3 })
        for (int NL : { 1, 2 })
        if (!one || (one[0] - '0' == T && one[2] - '0' == NL))
        {
            std::atomic<int> next{ 0 }, ready{ 0 };
            std::vector<std::thread> th; double t0 = 0;
            std::atomic<uint64_t> reads{ 0 };
            for (int t = 0; t < T; ++t)
                th.emplace_back([&] {
                    struct Lane { gn_stream* s; uint8_t* txt; uint32_t *nh, *rec, *sq, *ln; uint8_t* st; uint64_t* mo; gn_match* m; int state = 0; uint64_t age = 0; };
                    std::vector<Lane> lanes(NL);
                    for (auto& ln : lanes)
                    {
                        CK(gn_stream_create(f, n + n / 4, text.size() + (1 << 20), 0, &ln.s));
                        void* p; CK(gn_pinned_alloc(text.size(), &p)); ln.txt = (uint8_t*)p; memcpy(ln.txt, text.data(), text.size());
                        CK(gn_pinned_alloc((size_t)n * 4, &p)); ln.nh = (uint32_t*)p; CK(gn_pinned_alloc((size_t)n * 4, &p)); ln.rec = (uint32_t*)p;
                        CK(gn_pinned_alloc((size_t)n * 4, &p)); ln.sq = (uint32_t*)p; CK(gn_pinned_alloc((size_t)n * 4, &p)); ln.ln = (uint32_t*)p;
                        CK(gn_pinned_alloc(n, &p)); ln.st = (uint8_t*)p; CK(gn_pinned_alloc(((size_t)n + 1) * 8, &p)); ln.mo = (uint64_t*)p;
                        CK(gn_pinned_alloc((size_t)n * 4 * sizeof(gn_match), &p)); ln.m = (gn_match*)p;
                    }
                    // warm-up batch per lane, then everybody starts together
                    for (auto& ln : lanes) { uint32_t k; uint64_t nb, pb, need; CK(gn_stream_upload_fastq(ln.s, ln.txt, text.size())); CK(gn_stream_fastq_index(ln.s, &k, &nb, &pb)); CK(gn_stream_classify(ln.s, 19, 31, 0.75)); CK(gn_fetch_batch(ln.s, ln.nh, ln.st, ln.mo, nullptr, 0, &need)); }
                    if (ready.fetch_add(1) + 1 == T) t0 = now();
                    while (ready.load() < T) {}
                    auto pick = [&](int st) -> Lane* { Lane* b = nullptr; for (auto& l : lanes) if (l.state == st && (!b || l.age < b->age)) b = &l; return b; };
                    uint64_t age = 0; bool more = true;
                    for (;;)
                    {
                        if (Lane* u = pick(1)) { uint32_t k; uint64_t nb, pb; CK(gn_stream_fastq_index(u->s, &k, &nb, &pb)); CK(gn_stream_classify(u->s, 19, 31, 0.75)); u->state = 2; }
                        bool took = false; Lane* xl = more ? pick(0) : nullptr;
                        if (xl) { if (next.fetch_add(1) < batches) { took = true; xl->age = age++; CK(gn_stream_upload_fastq(xl->s, xl->txt, text.size())); xl->state = 1; } else more = false; }
                        Lane* q = pick(2);
                        if (q && (!took || !pick(0)))
                        {
                            uint64_t need = 0; CK(gn_fetch_batch(q->s, q->nh, q->st, q->mo, nullptr, 0, &need)); CK(gn_fetch_batch(q->s, nullptr, nullptr, nullptr, q->m, (uint64_t)n * 4, &need));
                            CK(gn_stream_fastq_records(q->s, q->rec, q->sq, q->ln)); q->state = 0; reads += n;
                        }
                        else if (!took && !pick(1) && !q && !more) break;
                    }
                });
            for (auto& t : th) t.join();
            const double wall = now() - t0; // (includes the threads' tear-down of their contexts: a few ms)
            printf("{\"threads\": %d, \"lanes\": %d, \"batches\": %d, \"wall_s\": %.3f, \"ms_per_batch\": %.3f, \"mreads_per_s\": %.1f, \"text_GBps\": %.1f}\n", T, NL, batches, wall, wall / batches * 1e3,
                   reads.load() / wall / 1e6, (double)text.size() * batches / wall / 1e9);
            fflush(stdout);
        }
    bg_stop = true;
    for (auto& t : bg) t.join();
    gn_filter_free(f);
    return 0;
}
