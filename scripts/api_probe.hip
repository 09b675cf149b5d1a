// api_probe.hip -- what the runtime calls of one batch cost the calling thread: an asynchronous 48 MiB host-to-device copy from
// page-locked memory, small memsets, kernel launches, a small device-to-host copy; one thread, then three threads at once (one
// stream each), with and without a busy device.   hipcc --offload-arch=gfx950 -O2 -o scripts/api_probe scripts/api_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void tiny(unsigned long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void busy(const uint4* __restrict__ p, size_t n, unsigned long long* out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567) out[1] = acc;
}
struct Res { double copy = 0, memset3 = 0, launch6 = 0, d2h = 0, sync = 0, total = 0; };
static void worker(int iters, size_t bytes, bool with_busy, const uint4* big, size_t big_n, Res* r)
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    void* h; hipHostMalloc(&h, bytes, hipHostMallocDefault);
    void* d; hipMalloc(&d, bytes);
    unsigned long long *dc, *hc; hipMalloc(&dc, 64); hipHostMalloc((void**)&hc, 64, hipHostMallocDefault); hipMemset(dc, 0, 64);
    for (int it = -3; it < iters; ++it)
    {
        Res x; const double t0 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); const double t1 = now();
        for (int i = 0; i < 3; ++i) hipMemsetAsync(dc + i, 0, 8, st); const double t2 = now();
        for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, dc);
        if (with_busy) hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, st, big, big_n, dc);
        const double t3 = now();
        hipMemcpyAsync(hc, dc, 64, hipMemcpyDeviceToHost, st); const double t4 = now();
        hipStreamSynchronize(st); const double t5 = now();
        if (it >= 0) { r->copy += t1 - t0; r->memset3 += t2 - t1; r->launch6 += t3 - t2; r->d2h += t4 - t3; r->sync += t5 - t4; r->total += t5 - t0; }
    }
    hipFree(d); hipHostFree(h); hipFree(dc); hipHostFree(hc); hipStreamDestroy(st);
}
int main()
{
    const size_t bytes = 48u << 20; const int iters = 60;
    const char* mode = getenv("PROBE_SYNC");
    if (mode && mode[0] == 'b') hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    const size_t big_bytes = 2ull << 30; uint4* big; hipMalloc(&big, big_bytes); hipMemset(big, 1, big_bytes);
    for (int with_busy = 0; with_busy < 2; ++with_busy)
        for (int nt : { 1, 3, 6 })
        {
            std::vector<Res> res(nt); std::vector<std::thread> th; const double t0 = now();
            for (int i = 0; i < nt; ++i) th.emplace_back(worker, iters, bytes, with_busy != 0, big, big_bytes / 16, &res[i]);
            for (auto& t : th) t.join();
            const double wall = now() - t0; Res s;
            for (auto& r : res) { s.copy += r.copy; s.memset3 += r.memset3; s.launch6 += r.launch6; s.d2h += r.d2h; s.sync += r.sync; s.total += r.total; }
            const double k = 1e3 / (iters * nt);
            printf("{\"threads\": %d, \"busy_kernel\": %d, \"per_batch_ms\": {\"h2d_48MiB_call\": %.3f, \"3_memsets\": %.3f, \"6_launches\": %.3f, \"d2h_call\": %.3f, \"sync_wait\": %.3f, \"total\": %.3f}, \"aggregate_GBps\": %.1f, \"wall_s\": %.3f}\n",
                   nt, with_busy, s.copy * k, s.memset3 * k, s.launch6 * k, s.d2h * k, s.sync * k, s.total * k, (double)bytes * iters * nt / wall / 1e9, wall);
        }
    return 0;
}
