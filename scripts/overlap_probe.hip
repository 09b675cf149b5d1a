// overlap_probe.hip -- do host-to-device copies of a read batch overlap a chip-filling HBM-bound kernel on this box?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/overlap_probe scripts/overlap_probe.hip ; scripts/overlap_probe
// Prints: H2D alone (150 MB from page-locked memory), the kernel alone (a persistent grid streaming a 16 GiB buffer: the shape of
// the count kernels), both at once on two streams, and a D2H of 20 MB beside them.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ p, size_t n, uint64_t* sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        sink[0] = acc.x;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t batch = 150u << 20, big = 16ull << 30, back = 20u << 20;
    void *h1, *h2, *hb;
    CK(hipHostMalloc(&h1, batch, hipHostMallocDefault));
    CK(hipHostMalloc(&h2, batch, hipHostMallocDefault));
    CK(hipHostMalloc(&hb, back, hipHostMallocDefault));
    memset(h1, 1, batch); memset(h2, 2, batch);
    void *d1, *d2, *dbig, *dsink, *dback;
    CK(hipMalloc(&d1, batch)); CK(hipMalloc(&d2, batch)); CK(hipMalloc(&dbig, big)); CK(hipMalloc(&dsink, 64)); CK(hipMalloc(&dback, back));
    CK(hipMemset(dbig, 1, big));
    hipStream_t sa, sb, sc;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const unsigned grid = prop.multiProcessorCount * 8;
    auto kernel = [&](hipStream_t s, int reps) { for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)dbig, big / 16, (uint64_t*)dsink); };
    // warm up
    CK(hipMemcpyAsync(d1, h1, batch, hipMemcpyHostToDevice, sa)); kernel(sb, 1); CK(hipDeviceSynchronize());
    const int R = 10;
    double t0 = now();
    for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(i & 1 ? d2 : d1, i & 1 ? h2 : h1, batch, hipMemcpyHostToDevice, sa));
    CK(hipStreamSynchronize(sa));
    const double t_copy = (now() - t0) / R;
    t0 = now(); kernel(sb, R); CK(hipStreamSynchronize(sb));
    const double t_kernel = (now() - t0) / R;
    t0 = now();
    for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(i & 1 ? d2 : d1, i & 1 ? h2 : h1, batch, hipMemcpyHostToDevice, sa));
    kernel(sb, R);
    CK(hipStreamSynchronize(sa)); const double t_copy_beside = (now() - t0) / R;
    CK(hipStreamSynchronize(sb)); const double t_both = (now() - t0) / R;
    t0 = now();
    for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(i & 1 ? d2 : d1, i & 1 ? h2 : h1, batch, hipMemcpyHostToDevice, sa)); CK(hipMemcpyAsync(hb, dback, back, hipMemcpyDeviceToHost, sc)); }
    kernel(sb, R);
    CK(hipDeviceSynchronize());
    const double t_three = (now() - t0) / R;
    // two uploads at once on two streams (two workers uploading): does the link rate hold?
    t0 = now();
    for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(d1, h1, batch, hipMemcpyHostToDevice, sa)); CK(hipMemcpyAsync(d2, h2, batch, hipMemcpyHostToDevice, sc)); }
    CK(hipDeviceSynchronize());
    const double t_two_copies = (now() - t0) / (2 * R);
    printf("{\"h2d_150mb_ms\": %.3f, \"h2d_gbs\": %.1f, \"kernel_16gib_ms\": %.3f, \"kernel_gbs\": %.0f, \"h2d_beside_kernel_ms\": %.3f, "
           "\"both_per_pair_ms\": %.3f, \"serial_would_be_ms\": %.3f, \"with_d2h_20mb_ms\": %.3f, \"two_uploads_at_once_ms_each\": %.3f}\n",
           t_copy * 1e3, batch / t_copy / 1e9, t_kernel * 1e3, big / t_kernel / 1e9, t_copy_beside * 1e3, t_both * 1e3, (t_copy + t_kernel) * 1e3,
           t_three * 1e3, t_two_copies * 1e3);
    return 0;
}
