// calib_gather.hip -- what a random gather of SHORT rows can reach on MI355X, and what rocprofv3's FETCH_SIZE reports for it.
//
// The HIBF level kernels (ganon_amd/csrc/gn_hibf.hip) fetch 32-byte rows (256-bin IBFs), 8 bytes per lane, at random row
// indices -- level 0 from a table of tens of MiB (inside the 256 MiB Infinity Cache), the lower levels from GiBs.  The
// flat kernels fetch 512-byte and 4-KiB rows.  This program issues exactly such gathers from a table of a chosen size
// with a KNOWN number of row requests, so that
//   * the rate it reaches is the roof for that row size and residency (nothing else in the kernel: index hash + load + xor),
//   * run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum / TCC_HIT_sum TCC_MISS_sum) the
//     counter can be calibrated against the algorithmic bytes for this access pattern.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/calib_gather scripts/calib_gather.hip
//   scripts/calib_gather <table MiB> <row bytes: 8|16|32|64|128|256|512> <row requests (millions)> [loads in flight per lane = 8] [blocks per CU = 8]
// prints one JSON line.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                         \
    do                                                                                                \
    {                                                                                                 \
        hipError_t e = (x);                                                                           \
        if (e != hipSuccess)                                                                          \
        {                                                                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                    \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_kernel(uint64_t* t, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = mix64(i + 1);
}

// A group of LPR lanes (8 bytes each) fetches one row per request; a wave issues 64/LPR different rows per load instruction.
// Every lane keeps U loads in flight (U independent row indices per trip).
template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t* __restrict__ table, uint64_t n_rows, uint64_t trips, uint64_t seed,
                                                     uint64_t* __restrict__ sink)
{
    const uint64_t gtid  = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t group = gtid / LPR;
    const uint32_t sub   = (uint32_t)(gtid % LPR);
    uint64_t       acc   = 0;
    for (uint64_t it = 0; it < trips; ++it)
    {
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const uint64_t x   = mix64(seed + (group * trips + it) * U + u);
            const uint64_t row = (uint64_t)(((unsigned __int128)x * n_rows) >> 64);
            v[u]               = table[row * LPR + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc ^= v[u];
    }
    if (acc == 0x1234567812345678ull)
        sink[0] = acc;
}

template <int LPR>
static int run(int U, const uint64_t* table, uint64_t n_rows, uint64_t trips, uint64_t* sink, unsigned blocks, hipStream_t st)
{
#define LAUNCH(UU) hipLaunchKernelGGL((gather_kernel<LPR, UU>), dim3(blocks), dim3(256), 0, st, table, n_rows, trips, 42ull, sink)
    switch (U)
    {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 6: LAUNCH(6); break;
        case 8: LAUNCH(8); break;
        case 12: LAUNCH(12); break;
        case 16: LAUNCH(16); break;
        default: fprintf(stderr, "loads in flight: 1 2 4 6 8 12 16\n"); return 1;
    }
#undef LAUNCH
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 4)
    {
        fprintf(stderr, "usage: %s <table MiB> <row bytes> <row requests, millions> [loads in flight] [blocks per CU]\n", argv[0]);
        return 2;
    }
    const uint64_t table_bytes = (uint64_t)atoll(argv[1]) << 20;
    const int      row_bytes   = atoi(argv[2]);
    const uint64_t want        = (uint64_t)(atof(argv[3]) * 1e6);
    const int      U           = argc > 4 ? atoi(argv[4]) : 8;
    const int      bpc         = argc > 5 ? atoi(argv[5]) : 8;
    const int      lpr         = row_bytes / 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned blocks  = (unsigned)prop.multiProcessorCount * bpc;
    const uint64_t lanes   = (uint64_t)blocks * 256, groups = lanes / lpr;
    const uint64_t trips   = (want + groups * U - 1) / (groups * U);
    const uint64_t n_req   = groups * U * trips;
    const uint64_t n_rows  = table_bytes / row_bytes;
    uint64_t *     table = nullptr, *sink = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&table), table_bytes));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, table, table_bytes / 8);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const int reps = 5;
    for (int rep = 0; rep < reps + 1; ++rep) // (first launch = warm-up, not timed)
    {
        CK(hipEventRecord(e0, st));
        int rc = 1;
        switch (lpr)
        {
            case 1: rc = run<1>(U, table, n_rows, trips, sink, blocks, st); break;
            case 2: rc = run<2>(U, table, n_rows, trips, sink, blocks, st); break;
            case 4: rc = run<4>(U, table, n_rows, trips, sink, blocks, st); break;
            case 8: rc = run<8>(U, table, n_rows, trips, sink, blocks, st); break;
            case 16: rc = run<16>(U, table, n_rows, trips, sink, blocks, st); break;
            case 32: rc = run<32>(U, table, n_rows, trips, sink, blocks, st); break;
            case 64: rc = run<64>(U, table, n_rows, trips, sink, blocks, st); break;
            default: fprintf(stderr, "row bytes: 8 16 32 64 128 256 512\n");
        }
        if (rc)
            return 1;
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep)
        {
            best = ms < best ? ms : best;
            sum += ms;
        }
    }
    const double bytes = (double)n_req * row_bytes;
    printf("{\"table_mib\": %llu, \"row_bytes\": %d, \"row_requests_per_launch\": %llu, \"algorithmic_bytes_per_launch\": %.0f, "
           "\"loads_in_flight_per_lane\": %d, \"blocks_per_cu\": %d, \"timed_launches\": %d, \"launches_total\": %d, \"ms_avg\": %.4f, "
           "\"ms_best\": %.4f, \"algorithmic_gbs\": %.1f, \"line128_gbs\": %.1f}\n",
           (unsigned long long)(table_bytes >> 20), row_bytes, (unsigned long long)n_req, bytes, U, bpc, reps, reps + 1, sum / reps, best,
           bytes / (sum / reps) / 1e6, (double)n_req * (row_bytes < 128 ? 128 : row_bytes) / (sum / reps) / 1e6);
    return 0;
}
