// gn_fastq.hip -- four-line FASTQ text tokenised on the device.
//
// The reference reads its input through seqan3::sequence_file_input on one thread per file (GanonClassify.cpp:1220-1287,1433);
// the host side here parses slabs of the file on several threads (host/seq_io.cpp, ParallelFastq).  On a 16-core host that parse
// is the largest consumer of CPU time of a whole run (profiles/r03_e2e_ab_cpu.txt: 3.5 of 8 CPU seconds per 64 M reads), and the
// run is bound by CPU seconds, not by the link or the kernels.  So for uncompressed FASTQ the host only copies the file, slab by
// slab, into page-locked memory; the text travels as it is and the records are found here:
//
//   1. newline count per 4 KiB tile, exclusive scan, newline positions (line l ends at nl[l]);
//   2. one thread per group of four lines checks what the host's slab parser checks for a record (ParallelFastq::Impl::parse):
//      line 0 begins with '@', line 2 begins with '+', the quality line is exactly as long as the sequence line ('\r' before the
//      '\n' of the sequence line does not count, as in RangeLines::line), every letter of the sequence is a dna15 letter;
//   3. sequence lengths are scanned into the offsets the minimiser kernels take (d_off1), the letters are copied back to back
//      into d_bases -- from here on the batch looks like one uploaded through gn_stream_upload_reads.
//
// The FIRST record that does not pass ends the batch: records before it are the batch, `parsed_bytes` says where it begins.
// What the caller does then is its business (the host hands the rest of the file to its sequential reader, which produces the
// records, the error message or the wrapped-line handling of the reference's parser -- the device never guesses).
#include "gn_internal.h"
#include "gn_scan.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define GN_FQ_TILE 4096u // bytes per block of the newline kernels: 256 threads x 16 bytes

// 0x80 in every byte of x that is '\n' (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t gn_fq_nl_mask(uint32_t x)
{
    const uint32_t t = x ^ 0x0A0A0A0Au;
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

__device__ __forceinline__ uint4 gn_fq_load16(const uint8_t* text, uint64_t n, uint64_t p)
{
    // (the buffer is padded to a multiple of 16 bytes; bytes at and beyond n read as 0)
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p < n)
    {
        v = *reinterpret_cast<const uint4*>(text + p);
        if (p + 16 > n)
        {
            uint32_t       w[4] = { v.x, v.y, v.z, v.w };
            const uint32_t keep = (uint32_t)(n - p); // 1..15
            for (uint32_t i = 0; i < 4; ++i)
            {
                const uint32_t lo = i * 4;
                if (keep <= lo)
                    w[i] = 0;
                else if (keep < lo + 4)
                    w[i] &= (1u << ((keep - lo) * 8)) - 1u;
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void gn_fq_count_kernel(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ tile_cnt)
{
    __shared__ uint32_t part[4];
    const uint64_t      p = (uint64_t)blockIdx.x * GN_FQ_TILE + threadIdx.x * 16u;
    const uint4         v = gn_fq_load16(text, n, p);
    uint32_t c = __popc(gn_fq_nl_mask(v.x)) + __popc(gn_fq_nl_mask(v.y)) + __popc(gn_fq_nl_mask(v.z)) + __popc(gn_fq_nl_mask(v.w));
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0)
        part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        tile_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void gn_fq_lines_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ tile_off,
                                                          uint32_t* __restrict__ nl, uint32_t nl_cap)
{
    __shared__ uint32_t part[4];
    const uint64_t      p = (uint64_t)blockIdx.x * GN_FQ_TILE + threadIdx.x * 16u;
    const uint4         v = gn_fq_load16(text, n, p);
    const uint32_t      m[4] = { gn_fq_nl_mask(v.x), gn_fq_nl_mask(v.y), gn_fq_nl_mask(v.z), gn_fq_nl_mask(v.w) };
    const uint32_t      c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
    const uint32_t      lane = threadIdx.x & 63u;
    uint32_t            inc = c; // inclusive prefix inside the wave
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= (uint32_t)o)
            inc += up;
    }
    if (lane == 63u)
        part[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = tile_off[blockIdx.x] + inc - c;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv)
        base += part[wv];
    for (uint32_t i = 0; i < 4; ++i)
    {
        uint32_t z = m[i];
        while (z)
        {
            const uint32_t b = (uint32_t)__builtin_ctz(z);
            z &= z - 1;
            if (base < nl_cap)
                nl[base] = (uint32_t)(p + i * 4u + (b >> 3));
            ++base;
        }
    }
}

// fq: [0] lines [1] first record that does not pass [2] records of the batch [3] their bases [4] bytes they cover [5] lines past capacity
// FASTA = false: groups of four lines  @id / letters / +... / quality.
// FASTA = true:  groups of two lines   >id / letters  -- a record whose letters are on ONE line, followed by the next record's '>'
//                (or by the end of the text); wrapped sequences, blank lines, ';' headers end the batch like any other record
//                the rule does not cover (the host's sequential reader takes it from there, white space, digits and all).
template <int FMT>
__global__ __launch_bounds__(256) void gn_fq_records_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ nl,
                                                            const uint32_t* __restrict__ n_lines_at, uint32_t max_reads, uint32_t n_threads,
                                                            uint64_t n_bytes, uint32_t* __restrict__ rec_at, uint32_t* __restrict__ seq_at,
                                                            uint32_t* __restrict__ seq_len, unsigned long long* __restrict__ fq)
{
    constexpr bool     FASTA = FMT == GN_TEXT_FASTA;
    constexpr uint8_t  HDR   = '>';
    constexpr uint32_t LPR   = FASTA ? 2u : 4u; // lines per record
    const uint32_t     r   = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_threads)
        return;
    const uint32_t n_lines = *n_lines_at;
    uint32_t       n_rec   = n_lines / LPR;
    if (n_rec > max_reads)
        n_rec = max_reads;
    if (r == 0)
        fq[0] = n_lines;
    if (r >= n_rec)
    {
        seq_len[r] = 0; // (scan input up to n_threads)
        return;
    }
    const uint32_t p0 = r ? nl[LPR * r - 1] + 1u : 0u;
    const uint32_t a = nl[LPR * r], b = nl[LPR * r + 1];
    uint32_t       slen = b - a - 1u;
    if (slen && text[b - 1] == '\r')
        --slen;
    bool ok;
    if (FASTA)
    {
        const uint8_t first = a + 1u < b ? text[a + 1] : (uint8_t)'A';
        ok = a > p0 && text[p0] == HDR && first != '>' && first != ';' && ((uint64_t)b + 1u >= n_bytes || text[b + 1] == HDR);
    }
    else
    {
        const uint32_t c = nl[4 * r + 2], d = nl[4 * r + 3];
        ok = a > p0 && text[p0] == '@' && c > b + 1u && text[b + 1] == '+' && d - c - 1u == slen;
    }
    rec_at[r]  = p0;
    seq_at[r]  = a + 1u;
    seq_len[r] = ok ? slen : 0u;
    if (!ok)
        atomicMin(&fq[1], (unsigned long long)r);
}

// dna15 letters, either case (host/seq_io.cpp LegalTable): A B C D G H K M N R S T U V W Y
#define GN_FQ_LEGAL                                                                                                                        \
    ((1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 6) | (1u << 7) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 17) | (1u << 18) |   \
     (1u << 19) | (1u << 20) | (1u << 21) | (1u << 22) | (1u << 24))

// one wave per record: letters back to back into `bases`, checked on the way
__global__ __launch_bounds__(256) void gn_fq_copy_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ seq_at,
                                                         const uint32_t* __restrict__ seq_len, const uint64_t* __restrict__ off,
                                                         const unsigned long long* fq_lines, uint32_t max_reads, uint32_t bound, uint32_t lpr,
                                                         uint8_t* __restrict__ bases, unsigned long long* fq)
{
    const uint32_t lane    = threadIdx.x & 63u;
    const uint32_t wave    = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t       n_rec   = (uint32_t)(fq_lines[0] / lpr);
    if (n_rec > max_reads)
        n_rec = max_reads;
    // The records kernel and the scan looked at `bound` groups of four lines (no more records of >= 6 bytes fit the text).  Text
    // that is mostly blank lines has more groups than that; one of the first `bound` is then too short to be a record and has
    // ended the batch already -- what lies beyond was never written (seq_len / seq_at / off) and is not touched here.
    if (n_rec > bound)
        n_rec = bound;
    for (uint32_t r = wave; r < n_rec; r += n_waves)
    {
        const uint32_t len = seq_len[r];
        const uint8_t* src = text + seq_at[r];
        uint8_t*       dst = bases + off[r];
        bool           bad = false;
        for (uint32_t j = lane; j < len; j += 64u)
        {
            const uint8_t  ch = src[j];
            const uint32_t u  = (uint32_t)(ch | 0x20u) - 'a';
            bad               = bad || u >= 26u || ((GN_FQ_LEGAL >> u) & 1u) == 0u || (ch & 0x40u) == 0u;
            dst[j]            = ch;
        }
        if (__ballot(bad) != 0 && lane == 0)
            atomicMin(&fq[1], (unsigned long long)r);
    }
}

// One text (fq2 = nullptr) or the two mate files of a pair: the batch is the records BOTH texts hold before their first
// group of lines that is no record -- fq[2]; fq[4] / fq2[4] = the bytes of each text those records cover.
__global__ void gn_fq_finish_kernel(const uint32_t* __restrict__ nl, const uint64_t* __restrict__ off, uint32_t max_reads, uint64_t n_bytes,
                                    uint32_t lpr, unsigned long long* __restrict__ fq, const uint32_t* __restrict__ nl2,
                                    unsigned long long* __restrict__ fq2, uint64_t bases_bound)
{
    auto taken = [&](const unsigned long long* f) {
        uint64_t n_rec = f[0] / lpr;
        if (n_rec > max_reads)
            n_rec = max_reads;
        return f[1] < n_rec ? (uint64_t)f[1] : n_rec;
    };
    uint64_t v = taken(fq);
    if (fq2)
    {
        const uint64_t v2 = taken(fq2);
        v                 = v2 < v ? v2 : v;
        fq2[2]            = v;
        fq2[4]            = v ? (uint64_t)nl2[lpr * v - 1] + 1u : 0u;
    }
    fq[2] = v;
    fq[3] = fq2 ? bases_bound : off[v];
    fq[4] = v ? (uint64_t)nl[lpr * v - 1] + 1u : 0u;
    fq[5] = n_bytes;
}

void gn_fastq_release(gn_stream* s)
{
    if (s->fq_probe[3] > 0)
        fprintf(stderr, "[hip call timing] gn_stream_upload_text x%.0f: initial sync %.3f ms, copy call %.3f ms, other calls %.3f ms per batch\n", s->fq_probe[3],
                s->fq_probe[0] / s->fq_probe[3] * 1e3, s->fq_probe[1] / s->fq_probe[3] * 1e3, s->fq_probe[2] / s->fq_probe[3] * 1e3);
    for (void* p : { (void*)s->d_text, (void*)s->d_fq_tile, (void*)s->d_fq_nl, (void*)s->d_fq_rec, (void*)s->d_fq_seq, (void*)s->d_fq_len, (void*)s->d_fq,
                     (void*)s->d_fq_scan, (void*)s->d_fq_hoff, (void*)s->d_fq_hdr, s->d_fq_hscan, (void*)s->d_fq2_tile, (void*)s->d_fq2_nl, (void*)s->d_fq2_rec, (void*)s->d_fq2_seq, (void*)s->d_fq2_len })
        if (p)
            hipFree(p);
    if (s->h_fq)
        hipHostFree(s->h_fq);
    s->d_text = nullptr;
    s->d_fq_tile = s->d_fq_nl = s->d_fq_rec = s->d_fq_seq = s->d_fq_len = nullptr;
    s->d_fq2_tile = s->d_fq2_nl = s->d_fq2_rec = s->d_fq2_seq = s->d_fq2_len = nullptr;
    s->d_fq = nullptr;
    s->d_fq_scan = nullptr;
    s->d_fq_hoff = nullptr;
    s->d_fq_hdr = nullptr;
    s->d_fq_hscan = nullptr;
    s->h_fq = nullptr;
}

static int gn_fastq_prepare(gn_stream* s, bool pair)
{
    if (!s->d_text)
    {
        // the text never holds more bytes than the stream holds bases (every base is a byte of it)
        s->fq_text_cap  = s->max_bases;
        s->fq_tiles_cap = (uint32_t)((s->fq_text_cap + GN_FQ_TILE - 1) / GN_FQ_TILE) + 1u;
        s->fq_nl_cap    = 4u * s->max_reads + 4u;
        GN_HIP(hipMalloc(&s->d_text, s->fq_text_cap + 64));
        GN_HIP(hipMalloc(&s->d_fq_tile, ((size_t)s->fq_tiles_cap + 1) * 2 * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq_nl, (size_t)s->fq_nl_cap * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq_rec, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq_seq, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq_len, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq, 16 * sizeof(unsigned long long))); // [0..7] the text, [8..15] the mates' text
        GN_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_fq), 16 * sizeof(unsigned long long), hipHostMallocDefault));
        size_t t1 = 0, t2 = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, t1, s->d_fq_tile, s->d_fq_tile, (int)(s->fq_tiles_cap + 1), s->st);
        gn_scan_counts_from(nullptr, t2, s->d_fq_len, s->d_off1, (uint64_t)0, (int)(s->max_reads + 1), s->st);
        s->fq_scan_bytes = std::max(t1, t2) + 256;
        GN_HIP(hipMalloc(&s->d_fq_scan, s->fq_scan_bytes));
    }
    if (pair && !s->d_fq2_nl) // the mates' text has line and record tables of its own
    {
        GN_HIP(hipMalloc(&s->d_fq2_tile, ((size_t)s->fq_tiles_cap + 1) * 2 * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq2_nl, (size_t)s->fq_nl_cap * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq2_rec, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq2_seq, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq2_len, ((size_t)s->max_reads + 1) * sizeof(uint32_t)));
    }
    return GN_OK;
}

// the kernels of one text, queued on st: lines -> records -> offsets (starting at off_init: where its letters begin in d_bases) -> letters
static int gn_fq_enqueue(gn_stream* s, const uint8_t* d_text, uint64_t n_bytes, int format, uint32_t* d_tile, uint32_t* d_nl, uint32_t* d_rec,
                         uint32_t* d_seq, uint32_t* d_len, unsigned long long* d_fq, uint64_t* d_off, uint64_t off_init, hipStream_t st)
{
    const bool     fasta = format != GN_TEXT_FASTQ; // (two lines per record)
    const uint32_t lpr   = fasta ? 2u : 4u;
    const uint32_t tiles = (uint32_t)((n_bytes + GN_FQ_TILE - 1) / GN_FQ_TILE);
    uint32_t*      cnt   = d_tile;
    uint32_t*      toff  = d_tile + s->fq_tiles_cap + 1;
    GN_HIP(hipMemsetAsync(cnt + tiles, 0, sizeof(uint32_t), st));
    GN_HIP(hipMemsetAsync(d_fq, 0, 8 * sizeof(unsigned long long), st));
    GN_HIP(hipMemsetAsync(d_fq + 1, 0xFF, sizeof(unsigned long long), st));
    if (tiles)
        hipLaunchKernelGGL(gn_fq_count_kernel, dim3(tiles), dim3(256), 0, st, d_text, n_bytes, cnt);
    size_t tmp = s->fq_scan_bytes;
    GN_HIP(hipcub::DeviceScan::ExclusiveSum(s->d_fq_scan, tmp, cnt, toff, (int)(tiles + 1), st));
    if (tiles)
        hipLaunchKernelGGL(gn_fq_lines_kernel, dim3(tiles), dim3(256), 0, st, d_text, n_bytes, toff, d_nl, s->fq_nl_cap);
    // a four-line record is at least 6 bytes ("@\n\n+\n\n" is not even legal), a two-line one 3 (">\n\n"): bound of the per-record launches
    const uint32_t bound = (uint32_t)std::min<uint64_t>(s->max_reads, n_bytes / (fasta ? 3 : 6)) + 1u;
    if (format == GN_TEXT_FASTA)
        hipLaunchKernelGGL(gn_fq_records_kernel<1>, dim3((bound + 255) / 256), dim3(256), 0, st, d_text, d_nl, toff + tiles, s->max_reads, bound,
                           n_bytes, d_rec, d_seq, d_len, d_fq);
    else
        hipLaunchKernelGGL(gn_fq_records_kernel<0>, dim3((bound + 255) / 256), dim3(256), 0, st, d_text, d_nl, toff + tiles, s->max_reads, bound,
                           n_bytes, d_rec, d_seq, d_len, d_fq);
    tmp = s->fq_scan_bytes;
    GN_HIP(gn_scan_counts_from(s->d_fq_scan, tmp, d_len, d_off, off_init, (int)bound, st));
    const uint32_t blocks = std::min<uint32_t>((bound + 3) / 4, (uint32_t)s->f->n_cu * 8u);
    hipLaunchKernelGGL(gn_fq_copy_kernel, dim3(blocks), dim3(256), 0, st, d_text, d_seq, d_len, d_off, d_fq, s->max_reads, bound, lpr, s->d_bases, d_fq);
    return GN_OK;
}

// src_device < 0: the texts are host memory; otherwise device memory of that device
static int gn_upload_texts(gn_stream* s, const uint8_t* text, uint64_t n_bytes, const uint8_t* text2, uint64_t n_bytes2, bool pair, int format, int src_device = -1,
                           int src_device2 = -2 /* -2: where text 1 lies */)
{
    if (!s || (!text && n_bytes) || (pair && !text2 && n_bytes2))
        return gn_fail(GN_EINVAL, "gn_stream_upload_text: null argument");
    if (format != GN_TEXT_FASTQ && format != GN_TEXT_FASTA)
        return gn_fail(GN_EINVAL, "gn_stream_upload_text: format %d", format);
    const bool     fasta = format != GN_TEXT_FASTQ; // (two lines per record)
    const uint64_t at2   = (n_bytes + 15) & ~15ull; // the mates' text begins on a 16-byte lane of the device buffer
    const uint64_t total = pair ? at2 + n_bytes2 : n_bytes;
    if (total > s->max_bases || n_bytes >= 0xFFFFFFF0ull || n_bytes2 >= 0xFFFFFFF0ull)
        return gn_fail(GN_EINVAL, "text of %llu bytes exceeds the stream capacity (%llu bytes)", (unsigned long long)total,
                       (unsigned long long)s->max_bases);
    GN_HIP(hipSetDevice(s->f->device));
    int rc = gn_fastq_prepare(s, pair);
    if (rc)
        return rc;
    const bool        probe = gn_sw().debug;
    auto              now   = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double      p0    = probe ? now() : 0;
    GN_HIP(hipStreamSynchronize(s->st)); // previous batch must be done before its inputs are overwritten
    const double p1 = probe ? now() : 0;
    hipStream_t  st = s->st;
    if (n_bytes && src_device >= 0 && src_device != s->f->device)
    {
        gn_peer_enable(s->f->device, src_device);
        GN_HIP(hipMemcpyPeerAsync(s->d_text, s->f->device, text, src_device, n_bytes, st));
    }
    else if (n_bytes)
        GN_HIP(hipMemcpyAsync(s->d_text, text, n_bytes, src_device >= 0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    const int dev2 = src_device2 == -2 ? src_device : src_device2;
    if (pair && n_bytes2 && dev2 >= 0 && dev2 != s->f->device)
    {
        gn_peer_enable(s->f->device, dev2);
        GN_HIP(hipMemcpyPeerAsync(s->d_text + at2, s->f->device, text2, dev2, n_bytes2, st));
    }
    else if (pair && n_bytes2)
        GN_HIP(hipMemcpyAsync(s->d_text + at2, text2, n_bytes2, dev2 >= 0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    const double p2 = probe ? now() : 0;
    rc = gn_fq_enqueue(s, s->d_text, n_bytes, format, s->d_fq_tile, s->d_fq_nl, s->d_fq_rec, s->d_fq_seq, s->d_fq_len, s->d_fq, s->d_off1, 0, st);
    if (rc)
        return rc;
    if (pair) // mate i's letters: behind every letter the first text can hold (off2 starts at the second text's place in the buffer)
    {
        rc = gn_fq_enqueue(s, s->d_text + at2, n_bytes2, format, s->d_fq2_tile, s->d_fq2_nl, s->d_fq2_rec, s->d_fq2_seq, s->d_fq2_len, s->d_fq + 8,
                           s->d_off2, at2, st);
        if (rc)
            return rc;
    }
    hipLaunchKernelGGL(gn_fq_finish_kernel, dim3(1), dim3(1), 0, st, s->d_fq_nl, s->d_off1, s->max_reads, n_bytes, fasta ? 2u : 4u, s->d_fq,
                       pair ? s->d_fq2_nl : nullptr, pair ? s->d_fq + 8 : nullptr, total);
    GN_HIP(hipMemcpyAsync(s->h_fq, s->d_fq, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    GN_HIP(hipGetLastError());
    if (probe)
    {
        const double p3 = now();
        s->fq_probe[0] += p1 - p0;
        s->fq_probe[1] += p2 - p1;
        s->fq_probe[2] += p3 - p2;
        s->fq_probe[3] += 1;
    }
    s->fq_pending = true;
    s->fq_pair    = pair;
    s->fq_bytes   = n_bytes;
    s->have_reads = false;
    s->classified = false;
    s->ctr_copied = false;
    s->hashed     = false;
    return GN_OK;
}

extern "C" int gn_stream_upload_fastq(gn_stream* s, const uint8_t* text, uint64_t n_bytes)
{
    return gn_upload_texts(s, text, n_bytes, nullptr, 0, false, GN_TEXT_FASTQ);
}

extern "C" int gn_stream_upload_text(gn_stream* s, const uint8_t* text, uint64_t n_bytes, int format)
{
    return gn_upload_texts(s, text, n_bytes, nullptr, 0, false, format);
}

extern "C" int gn_stream_upload_text_device(gn_stream* s, const uint8_t* d_text, uint64_t n_bytes, int format, int src_device)
{
    if (src_device < 0)
        return gn_fail(GN_EINVAL, "gn_stream_upload_text_device: device %d", src_device);
    return gn_upload_texts(s, d_text, n_bytes, nullptr, 0, false, format, src_device);
}

extern "C" int gn_stream_upload_text_pair_device(gn_stream* s, const uint8_t* d_text1, uint64_t n_bytes1, const uint8_t* d_text2, uint64_t n_bytes2, int format,
                                                 int src_device)
{
    if (src_device < 0)
        return gn_fail(GN_EINVAL, "gn_stream_upload_text_pair_device: device %d", src_device);
    return gn_upload_texts(s, d_text1, n_bytes1, d_text2, n_bytes2, true, format, src_device);
}

extern "C" int gn_stream_upload_text_pair_devices(gn_stream* s, const uint8_t* d_text1, uint64_t n_bytes1, int src_device1, const uint8_t* d_text2, uint64_t n_bytes2,
                                                  int src_device2, int format)
{
    if (src_device1 < 0 || src_device2 < 0)
        return gn_fail(GN_EINVAL, "gn_stream_upload_text_pair_devices: devices %d, %d", src_device1, src_device2);
    return gn_upload_texts(s, d_text1, n_bytes1, d_text2, n_bytes2, true, format, src_device1, src_device2);
}

extern "C" int gn_stream_upload_text_pair(gn_stream* s, const uint8_t* text1, uint64_t n_bytes1, const uint8_t* text2, uint64_t n_bytes2, int format)
{
    return gn_upload_texts(s, text1, n_bytes1, text2, n_bytes2, true, format);
}

static int gn_text_index(gn_stream* s, bool pair, uint32_t* n_reads, uint64_t* n_bases, uint64_t* parsed_bytes, uint64_t* parsed_bytes2)
{
    if (!s)
        return gn_fail(GN_EINVAL, "null stream");
    if (!s->fq_pending || s->fq_pair != pair)
        return gn_fail(GN_EINVAL, pair ? "gn_stream_text_pair_index: no pair of texts was uploaded on this stream"
                                       : "gn_stream_fastq_index: no text was uploaded on this stream");
    GN_HIP(hipSetDevice(s->f->device));
    GN_HIP(hipStreamSynchronize(s->st));
    s->fq_pending = false;
    s->v_hashes   = s->d_hashes; // (a batch of its own, as after gn_stream_upload_reads)
    s->v_slot_off = s->d_slot_off;
    s->v_nh       = s->d_nh;
    s->v_status   = s->d_status;
    s->src        = nullptr;
    s->n_reads    = (uint32_t)s->h_fq[2];
    s->n_bases    = s->h_fq[3];
    s->fq_reads   = s->n_reads;
    s->paired     = pair;
    s->have_reads = true;
    s->classified = false;
    s->hashed     = false;
    s->build_distinct = ~0ull;
    if (n_reads)
        *n_reads = s->n_reads;
    if (n_bases)
        *n_bases = s->n_bases;
    if (parsed_bytes)
        *parsed_bytes = s->h_fq[4];
    if (parsed_bytes2)
        *parsed_bytes2 = s->h_fq[8 + 4];
    return GN_OK;
}

extern "C" int gn_stream_fastq_index(gn_stream* s, uint32_t* n_reads, uint64_t* n_bases, uint64_t* parsed_bytes)
{
    return gn_text_index(s, false, n_reads, n_bases, parsed_bytes, nullptr);
}

extern "C" int gn_stream_text_pair_index(gn_stream* s, uint32_t* n_reads, uint64_t* parsed_bytes1, uint64_t* parsed_bytes2)
{
    return gn_text_index(s, true, n_reads, nullptr, parsed_bytes1, parsed_bytes2);
}

extern "C" int gn_stream_fastq_keep(gn_stream* s, uint32_t n_reads)
{
    if (!s || !s->have_reads || n_reads > s->fq_reads)
        return gn_fail(GN_EINVAL, "gn_stream_fastq_keep: not a tokenised batch, or more reads than it holds");
    if (s->classified || s->hashed)
        return gn_fail(GN_EINVAL, "gn_stream_fastq_keep: the batch is being classified already");
    s->n_reads = n_reads; // (n_bases stays an upper bound: the kernels go by the offsets)
    return GN_OK;
}

static int gn_text_records(gn_stream* s, const uint32_t* d_rec, const uint32_t* d_seq, const uint32_t* d_len, uint32_t* rec_at, uint32_t* seq_at,
                           uint32_t* seq_len)
{
    GN_HIP(hipSetDevice(s->f->device));
    const size_t nb = (size_t)s->n_reads * sizeof(uint32_t);
    if (nb)
    {
        if (rec_at)
            GN_HIP(hipMemcpyAsync(rec_at, d_rec, nb, hipMemcpyDeviceToHost, s->st));
        if (seq_at)
            GN_HIP(hipMemcpyAsync(seq_at, d_seq, nb, hipMemcpyDeviceToHost, s->st));
        if (seq_len)
            GN_HIP(hipMemcpyAsync(seq_len, d_len, nb, hipMemcpyDeviceToHost, s->st));
    }
    GN_HIP(hipStreamSynchronize(s->st));
    return GN_OK;
}

extern "C" int gn_stream_fastq_records(gn_stream* s, uint32_t* rec_at, uint32_t* seq_at, uint32_t* seq_len)
{
    if (!s || !s->have_reads || !s->d_text)
        return gn_fail(GN_EINVAL, "gn_stream_fastq_records: not a tokenised batch");
    return gn_text_records(s, s->d_fq_rec, s->d_fq_seq, s->d_fq_len, rec_at, seq_at, seq_len);
}

extern "C" int gn_stream_text_pair_records2(gn_stream* s, uint32_t* rec_at, uint32_t* seq_at, uint32_t* seq_len)
{
    if (!s || !s->have_reads || !s->d_fq2_rec || !s->fq_pair)
        return gn_fail(GN_EINVAL, "gn_stream_text_pair_records2: not a tokenised pair of texts");
    return gn_text_records(s, s->d_fq2_rec, s->d_fq2_seq, s->d_fq2_len, rec_at, seq_at, seq_len);
}

// ---- header lines of the batch's records, for a caller that does not hold the text (gn_stream_upload_text_device) ----------------------
__global__ void gn_fq_hlen_kernel(const uint32_t* __restrict__ rec, const uint32_t* __restrict__ seq, uint32_t n, uint32_t* __restrict__ hlen)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n)
        hlen[i] = i < n ? seq[i] - rec[i] : 0u; // (from the '@' or '>' to the newline, both included)
}

// one wave per 64 records; a record's header is copied by its lane eight bytes at a time where both ends allow it
__global__ void gn_fq_hcopy_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ rec, const uint32_t* __restrict__ hoff, uint32_t n,
                                   uint8_t* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint8_t* a = text + rec[i];
    uint8_t*       b = dst + hoff[i];
    const uint32_t len = hoff[i + 1] - hoff[i];
    for (uint32_t k = 0; k < len; ++k)
        b[k] = a[k];
}

extern "C" int gn_stream_fastq_headers(gn_stream* s, uint8_t* dst, uint64_t cap, uint32_t* hdr_off, uint64_t* n_bytes)
{
    if (!s || !s->have_reads || !s->d_text || !hdr_off || !n_bytes || (!dst && cap))
        return gn_fail(GN_EINVAL, "gn_stream_fastq_headers: not a tokenised batch, or null argument");
    GN_HIP(hipSetDevice(s->f->device));
    const uint32_t n = s->n_reads;
    *n_bytes         = 0;
    hdr_off[0]       = 0;
    if (n == 0)
        return GN_OK;
    if (!s->d_fq_hoff)
    {
        GN_HIP(hipMalloc(&s->d_fq_hoff, ((size_t)s->max_reads + 2) * 2 * sizeof(uint32_t)));
        GN_HIP(hipMalloc(&s->d_fq_hdr, s->fq_text_cap + 64));
        size_t tmp = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, s->d_fq_hoff, s->d_fq_hoff, (int)(s->max_reads + 1), s->st);
        s->fq_hscan_bytes = tmp + 256;
        GN_HIP(hipMalloc(&s->d_fq_hscan, s->fq_hscan_bytes));
    }
    uint32_t* hlen = s->d_fq_hoff;
    uint32_t* hoff = s->d_fq_hoff + s->max_reads + 2;
    hipLaunchKernelGGL(gn_fq_hlen_kernel, dim3((n + 256) / 256), dim3(256), 0, s->st, s->d_fq_rec, s->d_fq_seq, n, hlen);
    size_t tmp = s->fq_hscan_bytes;
    GN_HIP(hipcub::DeviceScan::ExclusiveSum(s->d_fq_hscan, tmp, hlen, hoff, (int)(n + 1), s->st));
    hipLaunchKernelGGL(gn_fq_hcopy_kernel, dim3((n + 255) / 256), dim3(256), 0, s->st, s->d_text, s->d_fq_rec, hoff, n, s->d_fq_hdr);
    GN_HIP(hipGetLastError());
    GN_HIP(hipMemcpyAsync(hdr_off, hoff, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    const uint64_t total = hdr_off[n];
    *n_bytes             = total;
    if (total > cap)
        return gn_fail(GN_EOVERFLOW, "gn_stream_fastq_headers: %llu bytes of header lines, room for %llu", (unsigned long long)total, (unsigned long long)cap);
    GN_HIP(hipMemcpyAsync(dst, s->d_fq_hdr, total, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    return GN_OK;
}

extern "C" int gn_stream_fetch_letters(gn_stream* s, uint8_t* bases, uint64_t cap, uint64_t* off1, uint64_t* off2, uint64_t* n_bytes)
{
    if (!s || !s->have_reads || !off1 || !n_bytes || (s->paired && !off2) || (!bases && cap))
        return gn_fail(GN_EINVAL, "gn_stream_fetch_letters: no batch, or null argument");
    GN_HIP(hipSetDevice(s->f->device));
    const size_t nb = ((size_t)s->n_reads + 1) * sizeof(uint64_t);
    GN_HIP(hipMemcpyAsync(off1, s->d_off1, nb, hipMemcpyDeviceToHost, s->st));
    if (s->paired)
        GN_HIP(hipMemcpyAsync(off2, s->d_off2, nb, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    const uint64_t end = s->paired ? off2[s->n_reads] : off1[s->n_reads];
    *n_bytes           = end;
    if (end > cap)
        return gn_fail(GN_EOVERFLOW, "gn_stream_fetch_letters: %llu bytes of letters, room for %llu", (unsigned long long)end, (unsigned long long)cap);
    if (end)
        GN_HIP(hipMemcpyAsync(bases, s->d_bases, end, hipMemcpyDeviceToHost, s->st));
    GN_HIP(hipStreamSynchronize(s->st));
    return GN_OK;
}
