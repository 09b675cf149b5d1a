// gn_hibf.hip -- HIBF counting agent on the device.
//
// Reference semantics: raptor::hierarchical_interleaved_bloom_filter::counting_agent_type
//   /root/reference/src/ganon-classify/include/ganon-classify/hierarchical_interleaved_bloom_filter.hpp:432-460 (bulk_count_impl)
//   :506-523 (bulk_count) and select_matches(Filter<THIBF>) at /root/reference/src/ganon-classify/GanonClassify.cpp:543-577.
//
// The data-dependent recursion becomes a breadth-first work queue: level 0 = (read, ibf 0) for every counted read;
// a level kernel counts all of the read's minimisers in the item's IBF, evaluates the IBF's bin RUNS (a merged bin is a
// run of its own; a split user bin is a run of equal filename index -- exactly where the reference resets its running
// uint16 `sum`), and for runs with sum >= T either appends (read, child ibf) to the next level's queue or emits
// (read, user bin, sum).  Matches are finally radix-sorted by (read, user bin).
//
// The levels are launched back to back WITHOUT a host round trip: a level's kernels read their queue length from
// device memory, the tree depth (known at upload) bounds the number of levels, and the host synchronises once per
// batch -- to learn the match count for the sort and to check that no queue overflowed.
//
// Two kernels per level:
//   gn_hibf_reg_kernel<HF>   IBFs of at most 64 words (4096 technical bins; raptor's t_max IBFs are 64..1024 bins) and
//                            reads of at most 127 minimisers: the fast flat kernel's scheme on (read, ibf) items --
//                            8-byte lanes, Gp = pow2 >= W lanes per row, 64/Gp hashes per wave iteration (16 for the
//                            32-byte rows of a 256-bin IBF), h coalesced row requests per hash, bit-sliced SWAR
//                            counters in registers, groups added with lane-xor shuffles, SWAR compare with the cutoff.
//                            Single-bin runs (all of them unless a user bin is split) go bin -> table -> queue/match;
//                            multi-bin runs are summed from a 4 KB LDS image of the counters.
//   gn_hibf_level_kernel     everything else (wider IBFs, longer reads): LDS counters, one wave per item.
#include "gn_internal.h"

#include <hipcub/hipcub.hpp>
#include "gn_scan.h"

#include <algorithm>
#include <vector>

#define GN_WAVE 64

__constant__ uint64_t GN_HIBF_SEEDS[GN_IBF_MAX_HASH_FUNS] = GN_IBF_SEED_LIST;   // include/ganon_ibf_hash.h

__device__ __forceinline__ uint32_t gn_hibf_row(uint64_t v, uint32_t i, uint32_t shift, uint64_t S)
{
    uint64_t x = v * GN_HIBF_SEEDS[i];
    x ^= x >> shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint32_t)__umul64hi(x, S);
}

__device__ __forceinline__ void gn_hibf_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long gn_hibf_bcast64(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// Wave-uniform loads through the scalar cache (s_load into SGPRs): the data is read-only for the whole launch, the
// address is the same in every lane.  They use their own counter (lgkmcnt), so waiting for row loads (vmcnt) never waits
// for them and vice versa, and they need no VGPRs.  The compiler only emits them for the constant address space.
template <typename T>
__device__ __forceinline__ T gn_sload(const T* ptr)
{
    typedef const __attribute__((address_space(4))) T* cptr;
    return *(cptr)(uintptr_t)ptr;
}
// A pointer that was itself loaded from memory is "flat" to the compiler (flat_load counts on both wait counters and
// forces conservative waits); filter rows and tables live in global memory.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* gn_global(const T* ptr)
{
    return (const __attribute__((address_space(1))) T*)(uintptr_t)ptr;
}

struct GnHibfLevelParams
{
    const GnHibfIbfDev*       ibfs;
    const uint64_t*           hashes;
    const uint64_t*           slot_off;
    const uint32_t*           n_hashes;
    double                    rel_cutoff;
    const uint2*              work_in;
    const unsigned long long* count_in;  // items in work_in (device resident; clamped to work_cap)
    uint2*                    work_out;
    unsigned long long*       count_out; // next level's queue length (allocated slots, holes included)
    uint2*                    defer_out; // reg kernel: items it leaves to the LDS kernel of the same level
    unsigned long long*       defer_count;
    uint32_t                  work_cap;
    unsigned long long*       ctr;       // [0] match cursor, [2] algo bytes, [6] exact match count
    uint64_t*                 keys;      // (read << ub_bits) | user_bin
    uint32_t*                 vals;      // raw uint16 sum
    uint64_t                  match_cap;
    uint32_t                  ub_bits;
    uint32_t                  lds_bins;  // LDS kernel: counters per wave
    uint32_t                  n_reads;   // level 0 of the register-counter kernels: the reads are the items ...
    uint32_t                  read_base; // ... reads [read_base, read_base + n_reads) of the batch (a batch may be run in read ranges)
    const uint8_t*            status;
    const unsigned long long* work_base; // packed kernel: its items begin at work_in[*work_base] (a class of the level's sorted list); nullptr = 0
    uint32_t                  pack_gp;   // packed kernel: log2 of the lanes per row every item of the launch must have
    // packed kernel, ONE launch for a level whose queue was sorted by row width: class c (c < n_cls) holds cls_count[c] items from
    // work_in[cls_base[c]] on, all of cls_gp[c] lanes per row; the persistent waves walk the classes one after the other (n_cls = 0:
    // the single class described by pack_gp / count_in / work_base)
    uint32_t                  n_cls;
    uint8_t                   cls_gp[8];
    const unsigned long long* cls_count;
    const unsigned long long* cls_base;
    unsigned long long*       grab;      // packed kernel: per class the cursor its waves take their later batches from (zero at launch)
    uint32_t                  fake_hashes;
    uint32_t                  reread;
    uint32_t                  lds_region;   // packed kernel: 8-byte words of dynamic LDS per wave
    uint32_t                  stage_hashes; // packed kernel: the first GN_HIBF_NQ minimisers of every item are staged in LDS (classes of at
                                            // most 4 lanes per row; lds_region holds (64 >> narrowest class) items' worth)
    unsigned long long*       lvl_bytes; // this level's algorithmic bytes / line bytes (beside the batch totals ctr[2] / ctr[1]): the
    unsigned long long*       lvl_lines; // per-level figures of gn_stream_hibf_levels need no copy between the levels
    uint32_t                  wide;      // the reference's -DLONGREADS build (value_t = uint32_t): sums do not wrap at 2^16 and reads
                                         // of more than 65535 minimisers (GN_READ_BIG) are counted like the others
};

#define GN_HIBF_CHUNK 64u // wave-private slices of the work queue / match buffer (one global atomic per slice)
#define GN_HIBF_CHUNK_MAX 4096u

// Wave-private chunked append to the next level's queue and to the match buffer (a single counter address sustains only
// ~90 atomics/us).  A chunk is filled front to back by successive appends; what is left of it when the wave moves on
// (or ends) is filled with sentinels -- (read = 0xFFFFFFFF) / key = ~0 -- that the next level and the final sort ignore.
// Sentinels only ever go to slots no entry was written to, so no store order between lanes is relied on.
struct GnHibfAppender
{
    unsigned long long wq_base = 0, mq_base = 0;
    uint32_t           wq_left = 0, mq_left = 0;
    uint32_t           mq_chunk = GN_HIBF_CHUNK; // doubles with every chunk a wave asks for, up to GN_HIBF_CHUNK_MAX: at low cutoffs
                                                  // (thousands of chance matches per read) fixed 64-pair chunks meant 21 M atomics on
                                                  // one address per 1.35 G pairs -- most of the leaf level's time
    unsigned long long n_matches = 0;

    __device__ __forceinline__ void close_work(const GnHibfLevelParams& p, int lane)
    {
        for (uint32_t i = lane; i < wq_left; i += GN_WAVE)
            if (wq_base + i < p.work_cap)
                p.work_out[wq_base + i] = make_uint2(0xFFFFFFFFu, 0u);
        wq_left = 0;
    }
    __device__ __forceinline__ void close_matches(const GnHibfLevelParams& p, int lane)
    {
        for (uint32_t i = lane; i < mq_left; i += GN_WAVE)
            if (mq_base + i < p.match_cap)
                p.keys[mq_base + i] = ~0ULL;
        mq_left = 0;
    }
    // every lane calls; `merged` / `leaf` are this lane's hits of the current trip
    __device__ __forceinline__ void push(const GnHibfLevelParams& p, int lane, bool merged, bool leaf, uint32_t read, uint32_t tgt,
                                         uint32_t sum)
    {
        const uint64_t mm = __ballot(merged);
        if (mm)
        {
            const uint32_t need = (uint32_t)__popcll(mm);
            if (need > wq_left)
            {
                close_work(p, lane);
                const uint32_t     take = need > GN_HIBF_CHUNK ? need : GN_HIBF_CHUNK;
                unsigned long long nb   = 0;
                if (lane == 0)
                    nb = atomicAdd(p.count_out, (unsigned long long)take);
                wq_base = gn_hibf_bcast64(nb);
                wq_left = take;
            }
            if (merged)
            {
                const unsigned long long o = wq_base + __popcll(mm & ((1ULL << lane) - 1ULL));
                if (o < p.work_cap)
                    p.work_out[o] = make_uint2(read, tgt);
            }
            wq_base += need;
            wq_left -= need;
        }
        const uint64_t lm = __ballot(leaf);
        if (lm)
        {
            const uint32_t need = (uint32_t)__popcll(lm);
            if (need > mq_left)
            {
                close_matches(p, lane);
                const uint32_t     take = need > mq_chunk ? need : mq_chunk;
                mq_chunk                = mq_chunk < GN_HIBF_CHUNK_MAX ? mq_chunk * 2u : mq_chunk;
                unsigned long long nb   = 0;
                if (lane == 0)
                    nb = atomicAdd(&p.ctr[0], (unsigned long long)take);
                mq_base = gn_hibf_bcast64(nb);
                mq_left = take;
            }
            if (leaf)
            {
                const unsigned long long o = mq_base + __popcll(lm & ((1ULL << lane) - 1ULL));
                if (o < p.match_cap)
                {
                    p.keys[o] = ((uint64_t)read << p.ub_bits) | tgt;
                    p.vals[o] = sum;
                }
            }
            mq_base += need;
            mq_left -= need;
            n_matches += need;
        }
    }
    __device__ __forceinline__ void finish(const GnHibfLevelParams& p, int lane)
    {
        close_work(p, lane);
        close_matches(p, lane);
        if (lane == 0 && n_matches)
            atomicAdd(&p.ctr[6], n_matches);
    }
};

// ================================================================================================
// packed register-counter level kernel: 64/Gp ITEMS per wave side by side
// ================================================================================================
// For the narrow IBFs an HIBF is made of (a 256-bin IBF has 32-byte rows = 4 lanes) the per-item kernel below spends
// most of its instructions adding up the 16 hash groups of a wave and hashing redundantly.  This kernel turns the
// layout around: a group of Gp lanes owns one (read, ibf) ITEM and walks that item's minimisers one per iteration,
// so a wave works on 64/Gp items at once, every lane accumulates exactly the 64 bins it will threshold itself --
// no cross-lane sums at all -- and the lanes of a group share the h row hashes (lane i of the group computes
// function i, the others fetch it with a lane permute).  Per iteration a wave still issues h coalesced requests that
// touch 64/Gp different rows, so the memory system sees the same parallelism.
// All items of a launch must use the same Gp (p.pack_gp: the most common lane width of the level's IBFs); items whose
// IBF differs or whose read has more than 127 minimisers go to the per-item kernel's list.  Multi-bin runs (split user
// bins: raptor puts the largest user bins of a level there, so the TOP IBF of a real index has them and every read visits
// it) are summed from a per-wave LDS image of the byte counters, as in the per-item kernel: a group's lanes share the
// item's runs.
__device__ __forceinline__ uint32_t gn_hibf_row_seed(uint64_t v, uint64_t seed, uint32_t shift, uint32_t S)
{
    uint64_t x = v * seed;
    x ^= x >> shift;
    x *= GN_IBF_MULTIPLIER;
    return (uint32_t)__umul64hi(x, (uint64_t)S);
}

// Minimisers of an item staged in LDS.  With one to four lanes per row a wave holds 64 .. 16 items and every lane (group) used to fetch
// its item's next hash from global memory in every iteration: 64 different lines per wave and iteration, far more than the L1 and L2
// keep until the next iteration -- measured, a fifth of a level's fabric requests (profiles/r05_hibf_probe_fake2.jsonl: 9.88 -> 7.80 ms
// with the loads taken out).  Now the lanes of a group copy the item's first GN_HIBF_NQ hashes (24: a 150 bp read has 12 .. 25) into LDS
// before the row loop -- all loads in flight at once, two lines per item -- and the loop reads LDS; hashes beyond come from global memory.
#define GN_HIBF_NQ 24u
#define GN_HIBF_NQ_STRIDE 25u // (odd stride in 8-byte words)

// Four waves a SIMD: with the class loop and the staging the allocator takes 140 registers (three waves) unless told otherwise; held to
// 128 it spills a few dwords and the lower levels of the skewed tree run 11 % faster (profiles/r05_probe3_skew_w4.jsonl).  -DGN_PACK_WAVES3
// builds the other variant for A/B runs.
// (A third row set in flight per lane was tried -- six more registers instead of a fourth wave -- and changed nothing:
// profiles/r05_probe5_skew_rs3.jsonl.)
#ifdef GN_PACK_WAVES3
#define GN_PACK_ATTR
#else
#define GN_PACK_ATTR __attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
#ifdef GN_PACK_PROF
// (profiling build only: per class -- wave cycles in the class / row loop / staging / tail, batches, iterations, item iterations, lines)
__device__ unsigned long long gn_pack_prof_buf[8][8];
__device__ unsigned long long gn_pack_prof_wave[8192][2]; // per wave: first and last tick of the 100 MHz clock
#define GN_PP_NOW() clock64()
#else
#define GN_PP_NOW() 0ull
#endif
template <int HF, bool LEVEL0>
__global__ __launch_bounds__(256) GN_PACK_ATTR void gn_hibf_pack_kernel(GnHibfLevelParams p)
{
    // Dynamic LDS, one region per wave: the 16 byte-counter registers of every lane (multi-bin runs, 4 KB) -- and, on launches with
    // narrow classes, the staged minimisers of the wave's items (GN_HIBF_NQ per item; the image is written after the row loop, when
    // the staged values are done with, and the next batch is staged after the image's readers)
    extern __shared__ uint64_t gn_pack_lds[];
    const uint32_t lane   = threadIdx.x & (GN_WAVE - 1);
    const uint32_t wave   = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t region = p.lds_region; // 8-byte words per wave (gn_hibf_pack_region)
    uint64_t* const hst   = gn_pack_lds + (size_t)wave * region;
    const uint32_t stride = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_id = (uint32_t)blockIdx.x * (blockDim.x >> 6) + wave;

#ifdef GN_PACK_PROF
    const unsigned long long pp_wall0 = wall_clock64();
#endif
    GnHibfAppender     app;
    unsigned long long my_bytes = 0, my_lines = 0;
    uint32_t           rot = 0; // batches of the classes before this one (mod the grid's waves): every class starts where the last one ended,
                                // so a small class does not land on the same few waves as the small class before it

  const uint32_t n_cls = LEVEL0 || p.n_cls == 0 ? 1u : p.n_cls;
  // The class with the most batches comes last: it is the one whose batches are handed out on demand (below), so whatever lead or lag a
  // wave has collected in the classes before is evened out there.
  uint32_t big = 0;
  if (n_cls > 1)
  {
      unsigned long long most = 0;
      for (uint32_t c = 0; c < n_cls; ++c)
      {
          const unsigned long long b = (p.cls_count[c] + (GN_WAVE >> p.cls_gp[c]) - 1) >> (6u - p.cls_gp[c]);
          if (b > most)
          {
              most = b;
              big  = c;
          }
      }
  }
  for (uint32_t ci = 0; ci < n_cls; ++ci)
  {
    const uint32_t cls    = ci + 1 == n_cls ? big : (ci < big ? ci : ci + 1);
    const uint32_t gpl    = (LEVEL0 || p.n_cls == 0) ? p.pack_gp : (uint32_t)p.cls_gp[cls];
    const uint32_t Gp     = 1u << gpl;
    const uint32_t H      = GN_WAVE >> gpl;          // items per wave
    const uint32_t gl     = lane & (Gp - 1);         // my word of the row
    const uint32_t grp    = lane >> gpl;             // my item of the batch
    const uint32_t gbase  = lane & ~(Gp - 1);        // first lane of my group
    const bool     share  = Gp >= (uint32_t)HF;      // lane i of a group hashes function i for the whole group
    const uint64_t my_seed = GN_HIBF_SEEDS[gl < (uint32_t)HF ? gl : 0u];

    uint32_t           n_work;
    unsigned long long work_first = 0;
    if constexpr (LEVEL0)
        n_work = p.n_reads;
    else
    {
        const unsigned long long nw64 = p.n_cls ? p.cls_count[cls] : *p.count_in;
        n_work                        = (uint32_t)(nw64 < p.work_cap ? nw64 : p.work_cap);
        if (p.n_cls || p.work_base)
        {
            work_first = p.n_cls ? p.cls_base[cls] : *p.work_base;
            if (work_first + n_work > p.work_cap)
                n_work = work_first < p.work_cap ? (uint32_t)(p.work_cap - work_first) : 0u;
        }
    }
    const uint32_t n_batches = (n_work + H - 1) / H;
    const uint32_t first     = wave_id >= rot ? wave_id - rot : wave_id + stride - rot;
    rot                      = (uint32_t)(((uint64_t)rot + n_batches) % stride);
    // Which batches a wave takes.  Up to round 5 wave i took batches i, i + waves, ... of every class: the same count for every wave -- and
    // the waves of a launch ended between 0.59 and 1.0 of its duration (level 1 of the skewed tree: first wave through after 4.9 ms, the
    // median after 6.4, the last after 8.3; level 0: 8.3 / 9.3 / 9.9, the XCDs with odd numbers 11 % behind the even ones:
    // profiles/r05_pack_prof_skew.txt).  That stays for the small classes.  In the LAST class of a launch (the largest; level 0 has one) a
    // wave starts with a chunk of its own and takes every later chunk from the class's cursor -- asked for one chunk ahead, so that the
    // answer is there when it is needed; about six chunks a wave, half-size ones for the last quarter of the class, quarter-size ones for
    // its last eighth.  (A counter address
    // takes ~90 atomics a microsecond: chunks, not batches, and not for classes of a few batches a wave -- tried: level 2 lost 0.35 ms.)
    const bool dyn = ci + 1 == n_cls && n_batches >= 8u * stride && p.grab != nullptr;
    uint32_t   G0  = dyn ? n_batches / (stride * 6u) : 1u;
    G0             = G0 < 1u ? 1u : (G0 > 16u ? 16u : G0);
    uint32_t batch = first * G0;
    if (batch >= n_batches)
        continue;
    uint32_t       end       = batch + G0 < n_batches ? batch + G0 : n_batches;
    const uint32_t dyn_from  = stride * G0; // the cursor counts from here
    const uint32_t tail_from = n_batches - n_batches / 4u, tail2_from = n_batches - n_batches / 8u;
    unsigned long long* const gctr = p.grab ? p.grab + cls : nullptr;
    uint32_t g_next = 0, g_size = G0; // (lane 0: what the cursor answered; the size asked for)
    auto     ask    = [&](uint32_t from) {
        g_size = from < tail_from ? G0 : (from < tail2_from ? (G0 >= 2u ? G0 / 2u : 1u) : (G0 >= 8u ? G0 / 4u : (G0 >= 2u ? 2u : 1u)));
        if (lane == 0)
            g_next = (uint32_t)atomicAdd(gctr, (unsigned long long)g_size);
    };
    if (dyn)
        ask(batch);

    // per-lane view of an item (the Gp lanes of a group hold identical copies)
    struct Item
    {
        uint32_t read, ibf, n, W, shift, S, nm; // nm: multi-bin runs of the item's IBF
        uint64_t slot;
        const __attribute__((address_space(1))) uint64_t* rows;
        bool     ok;    // counted here
        bool     defer; // valid, but left to the per-item kernel
    };
    auto load_item = [&](uint32_t b) -> Item {
        Item           m{};
        const uint32_t idx = b * H + grp;
        uint32_t       read = 0xFFFFFFFFu, ibf = 0;
        if (b < n_batches && idx < n_work)
        {
            if constexpr (LEVEL0)
                read = (p.status[idx + p.read_base] == GN_READ_OK || (p.wide && p.status[idx + p.read_base] == GN_READ_BIG))
                           ? idx + p.read_base
                           : 0xFFFFFFFFu;
            else
            {
                const uint2 e = p.work_in[work_first + idx];
                read = e.x;
                ibf  = e.y;
            }
        }
        m.read = read;
        m.ibf  = ibf;
        if (read != 0xFFFFFFFFu)
        {
            const GnHibfIbfDev* f = p.ibfs + ibf;
            m.n     = p.n_hashes[read];
            m.slot  = p.slot_off[read];
            m.W     = f->W;
            m.S     = (uint32_t)f->S;
            m.shift = f->shift;
            m.rows  = gn_global(f->rows);
            uint32_t g = m.W <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(m.W - 1);
            m.nm    = f->n_mruns;
            m.ok    = m.W <= GN_WAVE && g == gpl && m.n >= 1 && m.n <= 127u;
            m.defer = !m.ok && m.n >= 1;
        }
        return m;
    };

#ifdef GN_PACK_PROF
    unsigned long long pp_row = 0, pp_stage = 0, pp_tail = 0, pp_batches = 0, pp_iters = 0, pp_items = 0, pp_lines0 = my_lines;
    const unsigned long long pp_t0 = GN_PP_NOW();
#endif
    Item cur = load_item(batch);
    for (;;)
    {
        uint32_t next_batch = batch + 1u;
        if (next_batch >= end)
        {
            next_batch = batch + stride; // (a small class: batches i, i + waves, ...)
            end        = next_batch + 1u;
            if (dyn)
            {
                next_batch = dyn_from + (uint32_t)__builtin_amdgcn_readfirstlane((int)g_next); // (asked for a chunk ago)
                end        = next_batch + g_size < n_batches ? next_batch + g_size : n_batches;
                if (next_batch < n_batches)
                    ask(next_batch);
            }
        }
        const Item nxt = load_item(next_batch); // its loads fly behind this batch's row loop

        // items this kernel does not count: one entry per group (lane gl == 0) on the per-item kernel's list
        {
            const uint64_t dm = __ballot(cur.defer && gl == 0);
            if (dm)
            {
                unsigned long long base = 0;
                if (lane == 0)
                    base = atomicAdd(p.defer_count, (unsigned long long)__popcll(dm));
                base = gn_hibf_bcast64(base);
                if (cur.defer && gl == 0)
                {
                    const unsigned long long o = base + __popcll(dm & ((1ULL << lane) - 1ULL));
                    if (o < p.work_cap)
                        p.defer_out[o] = make_uint2(cur.read, cur.ibf);
                }
            }
        }

        const uint32_t n     = cur.ok ? cur.n : 0u;
        const bool     col   = gl < cur.W;
        const uint32_t gl_ld = col ? gl : 0u;
        uint32_t       nib[2][4], byt[2][4][2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                nib[d][j]    = 0;
                byt[d][j][0] = 0;
                byt[d][j][1] = 0;
            }
        // longest item of the batch (wave-uniform trip count)
        uint32_t n_max = n;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const uint32_t o = (uint32_t)__shfl_xor((int)n_max, off);
            n_max            = o > n_max ? o : n_max;
        }
        n_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_max);

        struct Rows
        {
            uint64_t m[HF];
        };
        const uint64_t hs_at = cur.slot; // (index into p.hashes)
        const uint64_t fake0 = p.fake_hashes ? p.hashes[n ? hs_at : 0ull] : 0ull;
        const bool     staged = p.stage_hashes && gpl >= 1 && gpl <= 2; // (wave-uniform; see gn_hibf_pack_region)
#ifdef GN_PACK_PROF
        const unsigned long long pp_t1 = GN_PP_NOW();
#endif
        if (staged)
        {
            // lane gl of a group takes the hashes q = gl, gl + Gp, ...; every load is issued before the first one is waited for
            gn_hibf_wave_sync(); // (the previous batch's image readers are through)
            // One load instruction per ITEM: lanes 0 .. 23 fetch the item's hashes 0 .. 23 -- 192 contiguous bytes, two or three line
            // requests -- instead of every lane (group) fetching its own item's hashes one by one (a request per hash: what limits a
            // level of narrow IBFs is the number of line requests, profiles/README.md).  Eight items in flight per round.
            const uint32_t lim  = n < GN_HIBF_NQ ? n : GN_HIBF_NQ; // (mine; the item loop reads the others' through readlane)
            const auto*    hsrc = gn_global(p.hashes);
            for (uint32_t i0 = 0; i0 < H; i0 += 8u)
            {
                uint64_t tmp[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                {
                    const uint32_t src_lane = (i0 + j) << gpl; // (i0 + j < H: H is 16, 32 or 64)
                    const uint32_t lim_i    = (uint32_t)__builtin_amdgcn_readlane((int)lim, (int)src_lane);
                    const uint64_t at_i     = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hs_at >> 32), (int)src_lane) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hs_at, (int)src_lane);
                    tmp[j] = lane < lim_i ? hsrc[at_i + lane] : 0ull;
                }
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                {
                    const uint32_t src_lane = (i0 + j) << gpl;
                    const uint32_t lim_i    = (uint32_t)__builtin_amdgcn_readlane((int)lim, (int)src_lane);
                    if (lane < lim_i)
                        hst[(i0 + j) * GN_HIBF_NQ_STRIDE + lane] = tmp[j];
                }
            }
            gn_hibf_wave_sync();
        }
#ifdef GN_PACK_PROF
        const unsigned long long pp_t2 = GN_PP_NOW();
        pp_stage += pp_t2 - pp_t1;
#endif
        // hash `it` of my item, requested two iterations before the rows that depend on it: the row loop's chain per iteration is then one
        // memory latency (the rows), not two (hash, then rows) -- taken out of the chain in an experiment a level of narrow IBFs ran 21 %
        // faster (profiles/r05_hibf_probe_fake2.jsonl)
        auto fetch = [&](uint32_t it) -> uint64_t {
            if (!(it < n || (p.reread && n)))
                return 0ull;
            const uint32_t q = it < n ? it : n - 1;
            if (p.fake_hashes) // timing experiment: one load per item, the other "hashes" derived from it
                return fake0 * (2ull * q + 1ull);
            if (staged && q < GN_HIBF_NQ)
                return hst[grp * GN_HIBF_NQ_STRIDE + q];
            return p.hashes[hs_at + q];
        };
        auto issue = [&](uint32_t it, Rows& R, uint64_t v) {
            // Lanes whose item is through (or is not counted here) issue nothing.  They used to re-read their last row "for free": with 64
            // items a wave the longest item has 1.29 x the mean number of minimisers, and a re-read row was long gone from the L1 and L2 --
            // a quarter of the level's row requests went to the fabric for nothing.  (A group's lanes share n: the branch is group-uniform,
            // the permutes below stay inside the group.)
            if (it < n || (p.reread && n))
            {
                // (A/B switch hibf_reread: finished lanes read their last row again, as up to round 4 -- fetch() repeats the last hash)
                uint32_t row[HF];
                if (share)
                {
                    const uint32_t mine = gn_hibf_row_seed(v, my_seed, cur.shift, cur.S);
#pragma unroll
                    for (int i = 0; i < HF; ++i)
                        row[i] = (uint32_t)__shfl((int)mine, (int)(gbase + (uint32_t)i));
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < HF; ++i)
                        row[i] = gn_hibf_row_seed(v, GN_HIBF_SEEDS[i], cur.shift, cur.S);
                }
#pragma unroll
                for (int i = 0; i < HF; ++i)
                    R.m[i] = cur.rows[(uint64_t)row[i] * cur.W + gl_ld];
            }
        };
        uint32_t acc_n = 0;
        auto consume = [&](const Rows& R, uint32_t it) {
            uint64_t a = R.m[0];
#pragma unroll
            for (int i = 1; i < HF; ++i)
                a &= R.m[i];
            if (!(col && it < n))
                a = 0;
            const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                nib[0][j] += (a0 >> j) & 0x11111111u;
                nib[1][j] += (a1 >> j) & 0x11111111u;
            }
            if (++acc_n == 15)
            {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        byt[d][j][0] += nib[d][j] & 0x0F0F0F0Fu;
                        byt[d][j][1] += (nib[d][j] >> 4) & 0x0F0F0F0Fu;
                        nib[d][j] = 0;
                    }
                acc_n = 0;
            }
        };
        if (n_max)
        {
            Rows     A, B;
            uint64_t ha = fetch(0), hb = fetch(1);
            issue(0, A, ha);
            ha          = fetch(2);
            uint32_t it = 0;
            for (; it + 2 < n_max; it += 2)
            {
                issue(it + 1, B, hb);
                hb = fetch(it + 3);
                consume(A, it);
                issue(it + 2, A, ha);
                ha = fetch(it + 4);
                consume(B, it + 1);
            }
            const bool two = it + 1 < n_max;
            if (two)
                issue(it + 1, B, hb);
            consume(A, it);
            if (two)
                consume(B, it + 1);
        }
#ifdef GN_PACK_PROF
        const unsigned long long pp_t3 = GN_PP_NOW();
        pp_row += pp_t3 - pp_t2;
        ++pp_batches;
        pp_iters += n_max;
        pp_items += gl == 0 ? n : 0u;
#endif
        if (n && gl == 0)
        {
            my_bytes += (unsigned long long)n * HF * gn_global(p.ibfs)[cur.ibf].Wl * 8ull; // algorithmic bytes of this visit (once per item)
            my_lines += (unsigned long long)n * HF * ((cur.W * 8ull + 127ull) & ~127ull); // ... in the 128-byte lines the rows occupy
        }

        // ---- threshold my own 64 bins; every bin of these IBFs is a run of its own (:445-458 with a run of length one) ----
        // threshold_cutoff = max(1, ceil(n * rel_cutoff))  (GanonClassify.cpp:492-495,720-724); passed to bulk_count (:553)
        uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
        if (T == 0)
            T = 1;
        const uint32_t Kc = (0x80u - T) * 0x01010101u; // counts <= n <= 127, 1 <= T <= 127: no carries between bytes
        uint32_t       c0 = 0, c1 = 0;                 // candidate bits of my two dwords
        if (n && col)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                {
                    byt[0][j][pp] += (nib[0][j] >> (4 * pp)) & 0x0F0F0F0Fu;
                    byt[1][j][pp] += (nib[1][j] >> (4 * pp)) & 0x0F0F0F0Fu;
                    c0 |= (((byt[0][j][pp] + Kc) & 0x80808080u) >> 7) << (4 * pp + j);
                    c1 |= (((byt[1][j][pp] + Kc) & 0x80808080u) >> 7) << (4 * pp + j);
                }
        }
        if (__ballot((c0 | c1) != 0))
        {
            const auto* bin_tab = (c0 | c1) ? gn_global(p.ibfs[cur.ibf].bin_tab) : gn_global((const uint32_t*)nullptr);
            do
            {
                bool     merged = false, leaf = false;
                uint32_t tgt = 0, sum = 0;
                if (c0 | c1)
                {
                    const uint32_t d = c0 ? 0u : 1u;
                    uint32_t&      c = c0 ? c0 : c1;
                    const uint32_t t = (uint32_t)__builtin_ctz(c);
                    c &= c - 1;
                    const uint32_t y = t >> 3, pp = (t >> 2) & 1u, j = t & 3u;
                    uint32_t       reg = 0;
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq)
                                if ((uint32_t)dd == d && (uint32_t)jj == j && (uint32_t)qq == pp)
                                    reg = byt[dd][jj][qq];
                    sum                = (reg >> (8 * y)) & 0xFFu;
                    const uint32_t tab = bin_tab[gl * 64 + 32 * d + t];
                    if (tab != 0xFFFFFFFFu)
                    {
                        merged = (tab & 0x80000000u) != 0;
                        leaf   = !merged;
                        tgt    = tab & 0x7FFFFFFFu;
                    }
                }
                app.push(p, (int)lane, merged, leaf, cur.read, tgt, sum);
            } while (__ballot((c0 | c1) != 0));
        }

        // ---- multi-bin runs (split user bins, :445-458 with the running sum over the run): the group's lanes share the item's runs ----
        const uint32_t nm = cur.ok ? cur.nm : 0u;
        if (__ballot(nm != 0))
        {
            uint32_t* img = reinterpret_cast<uint32_t*>(hst);
            gn_hibf_wave_sync(); // (the readers of the batch before are through)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        img[lane * 16 + (d * 4 + j) * 2 + pp] = (n && col) ? byt[d][j][pp] : 0u;
            gn_hibf_wave_sync();
            const auto* mruns = nm ? gn_global(p.ibfs[cur.ibf].mruns) : gn_global((const uint4*)nullptr);
            for (uint32_t r = gl; __ballot(r < nm); r += Gp)
            {
                bool     leaf = false;
                uint32_t tgt = 0, sum = 0;
                if (r < nm)
                {
                    const uint4 run = *reinterpret_cast<const uint4*>((uintptr_t)(mruns + r)); // first bin, n bins, user bin, -
                    for (uint32_t b = run.x; b < run.x + run.y; ++b)
                    {
                        const uint32_t t = b & 63u, d = t >> 5, tt = t & 31u;
                        const uint32_t v = img[(gbase + (b >> 6)) * 16 + (d * 4 + (tt & 3u)) * 2 + ((tt >> 2) & 1u)];
                        sum              = sum + ((v >> (8 * (tt >> 3))) & 0xFFu);
                        sum              = p.wide ? sum : (sum & 0xFFFFu); // value_t = uint16_t wraps (hibf.hpp:438,442)
                    }
                    leaf = sum >= T; // :455
                    tgt  = run.z;
                }
                app.push(p, (int)lane, false, leaf, cur.read, tgt, sum);
            }
        }

#ifdef GN_PACK_PROF
        pp_tail += GN_PP_NOW() - pp_t3;
#endif
        batch = next_batch;
        if (batch >= n_batches)
            break;
        cur = nxt;
    }
#ifdef GN_PACK_PROF
    {
        unsigned long long it_sum = pp_items, ln = my_lines - pp_lines0;
        for (int off = 32; off >= 1; off >>= 1)
        {
            it_sum += __shfl_xor(it_sum, off);
            ln += __shfl_xor(ln, off);
        }
        if (lane == 0)
        {
            atomicAdd(&gn_pack_prof_buf[cls][0], (unsigned long long)(GN_PP_NOW() - pp_t0));
            atomicAdd(&gn_pack_prof_buf[cls][1], pp_row);
            atomicAdd(&gn_pack_prof_buf[cls][2], pp_stage);
            atomicAdd(&gn_pack_prof_buf[cls][3], pp_tail);
            atomicAdd(&gn_pack_prof_buf[cls][4], pp_batches);
            atomicAdd(&gn_pack_prof_buf[cls][5], pp_iters);
            atomicAdd(&gn_pack_prof_buf[cls][6], it_sum);
            atomicAdd(&gn_pack_prof_buf[cls][7], ln);
        }
    }
#endif
  } // classes
#ifdef GN_PACK_PROF
    if (lane == 0 && wave_id < 8192u)
    {
        gn_pack_prof_wave[wave_id][0] = pp_wall0;
        gn_pack_prof_wave[wave_id][1] = wall_clock64();
    }
#endif
    app.finish(p, (int)lane);
    // one atomic per wave
    for (int off = 32; off >= 1; off >>= 1)
    {
        my_bytes += __shfl_xor(my_bytes, off);
        my_lines += __shfl_xor(my_lines, off);
    }
    if (lane == 0 && my_bytes)
    {
        atomicAdd(&p.ctr[2], my_bytes);
        atomicAdd(&p.ctr[1], my_lines);
        atomicAdd(p.lvl_bytes, my_bytes);
        atomicAdd(p.lvl_lines, my_lines);
    }
}

// ================================================================================================
// register-counter level kernel
// ================================================================================================
#define GN_HIBF_REG_NMAX 127u
#define GN_HIBF_REG_WMAX 64u

// x + (x rotated right by N lanes inside its row of 16 lanes): one DPP-modified add, no LDS crossbar
template <int N>
__device__ __forceinline__ uint32_t gn_hibf_add_ror(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 + N, 0xF, 0xF, false);
}

// LEVEL0: the items are the reads of the batch themselves (read i against IBF 0, if its status is OK) -- no queue, no
// seeding pass.  Deeper levels read (read, ibf) entries that the previous level appended.
template <int HF, bool LEVEL0>
__global__ __launch_bounds__(256) void gn_hibf_reg_kernel(GnHibfLevelParams p)
{
    __shared__ uint32_t gn_img[4][GN_WAVE * 16]; // per wave: the byte counters of the owner lanes (multi-bin runs only)
    const int      lane   = threadIdx.x & (GN_WAVE - 1);
    const int      wave   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t stride = gridDim.x * (blockDim.x >> 6);
    uint32_t*      img    = gn_img[wave];

    uint32_t n_work;
    if constexpr (LEVEL0)
        n_work = p.n_reads;
    else
    {
        const unsigned long long nw64 = *p.count_in;
        n_work                        = (uint32_t)(nw64 < p.work_cap ? nw64 : p.work_cap);
    }
    uint32_t item = (uint32_t)blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave;
    if (item >= n_work)
        return;

    GnHibfAppender     app;
    unsigned long long my_bytes = 0, my_lines = 0;

    // ---- software pipeline over a wave's items (item, item + stride, ...) ------------------------------------------
    // Everything an item needs arrives before the item is reached: its queue entry is fetched four items ahead, its
    // metadata (n, hash slot, IBF shape; wave-uniform -> scalar loads into SGPRs) three ahead, its first hashes two
    // ahead, and its first two row sets are requested one item ahead -- while the previous item's counters are being
    // reduced and thresholded.  Nothing is waited for at the point where it is requested.
    struct Meta
    {
        uint32_t        read, ibf, n, W, shift, n_mruns, S;
        uint64_t        slot;
        const uint64_t* rows;
        bool            valid;
    };
    // (item indices are wave-uniform: block index and an SGPR wave index)
    auto load_entry = [&](uint32_t it) -> uint2 {
        if (it >= n_work)
            return make_uint2(0xFFFFFFFFu, 0u);
        if constexpr (LEVEL0)
        {
            const uint32_t rd   = it + p.read_base;
            const uint32_t four = gn_sload(reinterpret_cast<const uint32_t*>(p.status) + (rd >> 2)); // status bytes rd&~3 ..
            const uint32_t st1  = (four >> (8u * (rd & 3u))) & 0xFFu;
            return make_uint2((st1 == GN_READ_OK || (p.wide && st1 == GN_READ_BIG)) ? rd : 0xFFFFFFFFu, 0u);
        }
        else
        {
            const uint64_t e = gn_sload(reinterpret_cast<const uint64_t*>(p.work_in) + it);
            return make_uint2((uint32_t)e, (uint32_t)(e >> 32));
        }
    };
    auto load_meta = [&](uint2 wk) -> Meta {
        Meta m{};
        m.read  = wk.x;
        m.ibf   = wk.y;
        m.valid = m.read != 0xFFFFFFFFu; // hole left by a chunked append of the previous level / read that is not counted
        if (m.valid)
        {
            const GnHibfIbfDev* f = p.ibfs + m.ibf;
            m.n       = gn_sload(p.n_hashes + m.read);
            m.slot    = gn_sload(p.slot_off + m.read);
            m.W       = gn_sload(&f->W);
            m.S       = (uint32_t)gn_sload(&f->S); // (bin_size < 2^32 is checked at upload)
            m.shift   = gn_sload(&f->shift);
            m.rows    = gn_sload(&f->rows);
            m.n_mruns = gn_sload(&f->n_mruns);
        }
        return m;
    };
    auto gp_of = [](uint32_t W) -> uint32_t { // lanes per row = 1 << gp
        uint32_t g = W <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(W - 1);
        return g < 6 ? g : 6u;
    };
    // items this kernel counts itself: rows of at most one wave, counters that fit (4-bit per lane in the loop: at most
    // 15 iterations; 8-bit after the groups are added: at most 127 minimisers)
    auto reg_ok = [&](const Meta& m) -> bool {
        return m.valid && m.W <= GN_HIBF_REG_WMAX && m.n >= 1 && m.n <= GN_HIBF_REG_NMAX && m.n <= 15u * (GN_WAVE >> gp_of(m.W));
    };
    // the hashes of the first two iterations of an item (q = hsub, H + hsub)
    auto load_hashes = [&](const Meta& m, uint64_t& h0, uint64_t& h1) {
        h0 = h1 = 0;
        if (reg_ok(m))
        {
            const uint32_t g = gp_of(m.W), H = GN_WAVE >> g, hs = (uint32_t)lane >> g;
            const uint32_t q0 = hs < m.n ? hs : m.n - 1, q1 = H + hs < m.n ? H + hs : m.n - 1;
            h0 = p.hashes[m.slot + q0];
            h1 = p.hashes[m.slot + q1];
        }
    };
    struct Rows
    {
        uint2 m[HF];
    };
    // h row requests of hash value v for item m; issued by every lane unconditionally (lanes without a column read
    // word 0 of the row and are masked when the rows are consumed, see gn_ibf_count_fast_kernel)
    auto issue = [&](const Meta& m, uint64_t v, Rows& R) {
        const uint32_t g = gp_of(m.W), gl = (uint32_t)lane & ((1u << g) - 1u);
        const uint32_t gl_ld = gl < m.W ? gl : 0u;
#pragma unroll
        for (int i = 0; i < HF; ++i)
        {
            const uint32_t row = gn_hibf_row(v, (uint32_t)i, m.shift, (uint64_t)m.S);
            const auto* src    = gn_global(m.rows) + ((uint64_t)row * m.W + gl_ld);
            R.m[i]             = make_uint2((uint32_t)*src, (uint32_t)(*src >> 32));
        }
    };

    Meta  m0 = load_meta(load_entry(item)), m1 = load_meta(load_entry(item + stride)), m2 = load_meta(load_entry(item + 2 * stride));
    uint2 wk3 = load_entry(item + 3 * stride);
    uint64_t hA, hB, h1A, h1B;
    Rows     RA{}, RB{};
    load_hashes(m0, hA, hB);
    load_hashes(m1, h1A, h1B);
    if (reg_ok(m0))
    {
        issue(m0, hA, RA);
        issue(m0, hB, RB);
    }

    for (;;)
    {
        const bool more = item + stride < n_work;
        Meta       m3   = load_meta(wk3);                       // three ahead (its entry arrived during the previous item)
        wk3             = load_entry(item + 4 * stride);        // four ahead
        uint64_t h2A = 0, h2B = 0;

        const bool     mine = reg_ok(m0);
        uint32_t       nib[2][4];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nib[d][j] = 0;
        const uint32_t gpl  = gp_of(m0.W);
        const uint32_t Gp   = 1u << gpl;
        const uint32_t H    = GN_WAVE >> gpl;
        const uint32_t gl   = (uint32_t)lane & (Gp - 1);
        const uint32_t hsub = (uint32_t)lane >> gpl;
        const bool     col_act = gl < m0.W;
        const uint32_t n    = m0.n;

        if (m0.valid && !mine)
        {
            if (n >= 1 && lane == 0) // left to the LDS-counter kernel of this level
            {
                const unsigned long long o = atomicAdd(p.defer_count, 1ULL);
                if (o < p.work_cap)
                    p.defer_out[o] = make_uint2(m0.read, m0.ibf);
            }
        }
        else if (mine)
        {
            auto consume = [&](const Rows& R, uint32_t it) {
                const uint32_t on = (col_act && it * H + hsub < n) ? 0xFFFFFFFFu : 0u;
                uint32_t       a0 = R.m[0].x & on, a1 = R.m[0].y & on;
#pragma unroll
                for (int i = 1; i < HF; ++i)
                {
                    a0 &= R.m[i].x;
                    a1 &= R.m[i].y;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    nib[0][j] += (a0 >> j) & 0x11111111u;
                    nib[1][j] += (a1 >> j) & 0x11111111u;
                }
            };
            const uint32_t  iters = (n + H - 1) / H; // <= 15
            const uint64_t* hs    = p.hashes + m0.slot;
            auto hash_of = [&](uint32_t it) -> uint64_t {
                const uint32_t q = it * H + hsub;
                return hs[q < n ? q : n - 1];
            };
            // RA / RB hold iterations 0 / 1 (requested while the previous item was finished)
            for (uint32_t it = 0; it < iters; it += 2)
            {
                consume(RA, it);
                if (it + 2 < iters)
                    issue(m0, hash_of(it + 2), RA);
                if (it + 1 < iters)
                    consume(RB, it + 1);
                if (it + 3 < iters)
                    issue(m0, hash_of(it + 3), RB);
            }
            my_bytes += (unsigned long long)n * HF * gn_sload(&p.ibfs[m0.ibf].Wl) * 8ull; // algorithmic bytes of this visit
            my_lines += (unsigned long long)n * HF * ((m0.W * 8ull + 127ull) & ~127ull);
        }
        // the row registers are free: request the next item's first two row sets, then the hashes of the one after
        if (more && reg_ok(m1))
        {
            issue(m1, h1A, RA);
            issue(m1, h1B, RB);
        }
        if (more)
            load_hashes(m2, h2A, h2B);

        if (mine)
        {
            // ---- epilogue: add the H hash groups, SWAR compare with the cutoff ----
            // threshold_cutoff = max(1, ceil(n * rel_cutoff))  (GanonClassify.cpp:492-495,720-724); passed to bulk_count (:553)
            uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
            if (T == 0)
                T = 1;
            const uint32_t Kc  = (0x80u - T) * 0x01010101u; // counts <= n <= 127, 1 <= T <= 127: no carries between bytes
            uint32_t       byt[2][4][2];
            uint32_t       any = 0;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                    {
                        uint32_t x = (nib[d][j] >> (4 * pp)) & 0x0F0F0F0Fu;
                        // groups inside a row of 16 lanes: DPP rotations; across rows: the LDS crossbar
                        if (Gp <= 1)
                            x = gn_hibf_add_ror<1>(x);
                        if (Gp <= 2)
                            x = gn_hibf_add_ror<2>(x);
                        if (Gp <= 4)
                            x = gn_hibf_add_ror<4>(x);
                        if (Gp <= 8)
                            x = gn_hibf_add_ror<8>(x);
                        if (Gp <= 16)
                            x += __shfl_xor(x, 16);
                        if (Gp <= 32)
                            x += __shfl_xor(x, 32);
                        byt[d][j][pp] = x;
                        any |= (x + Kc) & 0x80808080u;
                    }
            const bool owner = hsub == 0 && col_act;
            // single-bin runs: every bin with count >= T is a hit (sum of one bin; :445-458 with a run of length one)
            uint32_t c0 = 0, c1 = 0; // candidate bits of the lane's two dwords
            if (owner && any)
            {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                    {
                        c0 |= (((byt[0][j][pp] + Kc) & 0x80808080u) >> 7) << (4 * pp + j);
                        c1 |= (((byt[1][j][pp] + Kc) & 0x80808080u) >> 7) << (4 * pp + j);
                    }
            }
            if (__ballot((c0 | c1) != 0))
            {
                const auto* bin_tab = gn_global(gn_sload(&p.ibfs[m0.ibf].bin_tab));
                do
                {
                    bool     merged = false, leaf = false;
                    uint32_t tgt = 0, sum = 0;
                    if (c0 | c1)
                    {
                        const uint32_t d = c0 ? 0u : 1u;
                        uint32_t&      c = c0 ? c0 : c1;
                        const uint32_t t = (uint32_t)__builtin_ctz(c);
                        c &= c - 1;
                        const uint32_t y = t >> 3, pp = (t >> 2) & 1u, j = t & 3u;
                        uint32_t       reg = 0;
#pragma unroll
                        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int qq = 0; qq < 2; ++qq)
                                    if ((uint32_t)dd == d && (uint32_t)jj == j && (uint32_t)qq == pp)
                                        reg = byt[dd][jj][qq];
                        sum                = (reg >> (8 * y)) & 0xFFu;
                        const uint32_t tab = bin_tab[gl * 64 + 32 * d + t];
                        if (tab != 0xFFFFFFFFu)
                        {
                            merged = (tab & 0x80000000u) != 0;
                            leaf   = !merged;
                            tgt    = tab & 0x7FFFFFFFu;
                        }
                    }
                    app.push(p, lane, merged, leaf, m0.read, tgt, sum);
                } while (__ballot((c0 | c1) != 0));
            }
            // multi-bin runs (split user bins): sums from an image of the owner lanes' byte counters
            if (m0.n_mruns)
            {
                const auto* mruns = gn_global(gn_sload(&p.ibfs[m0.ibf].mruns));
                gn_hibf_wave_sync();
                if (owner)
                {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int pp = 0; pp < 2; ++pp)
                                img[gl * 16 + (d * 4 + j) * 2 + pp] = byt[d][j][pp];
                }
                gn_hibf_wave_sync();
                for (uint32_t r0 = 0; r0 < m0.n_mruns; r0 += GN_WAVE)
                {
                    const uint32_t r = r0 + (uint32_t)lane;
                    bool           leaf = false;
                    uint32_t       tgt = 0, sum = 0;
                    if (r < m0.n_mruns)
                    {
                        const uint4 run = *reinterpret_cast<const uint4*>((uintptr_t)(mruns + r)); // first bin, n bins, user bin, -
                        for (uint32_t b = run.x; b < run.x + run.y; ++b)
                        {
                            const uint32_t t = b & 63u, d = t >> 5, tt = t & 31u;
                            const uint32_t v = img[(b >> 6) * 16 + (d * 4 + (tt & 3u)) * 2 + ((tt >> 2) & 1u)];
                            sum              = sum + ((v >> (8 * (tt >> 3))) & 0xFFu);
                            sum              = p.wide ? sum : (sum & 0xFFFFu); // value_t = uint16_t wraps (hibf.hpp:438,442)
                        }
                        leaf = sum >= T; // :455
                        tgt  = run.z;
                    }
                    app.push(p, lane, false, leaf, m0.read, tgt, sum);
                }
            }
        }

        if (!more)
            break;
        item += stride;
        m0  = m1;
        m1  = m2;
        m2  = m3;
        h1A = h2A;
        h1B = h2B;
    }
    app.finish(p, lane);
    if (lane == 0 && my_bytes)
    {
        atomicAdd(&p.ctr[2], my_bytes);
        atomicAdd(&p.ctr[1], my_lines);
        atomicAdd(p.lvl_bytes, my_bytes);
        atomicAdd(p.lvl_lines, my_lines);
    }
}

// ================================================================================================
// LDS-counter level kernel (IBFs wider than 64 words, reads with more than 127 minimisers)
// ================================================================================================
__global__ __launch_bounds__(256) void gn_hibf_level_kernel(GnHibfLevelParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gn_hl[];
    const int      lane   = threadIdx.x & (GN_WAVE - 1);
    const int      wave   = threadIdx.x >> 6; // (kept in a VGPR: making it an SGPR slowed this kernel, 5.2 -> 7.5 ms)
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    uint32_t*      cnt    = gn_hl + (size_t)wave * p.lds_bins;

    unsigned long long nw64   = *p.count_in;
    const uint32_t     n_work = (uint32_t)(nw64 < p.work_cap ? nw64 : p.work_cap);

    GnHibfAppender     app;
    unsigned long long my_bytes = 0, my_lines = 0;

    for (uint32_t item = blockIdx.x * (blockDim.x >> 6) + wave; item < n_work; item += nwaves)
    {
        const uint2    wk   = p.work_in[item];
        const uint32_t read = wk.x;
        if (read == 0xFFFFFFFFu) // hole left by a chunked append of the previous level
            continue;
        const GnHibfIbfDev f  = p.ibfs[wk.y];
        const uint32_t     n  = p.n_hashes[read];
        const uint64_t*    hs = p.hashes + p.slot_off[read];
        const uint32_t     TB = f.W * 64;

        gn_hibf_wave_sync();
        for (uint32_t i = lane; i < TB; i += GN_WAVE)
            cnt[i] = 0;
        gn_hibf_wave_sync();

        // lanes = (hash sub-index, word): Gp lanes cover the W words of a row (W <= 64), or the row is walked in
        // 64-word chunks with one hash per iteration (W > 64)
        uint32_t gp_log2 = 0;
        while ((1u << gp_log2) < f.W && gp_log2 < 6)
            ++gp_log2;
        const uint32_t Gp   = 1u << gp_log2;
        const uint32_t H    = GN_WAVE >> gp_log2;
        const uint32_t gl   = lane & (Gp - 1);
        const uint32_t hsub = lane >> gp_log2;

        for (uint32_t q0 = 0; q0 < n; q0 += H)
        {
            const uint32_t q = q0 + hsub;
            if (q < n)
            {
                const uint64_t v = hs[q];
                uint32_t       rows[5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    rows[i] = (uint32_t)i < f.h ? gn_hibf_row(v, i, f.shift, f.S) : 0u;
                for (uint32_t wd = gl; wd < f.W; wd += Gp)
                {
                    uint64_t m = ~0ULL;
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if ((uint32_t)i < f.h)
                            m &= f.rows[(uint64_t)rows[i] * f.W + wd];
                    while (m)
                    {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        atomicAdd(&cnt[wd * 64 + b], 1u);
                    }
                }
            }
        }
        gn_hibf_wave_sync();

        // threshold_cutoff = max(1, ceil(n * rel_cutoff))  (GanonClassify.cpp:492-495,720-724); passed to bulk_count (:553)
        uint32_t T = (uint32_t)(uint64_t)ceil(__dmul_rn((double)n, p.rel_cutoff));
        if (T == 0)
            T = 1;

        for (uint32_t r0 = 0; r0 < f.n_runs; r0 += GN_WAVE)
        {
            const uint32_t r = r0 + lane;
            bool           hit = false, merged = false;
            uint32_t       sum = 0;
            int32_t        tgt = 0;
            if (r < f.n_runs)
            {
                const uint4 run = f.runs[r]; // first bin, n bins, user bin (-1 merged), child ibf
                for (uint32_t b = 0; b < run.y; ++b)
                    sum = p.wide ? sum + cnt[run.x + b] : ((sum + cnt[run.x + b]) & 0xFFFFu); // value_t = uint16_t wraps (hibf.hpp:438,442)
                merged = (int32_t)run.z < 0;
                tgt    = merged ? (int32_t)run.w : (int32_t)run.z;
                hit    = sum >= T; // :447 / :455
            }
            app.push(p, lane, hit && merged, hit && !merged, read, (uint32_t)tgt, sum);
        }
        my_bytes += (unsigned long long)n * f.h * f.Wl * 8ull; // algorithmic bytes of this visit
        my_lines += (unsigned long long)n * f.h * ((f.W * 8ull + 127ull) & ~127ull);
    }
    app.finish(p, lane);
    if (lane == 0 && my_bytes)
    {
        atomicAdd(&p.ctr[2], my_bytes);
        atomicAdd(&p.ctr[1], my_lines);
        atomicAdd(p.lvl_bytes, my_bytes);
        atomicAdd(p.lvl_lines, my_lines);
    }
}

__global__ void gn_hibf_seed_kernel(uint2* work, const uint8_t* status, uint32_t read_base, uint32_t n_reads, unsigned long long* count,
                                    uint32_t wide)
{
    const uint32_t i  = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r  = read_base + i;
    const bool     ok = i < n_reads && (status[r] == GN_READ_OK || (wide && status[r] == GN_READ_BIG));
    const uint64_t bm = __ballot(ok);
    const int      lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0 && bm)
        base = atomicAdd(count, (unsigned long long)__popcll(bm));
    base = gn_hibf_bcast64(base);
    if (ok)
        work[base + __popcll(bm & ((1ULL << lane) - 1ULL))] = make_uint2(r, 0u);
}

// sorted (key, raw sum) -> gn_match with the cap of select_matches (GanonClassify.cpp:561-564) + per-read histogram
// (out_base: matches of the read ranges before this one; the valid pairs of a sorted range are its first entries)
__global__ __launch_bounds__(256) void gn_hibf_finish_kernel(const uint64_t* keys, const uint32_t* vals, uint64_t n, uint32_t ub_bits,
                                                             const uint32_t* n_hashes, gn_match* out_all, uint32_t* seg_count,
                                                             const unsigned long long* out_base, uint64_t out_cap)
{
    gn_match* out = out_all + *out_base;
    if (*out_base + n > out_cap) // (the host sees the same and has the batch run again with more room)
        return;
    const uint64_t i    = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t key  = i < n ? keys[i] : ~0ULL;
    const bool     valid = key != ~0ULL; // (chunk holes are sorted to the end: the valid pairs of a wave are its first lanes)
    const uint32_t read = valid ? (uint32_t)(key >> ub_bits) : 0xFFFFFFFFu;
    if (valid)
    {
        const uint32_t nh = n_hashes[read];
        gn_match       m;
        m.read   = read;
        m.target = (uint32_t)(key & ((1ULL << ub_bits) - 1ULL));
        m.count  = vals[i] > nh ? nh : vals[i];
        out[i]   = m;
    }
    // histogram per read: the pairs are sorted, so a wave adds one number per run of equal reads instead of one per pair
    // (thousands of pairs per read at low cutoffs: the per-pair atomics on one address were 35 % of such a batch)
    const uint32_t prev  = (uint32_t)__shfl_up((int)read, 1);
    const bool     start = valid && (lane == 0 || prev != read);
    const uint64_t sm = __ballot(start), vm = __ballot(valid);
    if (start)
    {
        const uint64_t above = lane == 63 ? 0ULL : (sm & ~((2ULL << lane) - 1ULL));
        const uint32_t end   = above ? (uint32_t)__builtin_ctzll(above) : (uint32_t)__popcll(vm);
        atomicAdd(&seg_count[read], end - lane);
    }
}

// after a range's finish: the next range's matches begin behind this one's valid pairs (holes are sorted to the end)
__global__ void gn_hibf_advance_kernel(const uint64_t* keys, uint64_t n, unsigned long long* out_base)
{
    uint64_t lo = 0, hi = n; // first index whose key is the all-ones sentinel
    while (lo < hi)
    {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] == ~0ULL)
            hi = mid;
        else
            lo = mid + 1;
    }
    *out_base += lo;
}

// ---- matches that a filter_matches pre-pass is bound to drop never reach the sort --------------------------------------------
// With gn_stream_set_postfilter on the stream (gn_postfilter.hip) most of what the level kernels append at a low --rel-cutoff
// goes again a moment later, after a radix sort over all of it.  The rule is the fast kernel's (gn_kernels.hip, epilogue):
// the read's --rel-filter threshold cannot be below t2 = threshold(largest count of the read, lower bound of its minimum), so
// pairs under t2 are counted, their smallest count is kept per read (seg_min: the read's minimum is over everything that passed
// the cutoff) and only the rest is copied -- in wave-private chunks, holes = all-ones keys like the appender's -- to the buffer
// the sort then reads.  Here the maximum is the read's true one: a first pass over the raw pairs collects it.
__global__ void gn_hibf_premax_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, uint32_t ub_bits,
                                      const uint32_t* __restrict__ n_hashes, uint32_t* rmax)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k = keys[i];
        if (k == ~0ULL)
            continue;
        const uint32_t read = (uint32_t)(k >> ub_bits), nh = n_hashes[read];
        const uint32_t cv   = vals[i] > nh ? nh : vals[i];
        if (cv > rmax[read]) // (a stale value only costs an atomic too many: the maximum never goes down)
            atomicMax(&rmax[read], cv);
    }
}

__global__ __launch_bounds__(256) void gn_hibf_predrop_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n,
                                                              uint32_t ub_bits, const uint32_t* __restrict__ n_hashes,
                                                              const uint32_t* __restrict__ rmax, double rel_cutoff, uint32_t pre_mode,
                                                              double pre_rel, uint64_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint64_t cap, unsigned long long* cursor,
                                                              uint32_t* seg_min, unsigned long long* pre_ctr)
{
    const uint32_t     lane = threadIdx.x & 63u;
    const uint64_t     wid  = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long cbase = 0;
    uint32_t           cleft = 0, n_drop = 0;
    auto               close = [&]() {
        for (uint32_t x = lane; x < cleft; x += 64)
            if (cbase + x < cap)
                keys_out[cbase + x] = ~0ULL;
        cleft = 0;
    };
    for (uint64_t b = wid * 64; b < n; b += nw * 64) // (wave-uniform trip count)
    {
        const uint64_t i = b + lane;
        uint64_t       k = ~0ULL;
        uint32_t       v = 0;
        if (i < n)
        {
            k = keys[i];
            v = vals[i];
        }
        bool keep = false;
        if (k != ~0ULL)
        {
            const uint32_t read = (uint32_t)(k >> ub_bits), nh = n_hashes[read];
            const uint32_t cv   = v > nh ? nh : v;
            uint32_t       lb   = 0;
            if (pre_mode == 1) // this filter sees all of the read's matches: none is below the cutoff count
            {
                lb = (uint32_t)(uint64_t)ceil(__dmul_rn((double)nh, rel_cutoff));
                lb = lb ? lb : 1u;
            }
            const uint32_t mx = rmax[read];
            const uint32_t t2 = mx >= lb ? gn_pf_threshold(mx, lb, pre_rel) : 0u;
            keep = cv >= t2;
            if (!keep)
            {
                ++n_drop;
                if (cv < seg_min[read])
                    atomicMin(&seg_min[read], cv);
            }
        }
        const uint64_t km = __ballot(keep);
        if (km)
        {
            const uint32_t need = (uint32_t)__popcll(km);
            if (need > cleft)
            {
                close();
                unsigned long long nb = 0;
                if (lane == 0)
                    nb = atomicAdd(cursor, (unsigned long long)GN_HIBF_CHUNK);
                cbase = gn_hibf_bcast64(nb);
                cleft = GN_HIBF_CHUNK;
            }
            if (keep)
            {
                const unsigned long long o = cbase + __popcll(km & ((1ULL << lane) - 1ULL));
                if (o < cap)
                {
                    keys_out[o] = k;
                    vals_out[o] = v;
                }
            }
            cbase += need;
            cleft -= need;
        }
    }
    close();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        n_drop += (uint32_t)__shfl_xor((int)n_drop, off);
    if (lane == 0 && n_drop)
        atomicAdd(pre_ctr, (unsigned long long)n_drop);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static uint32_t gn_bits_for(uint64_t v) // smallest b with v < 2^b
{
    uint32_t b = 0;
    while (b < 64 && (v >> b))
        ++b;
    return b;
}

int gn_hibf_build(gn_filter* f, uint32_t n_ibf, const gn_ibf_desc* ibfs, const int64_t* const* next_ibf_id,
                  const int64_t* const* bin2userbin, uint64_t n_user_bins)
{
    if (n_user_bins >= (1ull << 31))
        return gn_fail(GN_ERANGE, "more than 2^31 user bins");
    std::vector<GnHibfIbfDev>          dev(n_ibf);
    std::vector<std::vector<uint32_t>> children(n_ibf);
    uint32_t                           max_tb = 0;
    for (uint32_t i = 0; i < n_ibf; ++i)
    {
        const uint64_t B = ibfs[i].bins;
        std::vector<uint4>    runs, mruns;
        std::vector<uint32_t> tab((size_t)f->ibfs[i].Ws * 64, 0xFFFFFFFFu); // (the kernels see the padded row: its extra bins lead nowhere)
        uint64_t b = 0;
        while (b < B)
        {
            const int64_t u = bin2userbin[i][b];
            if (u >= (int64_t)n_user_bins)
                return gn_fail(GN_EINVAL, "ibf %u bin %llu: user bin %lld out of range", i, (unsigned long long)b, (long long)u);
            if (u < 0)
            {
                const int64_t c = next_ibf_id[i][b];
                if (c < 0 || c >= (int64_t)n_ibf || c == (int64_t)i)
                    return gn_fail(GN_EINVAL, "ibf %u bin %llu: merged bin without a valid child ibf (%lld)", i,
                                   (unsigned long long)b, (long long)c);
                runs.push_back(make_uint4((uint32_t)b, 1u, 0xFFFFFFFFu, (uint32_t)c));
                tab[b] = 0x80000000u | (uint32_t)c;
                children[i].push_back((uint32_t)c);
                ++b;
            }
            else
            {
                uint64_t e = b + 1;
                while (e < B && bin2userbin[i][e] == u)
                    ++e;
                runs.push_back(make_uint4((uint32_t)b, (uint32_t)(e - b), (uint32_t)u, 0u));
                if (e - b == 1)
                    tab[b] = (uint32_t)u;
                else
                    mruns.push_back(runs.back());
                b = e;
            }
        }
        uint4*    d_runs  = nullptr;
        uint4*    d_mruns = nullptr;
        uint32_t* d_tab   = nullptr;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&d_runs), std::max<size_t>(1, runs.size()) * sizeof(uint4)));
        f->hibf_allocs.push_back(d_runs);
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&d_mruns), std::max<size_t>(1, mruns.size()) * sizeof(uint4)));
        f->hibf_allocs.push_back(d_mruns);
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&d_tab), tab.size() * sizeof(uint32_t)));
        f->hibf_allocs.push_back(d_tab);
        if (!runs.empty())
            GN_HIP(hipMemcpy(d_runs, runs.data(), runs.size() * sizeof(uint4), hipMemcpyHostToDevice));
        if (!mruns.empty())
            GN_HIP(hipMemcpy(d_mruns, mruns.data(), mruns.size() * sizeof(uint4), hipMemcpyHostToDevice));
        GN_HIP(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        dev[i].rows    = f->ibfs[i].d_rows;
        dev[i].S       = f->ibfs[i].S;
        dev[i].W       = (uint32_t)f->ibfs[i].Ws; // width = stride on the device: the words a row is padded with are zero
        dev[i].Wl      = (uint32_t)f->ibfs[i].W;
        dev[i].B       = (uint32_t)f->ibfs[i].B;
        dev[i].shift   = f->ibfs[i].shift;
        dev[i].h       = f->ibfs[i].h;
        dev[i].runs    = d_runs;
        dev[i].n_runs  = (uint32_t)runs.size();
        dev[i].bin_tab = d_tab;
        dev[i].mruns   = d_mruns;
        dev[i].n_mruns = (uint32_t)mruns.size();
        max_tb         = std::max(max_tb, dev[i].W * 64u);
        if (dev[i].h != dev[0].h)
            return gn_fail(GN_ERANGE, "IBFs of one HIBF with different numbers of hash functions (%u vs %u)", dev[i].h, dev[0].h);
    }
    if ((size_t)max_tb * 4 > 144 * 1024)
        return gn_fail(GN_ERANGE, "an IBF of the HIBF has %u technical bins; the level kernel supports up to 36864", max_tb);
    // depth of the tree below IBF 0 = number of levels a batch can take (the level kernels are launched that many times)
    {
        std::vector<uint32_t> depth(n_ibf, 0);
        std::vector<uint32_t> stack{ 0u };
        depth[0]        = 1;
        uint32_t deepest = 1;
        while (!stack.empty())
        {
            const uint32_t i = stack.back();
            stack.pop_back();
            for (uint32_t c : children[i])
            {
                if (depth[i] + 1 > GN_HIBF_MAXDEPTH)
                    return gn_fail(GN_EINVAL, "HIBF deeper than %d levels (cycle in next_ibf_id?)", GN_HIBF_MAXDEPTH);
                if (depth[c] < depth[i] + 1)
                {
                    depth[c] = depth[i] + 1;
                    deepest  = std::max(deepest, depth[c]);
                    stack.push_back(c);
                }
            }
        }
        f->max_depth = deepest;
        // lanes per row that most IBFs of a level have: what the packed kernel of that level is launched for
        f->level_gp.assign(deepest, 0u);
        std::vector<std::vector<uint32_t>> votes(deepest, std::vector<uint32_t>(7, 0u));
        for (uint32_t i = 0; i < n_ibf; ++i)
            if (depth[i] >= 1)
            {
                uint32_t g = 0;
                while ((1u << g) < dev[i].W && g < 6)
                    ++g;
                votes[depth[i] - 1][g]++;
            }
        f->level_gps.assign(deepest, std::vector<uint32_t>());
        for (uint32_t l = 0; l < deepest; ++l)
        {
            f->level_gp[l] = (uint32_t)(std::max_element(votes[l].begin(), votes[l].end()) - votes[l].begin());
            // every lane width the level's IBFs have, most common first: the packed kernel runs once per width, each pass on what
            // the pass before left (raptor's lower levels mix IBFs of 64 ... 1024 bins)
            std::vector<uint32_t> order;
            for (uint32_t g = 0; g < 7; ++g)
                if (votes[l][g])
                    order.push_back(g);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return votes[l][a] > votes[l][b]; });
            f->level_gps[l] = order;
        }
        f->level_bytes.assign(deepest, 0ull);
        f->level_row_bytes.assign(deepest, 0u);
        for (uint32_t i = 0; i < n_ibf; ++i)
            if (depth[i] >= 1)
                f->level_bytes[depth[i] - 1] += dev[i].S * (uint64_t)dev[i].W * 8ull;
        for (uint32_t l = 0; l < deepest; ++l)
            f->level_row_bytes[l] = 8u << f->level_gp[l];
    }
    GN_HIP(hipMalloc(reinterpret_cast<void**>(&f->d_hibf), n_ibf * sizeof(GnHibfIbfDev)));
    GN_HIP(hipMemcpy(f->d_hibf, dev.data(), n_ibf * sizeof(GnHibfIbfDev), hipMemcpyHostToDevice));
    f->n_user_bins = n_user_bins;
    f->max_bins    = max_tb;
    return GN_OK;
}

static int gn_hibf_ensure_sort_buffers(gn_stream* s)
{
    if (s->hibf_cap >= s->match_cap && s->d_keys[0])
        return GN_OK;
    for (int i = 0; i < 2; ++i)
    {
        if (s->d_keys[i])
            hipFree(s->d_keys[i]);
        if (s->d_vals[i])
            hipFree(s->d_vals[i]);
        s->d_keys[i] = nullptr;
        s->d_vals[i] = nullptr;
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_keys[i]), std::max<uint64_t>(1, s->match_cap) * 8));
        GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_vals[i]), std::max<uint64_t>(1, s->match_cap) * 4));
    }
    if (s->d_sort_tmp)
        hipFree(s->d_sort_tmp);
    s->d_sort_tmp = nullptr;
    size_t tmp    = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, s->d_keys[0], s->d_keys[1], s->d_vals[0], s->d_vals[1],
                                       (int)std::min<uint64_t>(s->match_cap, 0x7FFFFFFFull), 0, 64, s->st);
    s->sort_tmp_bytes = tmp + 256;
    GN_HIP(hipMalloc(&s->d_sort_tmp, s->sort_tmp_bytes));
    s->hibf_cap = s->match_cap;
    return GN_OK;
}

// persistent grid = what is resident at once (the register budget decides)
template <int HF, bool LEVEL0>
static void gn_hibf_launch_reg2(const GnHibfLevelParams& p, uint32_t n_cu, uint32_t bpc, hipStream_t st)
{
    if (bpc == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_hibf_reg_kernel<HF, LEVEL0>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        bpc = (uint32_t)per_cu;
    }
    hipLaunchKernelGGL((gn_hibf_reg_kernel<HF, LEVEL0>), dim3(n_cu * bpc), dim3(256), 0, st, p);
}
template <int HF>
static void gn_hibf_launch_reg(const GnHibfLevelParams& p, bool level0, uint32_t n_cu, uint32_t bpc, hipStream_t st)
{
    if (level0)
        gn_hibf_launch_reg2<HF, true>(p, n_cu, bpc, st);
    else
        gn_hibf_launch_reg2<HF, false>(p, n_cu, bpc, st);
}

// dynamic LDS of a packed launch: per wave the 4 KB counter image, or the staged hashes of the items a wave of the narrowest class holds
static uint32_t gn_hibf_pack_region(GnHibfLevelParams& p, bool level0)
{
    // Staged are the classes of two and four lanes per row (32 and 16 items a wave: 6.4 and 3.2 KB of LDS a wave).  One lane per row
    // would need 12.8 KB a wave -- three workgroups a CU instead of four, which costs more than the staging gains there (measured:
    // profiles/r05_probe3_skew*.jsonl); those classes prefetch their hashes two iterations ahead instead.
    uint32_t min_gp = 8;
    auto     take   = [&](uint32_t g) {
        if (g >= 1 && g <= 2)
            min_gp = std::min(min_gp, g);
    };
    if (level0 || p.n_cls == 0)
        take(p.pack_gp);
    else
        for (uint32_t c = 0; c < p.n_cls; ++c)
            take(p.cls_gp[c]);
    p.stage_hashes = min_gp <= 2 && !gn_sw().hibf_stage ? 1u : 0u;
    p.lds_region   = std::max<uint32_t>(GN_WAVE * 8u, p.stage_hashes ? (GN_WAVE >> min_gp) * GN_HIBF_NQ_STRIDE : 0u);
    return p.lds_region * 8u * 4u; // bytes per workgroup of four waves
}

template <int HF, bool LEVEL0>
static void gn_hibf_launch_pack2(GnHibfLevelParams p, uint32_t n_cu, uint32_t bpc, hipStream_t st)
{
    const uint32_t lds = gn_hibf_pack_region(p, LEVEL0);
    if (bpc == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_hibf_pack_kernel<HF, LEVEL0>, 256, lds) != hipSuccess || per_cu < 1)
            per_cu = 2;
        bpc = (uint32_t)per_cu;
    }
    hipLaunchKernelGGL((gn_hibf_pack_kernel<HF, LEVEL0>), dim3(n_cu * bpc), dim3(256), lds, st, p);
#ifdef GN_PACK_PROF
    {
        unsigned long long h[8][8] = {};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gn_pack_prof_buf), sizeof(h));
        const uint32_t nc = LEVEL0 || p.n_cls == 0 ? 1u : p.n_cls;
        for (uint32_t c = 0; c < nc; ++c)
            fprintf(stderr, "[pack prof] level0=%d waves=%u cls=%u gp=%u wave_cycles=%llu row=%llu stage=%llu tail=%llu batches=%llu iters=%llu item_iters=%llu lines=%llu\n",
                    (int)LEVEL0, n_cu * bpc * 4u, c, (LEVEL0 || p.n_cls == 0) ? p.pack_gp : (uint32_t)p.cls_gp[c], h[c][0], h[c][1], h[c][2], h[c][3], h[c][4],
                    h[c][5], h[c][6], h[c][7]);
        unsigned long long z[8][8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gn_pack_prof_buf), z, sizeof(z));
        {
            static std::vector<unsigned long long> w(8192 * 2);
            (void)hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(gn_pack_prof_wave), w.size() * 8);
            const uint32_t nw = std::min<uint32_t>(n_cu * bpc * 4u, 8192u);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (uint32_t i = 0; i < nw; ++i)
                if (w[2 * i + 1])
                {
                    t0 = std::min(t0, w[2 * i]);
                    t1 = std::max(t1, w[2 * i + 1]);
                }
            std::vector<double> ends;
            double              xs[8] = {}, xm[8] = {};
            uint32_t            xn[8] = {};
            for (uint32_t i = 0; i < nw; ++i)
                if (w[2 * i + 1])
                {
                    const double e = (double)(w[2 * i + 1] - t0) / 100.0; // microseconds
                    ends.push_back(e);
                    const uint32_t x = (i / 4u) % 8u; // workgroup -> XCD (round robin)
                    xs[x] += e;
                    xm[x] = std::max(xm[x], e);
                    ++xn[x];
                }
            std::sort(ends.begin(), ends.end());
            if (!ends.empty())
            {
                fprintf(stderr, "[pack prof] level0=%d wave end times (us): span %.1f  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f;  per XCD mean/max:", (int)LEVEL0,
                        (double)(t1 - t0) / 100.0, ends.front(), ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back());
                for (int x = 0; x < 8; ++x)
                    fprintf(stderr, " %.0f/%.0f", xn[x] ? xs[x] / xn[x] : 0.0, xm[x]);
                fprintf(stderr, "\n");
            }
            std::fill(w.begin(), w.end(), 0ull);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(gn_pack_prof_wave), w.data(), w.size() * 8);
        }
    }
#endif
}
template <int HF>
static void gn_hibf_launch_pack(const GnHibfLevelParams& p, bool level0, uint32_t n_cu, uint32_t bpc, hipStream_t st)
{
    if (level0)
        gn_hibf_launch_pack2<HF, true>(p, n_cu, bpc, st);
    else
        gn_hibf_launch_pack2<HF, false>(p, n_cu, bpc, st);
}

// ---- a level's items sorted by the lane width of their IBF ------------------------------------------------------------------
// raptor's lower levels mix IBFs of 64 ... 1024 technical bins; the packed kernel wants every item of a launch to have the same
// lanes per row.  Two passes over the level's queue (8 bytes per item): count per class, then scatter -- class c = the c-th most
// common width of the level (cls_of_gp), class 7 = what the packed kernel does not take (wider than 64 words, more than 127
// minimisers), which goes straight to the per-item kernels' list.  Holes of the chunked queue are dropped on the way.
struct GnHibfBucketParams
{
    const GnHibfIbfDev*       ibfs;
    const uint32_t*           n_hashes;
    const uint2*              work_in;
    const unsigned long long* count_in;
    uint32_t                  work_cap;
    uint8_t                   cls_of_gp[8];
    unsigned long long*       cls_count; // [8]
    unsigned long long*       cls_base;  // [8] exclusive prefix of cls_count[0..6]
    unsigned long long*       cls_cursor; // [8]
    uint2*                    sorted_out;
    uint2*                    rest_out;  // class 7
    unsigned long long*       rest_count;
};

__device__ __forceinline__ uint32_t gn_hibf_item_class(const GnHibfBucketParams& p, uint2 e)
{
    if (e.x == 0xFFFFFFFFu)
        return 8u; // a hole
    const uint32_t W = p.ibfs[e.y].W, n = p.n_hashes[e.x];
    if (n == 0)
        return 8u;
    if (W > GN_WAVE || n > 127u)
        return 7u;
    const uint32_t g = W <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(W - 1);
    return p.cls_of_gp[g];
}

// A workgroup takes the queue in chunks of 4096 items; per chunk ONE atomic per class (a counter address sustains only ~90 atomics
// per microsecond: per-wave atomics made these two passes the most expensive kernels of the level).
#define GN_HIBF_BUCKET_ROUNDS 16u
template <bool SCATTER>
__global__ __launch_bounds__(256) void gn_hibf_bucket_kernel(GnHibfBucketParams p)
{
    __shared__ uint32_t           wave_cnt[4][8];
    __shared__ unsigned long long chunk_base[8];
    const unsigned long long      nw64 = *p.count_in;
    const uint32_t                n    = (uint32_t)(nw64 < p.work_cap ? nw64 : p.work_cap);
    const uint32_t                lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t                chunk = 256u * GN_HIBF_BUCKET_ROUNDS;
    const uint32_t                n_chunks = (n + chunk - 1) / chunk;
    for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        const uint32_t first = ch * chunk + wave * 64u * GN_HIBF_BUCKET_ROUNDS; // this wave's 1024 items
        uint32_t       mine[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            mine[c] = 0;
        for (uint32_t k = 0; k < GN_HIBF_BUCKET_ROUNDS; ++k)
        {
            const uint32_t i = first + k * 64u + lane;
            const uint32_t c = gn_hibf_item_class(p, i < n ? p.work_in[i] : make_uint2(0xFFFFFFFFu, 0u));
#pragma unroll
            for (uint32_t cc = 0; cc < 8; ++cc)
                mine[cc] += (uint32_t)__popcll(__ballot(c == cc)); // (wave-uniform)
        }
        if (lane < 8)
        {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t cc = 0; cc < 8; ++cc)
                v = lane == cc ? mine[cc] : v;
            wave_cnt[wave][lane] = v;
        }
        __syncthreads();
        if (threadIdx.x < 8)
        {
            const uint32_t c     = threadIdx.x;
            const uint32_t total = wave_cnt[0][c] + wave_cnt[1][c] + wave_cnt[2][c] + wave_cnt[3][c];
            if (!SCATTER)
            {
                if (total)
                    atomicAdd(&p.cls_count[c], (unsigned long long)total);
            }
            else
                chunk_base[c] = total ? atomicAdd(c == 7 ? p.rest_count : &p.cls_cursor[c], (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        if (SCATTER)
        {
            unsigned long long at[8];
#pragma unroll
            for (uint32_t cc = 0; cc < 8; ++cc)
            {
                at[cc] = chunk_base[cc] + (cc == 7 ? 0ull : p.cls_base[cc]);
                for (uint32_t w = 0; w < wave; ++w)
                    at[cc] += wave_cnt[w][cc];
            }
            for (uint32_t k = 0; k < GN_HIBF_BUCKET_ROUNDS; ++k)
            {
                const uint32_t i = first + k * 64u + lane;
                const uint2    e = i < n ? p.work_in[i] : make_uint2(0xFFFFFFFFu, 0u);
                const uint32_t c = gn_hibf_item_class(p, e);
#pragma unroll
                for (uint32_t cc = 0; cc < 8; ++cc)
                {
                    const uint64_t m = __ballot(c == cc);
                    if (c == cc)
                    {
                        const unsigned long long o = at[cc] + __popcll(m & ((1ULL << lane) - 1ULL));
                        if (cc == 7)
                        {
                            if (o < p.work_cap)
                                p.rest_out[o] = e;
                        }
                        else
                            p.sorted_out[o] = e;
                    }
                    at[cc] += __popcll(m);
                }
            }
        }
        __syncthreads(); // (wave_cnt / chunk_base are reused by the next chunk)
    }
}

// ---- ... and, inside a width class, by the number of minimisers ---------------------------------------------------------------------
// A wave of the packed kernel holds up to 64 items and runs as many iterations as its LONGEST item has minimisers: reads of 150 bp have
// 12 .. 25 (mean 17.6), the longest of 64 has 22.6 -- in a fifth of a wave's iterations part of its lanes have nothing in flight, on
// levels that are bound by the requests a wave keeps outstanding.  The same two passes therefore sort by (class, n / 2): 16 bins of
// two, everything from 30 minimisers up in the last.  Counting and scattering go through an LDS histogram per 4096-item chunk (one
// global atomic per key and chunk); the order inside a key is whatever the LDS atomics give -- the matches are sorted at the end anyway.
#define GN_HIBF_NBINS 16u
#define GN_HIBF_NKEYS (7u * GN_HIBF_NBINS) // key 112 = what the packed kernel does not take (class 7), 113 = holes
struct GnHibfSubParams
{
    unsigned long long* sub_count;  // [128] of this level
    unsigned long long* sub_base;   // [128]
    unsigned long long* sub_cursor; // [128]
};

__device__ __forceinline__ uint32_t gn_hibf_item_key(const GnHibfBucketParams& p, uint2 e)
{
    if (e.x == 0xFFFFFFFFu)
        return GN_HIBF_NKEYS + 1u;
    const uint32_t W = p.ibfs[e.y].W, n = p.n_hashes[e.x];
    if (n == 0)
        return GN_HIBF_NKEYS + 1u;
    if (W > GN_WAVE || n > 127u)
        return GN_HIBF_NKEYS;
    const uint32_t g = W <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(W - 1);
    const uint32_t c = p.cls_of_gp[g];
    if (c >= 7u)
        return GN_HIBF_NKEYS;
    const uint32_t nb = (n >> 1) < GN_HIBF_NBINS ? (n >> 1) : GN_HIBF_NBINS - 1u;
    return c * GN_HIBF_NBINS + nb;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void gn_hibf_nsort_kernel(GnHibfBucketParams p, GnHibfSubParams q)
{
    __shared__ uint32_t           hist[128];
    __shared__ unsigned long long chunk_base[128];
    const unsigned long long      nw64 = *p.count_in;
    const uint32_t                n    = (uint32_t)(nw64 < p.work_cap ? nw64 : p.work_cap);
    const uint32_t                chunk = 256u * GN_HIBF_BUCKET_ROUNDS;
    const uint32_t                n_chunks = (n + chunk - 1) / chunk;
    for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        if (threadIdx.x < 128)
            hist[threadIdx.x] = 0;
        __syncthreads();
        uint2    mine[GN_HIBF_BUCKET_ROUNDS];
        uint8_t  key[GN_HIBF_BUCKET_ROUNDS];
#pragma unroll
        for (uint32_t k = 0; k < GN_HIBF_BUCKET_ROUNDS; ++k)
        {
            const uint32_t i = ch * chunk + k * 256u + threadIdx.x;
            mine[k]          = i < n ? p.work_in[i] : make_uint2(0xFFFFFFFFu, 0u);
        }
#pragma unroll
        for (uint32_t k = 0; k < GN_HIBF_BUCKET_ROUNDS; ++k)
        {
            key[k] = (uint8_t)gn_hibf_item_key(p, mine[k]);
            if (key[k] <= GN_HIBF_NKEYS)
                atomicAdd(&hist[key[k]], 1u);
        }
        __syncthreads();
        if (threadIdx.x <= GN_HIBF_NKEYS)
        {
            const uint32_t k = threadIdx.x, total = hist[k];
            if (!SCATTER)
            {
                if (total)
                    atomicAdd(&q.sub_count[k], (unsigned long long)total);
            }
            else
            {
                chunk_base[k] = total ? (k == GN_HIBF_NKEYS ? atomicAdd(p.rest_count, (unsigned long long)total)
                                                            : q.sub_base[k] + atomicAdd(&q.sub_cursor[k], (unsigned long long)total))
                                      : 0ull;
                hist[k] = 0; // (the chunk's cursor inside the key from here on)
            }
        }
        __syncthreads();
        if (SCATTER)
        {
#pragma unroll
            for (uint32_t k = 0; k < GN_HIBF_BUCKET_ROUNDS; ++k)
                if (key[k] <= GN_HIBF_NKEYS)
                {
                    const unsigned long long o = chunk_base[key[k]] + atomicAdd(&hist[key[k]], 1u);
                    if (key[k] == GN_HIBF_NKEYS)
                    {
                        if (o < p.work_cap)
                            p.rest_out[o] = mine[k];
                    }
                    else
                        p.sorted_out[o] = mine[k];
                }
            __syncthreads();
        }
    }
}

// bases of the 112 (class, n-bin) keys in key order, and the eight per-class counts / bases the packed kernel reads
__global__ void gn_hibf_nsort_bases_kernel(const unsigned long long* __restrict__ sub_count, unsigned long long* __restrict__ sub_base,
                                           unsigned long long* __restrict__ cls_count, unsigned long long* __restrict__ cls_base)
{
    unsigned long long at = 0;
    for (uint32_t c = 0; c < 7; ++c)
    {
        cls_base[c]           = at;
        unsigned long long in = 0;
        for (uint32_t b = 0; b < GN_HIBF_NBINS; ++b)
        {
            sub_base[c * GN_HIBF_NBINS + b] = at;
            at += sub_count[c * GN_HIBF_NBINS + b];
            in += sub_count[c * GN_HIBF_NBINS + b];
        }
        cls_count[c] = in;
    }
    cls_base[7]  = at;
    cls_count[7] = sub_count[GN_HIBF_NKEYS];
}

__global__ void gn_hibf_bucket_bases_kernel(const unsigned long long* __restrict__ cnt, unsigned long long* __restrict__ base)
{
    unsigned long long at = 0;
    for (int c = 0; c < 7; ++c)
    {
        base[c] = at;
        at += cnt[c];
    }
    base[7] = at;
}

// Runs all levels back to back (queue lengths stay on the device), synchronises ONCE, then sorts/groups the matches.
int gn_finish_batch(gn_stream* s); // gn_capi.hip

// One HIBF batch.  The raw (read, user bin) pairs of a batch must fit the pair buffers before they are pre-dropped and sorted,
// and the radix sort counts its items in an int: at low cutoffs (thousands of chance pairs per read: 27 G pairs for 10 M reads
// against 65 536 user bins at --rel-cutoff 0.2) a batch holds more than either allows.  Such a batch is run in READ RANGES:
// every range goes through the levels, the pre-drop, the sort and the finish on its own, appending its matches behind the
// previous range's (ranges ascend, so the result is grouped by read as ever).  A range whose pairs exceed what memory or the
// sort can take is halved; one that merely exceeds the current buffers makes the caller grow them (gn_finish) as before.
int gn_hibf_classify(gn_stream* s, gn_filter* f, hipStream_t st)
{
    int rc = gn_hibf_ensure_sort_buffers(s);
    if (rc)
        return rc;
    const uint32_t n      = s->n_reads;
    const uint32_t depth  = f->max_depth ? f->max_depth : 1;
    const uint32_t NL     = GN_HIBF_MAXDEPTH + 1;
    const uint32_t ub_bits = std::max(1u, gn_bits_for(f->n_user_bins ? f->n_user_bins - 1 : 0));
    const uint32_t rd_bits = std::max(1u, gn_bits_for(n)); // a read index is < n <= 2^rd_bits - 1: below the all-ones sentinel
    if (ub_bits + rd_bits > 64)
        return gn_fail(GN_ERANGE, "read index and user bin do not fit one 64-bit sort key");
    GN_HIP(hipMemsetAsync(s->d_ctr + 1, 0, 3 * sizeof(unsigned long long), st)); // line bytes, algo bytes, (unused)
    GN_HIP(hipMemsetAsync(s->d_ctr + 6, 0, sizeof(unsigned long long), st));     // exact match count
    GN_HIP(hipMemsetAsync(s->d_hctr, 0, (37 * NL + 2) * sizeof(unsigned long long), st)); // queues, per-level bytes, [4NL] output base, [4NL+2..] per-level line bytes, [5NL+2..] per level: 8 class counts, bases, cursors, [29NL+2..] batch cursors of the packed kernel
    GN_HIP(hipMemsetAsync(s->d_seg_count, 0, ((size_t)n + 1) * 4, st));
    unsigned long long* d_out_base = s->d_hctr + 4 * NL;
    const uint32_t h      = f->ibfs[0].h;
    // tests / A-B: skip the packed kernel (the per-item register kernel takes whole levels), or both (LDS kernel only)
    const bool     no_reg  = gn_sw().hibf_reg;
    const bool     no_pack = no_reg || gn_sw().hibf_pack;
    const uint32_t reg_bpc = gn_sw().hibf_bpc; // workgroups per CU of the register kernels: 0 = what the occupancy query says
    // with a filter_matches pre-pass on the stream, what it is bound to drop does not reach the sort
    const bool may_predrop = s->pf_on && !s->pf_merge && s->d_pf_segmin && s->d_pf_rmax && s->pf_rel_filter >= 0.0 && s->pf_rel_filter < 1.0 &&
                             (uint64_t)n + 1 <= s->pf_segmin_cap && !gn_sw().predrop;
    s->pf_predrop = false;
    if (may_predrop)
    {
        GN_HIP(hipMemsetAsync(s->d_pf_rmax, 0, ((size_t)n + 1) * 4, st));
        GN_HIP(hipMemsetAsync(s->d_pf_segmin, 0xFF, ((size_t)n + 1) * 4, st));
        GN_HIP(hipMemsetAsync(s->d_pf_pre, 0, 2 * sizeof(unsigned long long), st)); // [0] pairs left out [1] output cursor
    }
    // pairs one range may have: the sort's int, and what the device could hold if the buffers were grown for it (two pair
    // buffers, two match buffers, the sort's scratch: ~64 bytes a pair) -- switch hibf_pair_limit=N for tests
    uint64_t pair_limit = 0x7FFFFFF0ull;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess)
            pair_limit = std::min<uint64_t>(pair_limit, std::max<uint64_t>(1u << 20, ((uint64_t)fr + s->match_cap * 64ull) / 80ull));
        if (gn_sw().hibf_pair_limit)
            pair_limit = gn_sw().hibf_pair_limit;
    }

    // the levels of reads [lo, lo + cnt): queue lengths and the match cursor come back to the host (one sync)
    auto run_levels = [&](uint32_t lo, uint32_t cnt, bool stamp) -> int {
        GN_HIP(hipMemsetAsync(s->d_ctr, 0, sizeof(unsigned long long), st));
        GN_HIP(hipMemsetAsync(s->d_hctr, 0, 3 * NL * sizeof(unsigned long long), st));
        GN_HIP(hipMemsetAsync(s->d_hctr + 5 * NL + 2, 0, 32 * NL * sizeof(unsigned long long), st));
        if (s->d_hsub)
            GN_HIP(hipMemsetAsync(s->d_hsub, 0, (size_t)NL * 384 * sizeof(unsigned long long), st));
        if (cnt && no_reg) // (the register-counter kernels take level 0 straight from the batch)
            hipLaunchKernelGGL(gn_hibf_seed_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, s->d_work[0], s->v_status, lo, cnt, s->d_hctr,
                               s->long_reads ? 1u : 0u);
        for (uint32_t lvl = 0; lvl < depth && cnt; ++lvl)
        {
            if (stamp && lvl < GN_HIBF_TIMED_LEVELS)
            {
                if (!s->ev_lvl[lvl])
                    GN_HIP(hipEventCreate(&s->ev_lvl[lvl]));
                GN_HIP(hipEventRecord(s->ev_lvl[lvl], st));
            }
            GnHibfLevelParams p{};
            p.lvl_bytes   = s->d_hctr + 3 * NL + lvl;     // (the level's kernels add to their own slots: no copy between the levels)
            p.lvl_lines   = s->d_hctr + 4 * NL + 2 + lvl;
            p.ibfs        = f->d_hibf;
            p.hashes      = s->v_hashes;
            p.slot_off    = s->v_slot_off;
            p.n_hashes    = s->v_nh;
            p.rel_cutoff  = s->rel_cutoff;
            p.wide        = s->long_reads ? 1u : 0u;
            p.reread      = gn_sw().hibf_reread ? 1u : 0u;
            p.fake_hashes = gn_sw().hibf_fake_hashes && lvl > 0 ? 1u : 0u; // (level 0 keeps its real hashes: the lower levels get their real items)
            p.work_in     = s->d_work[lvl & 1];
            p.count_in    = s->d_hctr + lvl;
            p.work_out    = s->d_work[(lvl + 1) & 1];
            p.count_out   = s->d_hctr + lvl + 1;
            p.work_cap    = s->work_cap;
            p.ctr         = s->d_ctr;
            p.keys        = s->d_keys[0];
            p.vals        = s->d_vals[0];
            p.match_cap   = s->match_cap;
            p.ub_bits     = ub_bits;
            p.lds_bins    = f->max_bins;
            p.n_reads     = cnt;
            p.read_base   = lo;
            p.status      = s->v_status;
            p.pack_gp     = lvl < f->level_gp.size() ? f->level_gp[lvl] : 0u;
            p.grab        = gn_sw().on_demand ? nullptr : s->d_hctr + 29 * NL + 2 + (size_t)lvl * 8;
            bool level0   = lvl == 0; // the first register kernel of level 0 takes the reads themselves as its items
            uint2* defer_next = s->d_hdefer; // the list the next kernel of this level writes what it leaves
            auto   launch_pack = [&]() {
                switch (h)
                {
                    case 1: gn_hibf_launch_pack<1>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 2: gn_hibf_launch_pack<2>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 3: gn_hibf_launch_pack<3>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 4: gn_hibf_launch_pack<4>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    default: gn_hibf_launch_pack<5>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                }
            };
            const bool one_pass = gn_sw().hibf_one_pack; // tests: the level's most common width only
            const std::vector<uint32_t>& gps = lvl < f->level_gps.size() ? f->level_gps[lvl] : std::vector<uint32_t>();
            if (!no_pack && !level0 && gps.size() > 1 && !one_pass)
            {
                // IBFs of several widths on this level: the queue is sorted by width first, then one packed launch per width over its
                // part of the sorted list; what those do not take (and the widths beyond 64 words) is the per-item kernels' list
                GnHibfBucketParams bp{};
                bp.ibfs     = f->d_hibf;
                bp.n_hashes = s->v_nh;
                bp.work_in  = p.work_in;
                bp.count_in = p.count_in;
                bp.work_cap = s->work_cap;
                for (uint32_t g = 0; g < 8; ++g)
                    bp.cls_of_gp[g] = 7;
                for (size_t c = 0; c < gps.size() && c < 7; ++c)
                    bp.cls_of_gp[gps[c]] = (uint8_t)c;
                bp.cls_count  = s->d_hctr + 5 * NL + 2 + (size_t)lvl * 8;
                bp.cls_base   = s->d_hctr + 13 * NL + 2 + (size_t)lvl * 8;
                bp.cls_cursor = s->d_hctr + 21 * NL + 2 + (size_t)lvl * 8;
                bp.sorted_out = s->d_hdefer;
                bp.rest_out   = s->d_hdefer2;
                bp.rest_count = s->d_hctr + NL + lvl;
                const dim3 grid((uint32_t)f->n_cu * 8u);
                if (!gn_sw().hibf_nsort && s->d_hsub)
                {
                    GnHibfSubParams sp{ s->d_hsub + (size_t)lvl * 384, s->d_hsub + (size_t)lvl * 384 + 128, s->d_hsub + (size_t)lvl * 384 + 256 };
                    hipLaunchKernelGGL(gn_hibf_nsort_kernel<false>, grid, dim3(256), 0, st, bp, sp);
                    hipLaunchKernelGGL(gn_hibf_nsort_bases_kernel, dim3(1), dim3(1), 0, st, sp.sub_count, sp.sub_base, bp.cls_count, bp.cls_base);
                    hipLaunchKernelGGL(gn_hibf_nsort_kernel<true>, grid, dim3(256), 0, st, bp, sp);
                }
                else
                {
                    hipLaunchKernelGGL(gn_hibf_bucket_kernel<false>, grid, dim3(256), 0, st, bp);
                    hipLaunchKernelGGL(gn_hibf_bucket_bases_kernel, dim3(1), dim3(1), 0, st, bp.cls_count, bp.cls_base);
                    hipLaunchKernelGGL(gn_hibf_bucket_kernel<true>, grid, dim3(256), 0, st, bp);
                }
                GN_HIP(hipGetLastError());
                p.work_in     = s->d_hdefer;
                p.defer_out   = s->d_hdefer2;
                p.defer_count = s->d_hctr + NL + lvl;
                if (!gn_sw().hibf_persistent)
                {
                    // ONE persistent launch walks the classes of the sorted list one after the other (a launch per width left the chip
                    // draining five times a level, and the narrow classes took a launch each for a few thousand items)
                    p.n_cls     = (uint32_t)std::min<size_t>(gps.size(), 7);
                    for (uint32_t c = 0; c < p.n_cls; ++c)
                        p.cls_gp[c] = (uint8_t)gps[c];
                    p.cls_count = bp.cls_count;
                    p.cls_base  = bp.cls_base;
                    launch_pack();
                    GN_HIP(hipGetLastError());
                    p.n_cls = 0;
                }
                else
                    for (size_t c = 0; c < gps.size() && c < 7; ++c)
                    {
                        p.pack_gp   = gps[c];
                        p.count_in  = bp.cls_count + c;
                        p.work_base = bp.cls_base + c;
                        p.grab      = gn_sw().on_demand ? nullptr : s->d_hctr + 29 * NL + 2 + (size_t)lvl * 8 + c; // (a launch per width: a cursor per launch)
                        launch_pack();
                        GN_HIP(hipGetLastError());
                    }
                p.work_base = nullptr;
                p.grab      = gn_sw().on_demand ? nullptr : s->d_hctr + 29 * NL + 2 + (size_t)lvl * 8 + 7;
                p.work_in   = s->d_hdefer2;
                p.count_in  = s->d_hctr + NL + lvl;
                defer_next  = s->d_hdefer; // (the sorted list is done with)
            }
            else if (!no_pack)
            {
                p.defer_out   = defer_next;
                p.defer_count = s->d_hctr + NL + lvl;
                launch_pack();
                GN_HIP(hipGetLastError());
                p.work_in  = p.defer_out;
                p.count_in = p.defer_count;
                defer_next = s->d_hdefer2;
                level0     = false;
            }
            if (!no_reg)
            {
                p.defer_out   = defer_next;
                p.defer_count = s->d_hctr + 2 * NL + lvl;
                switch (h)
                {
                    case 1: gn_hibf_launch_reg<1>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 2: gn_hibf_launch_reg<2>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 3: gn_hibf_launch_reg<3>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    case 4: gn_hibf_launch_reg<4>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                    default: gn_hibf_launch_reg<5>(p, level0, (uint32_t)f->n_cu, reg_bpc, st); break;
                }
                GN_HIP(hipGetLastError());
                p.work_in  = p.defer_out;
                p.count_in = s->d_hctr + 2 * NL + lvl;
            }
            // LDS-counter kernel: what the register kernels left (or, with the switch above, the whole level)
            const uint32_t wpb = (size_t)f->max_bins * 4 * 4 <= 64 * 1024 ? 4 : 1; // waves per block by LDS need
            const size_t   lds = (size_t)f->max_bins * 4 * wpb;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gn_hibf_level_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
            hipLaunchKernelGGL(gn_hibf_level_kernel, dim3((uint32_t)f->n_cu * (wpb == 4 ? 8u : 16u)), dim3(wpb * 64), lds, st, p);
            GN_HIP(hipGetLastError());
        }
        if (cnt && stamp)
        {
            const uint32_t last = depth < GN_HIBF_TIMED_LEVELS ? depth : GN_HIBF_TIMED_LEVELS;
            if (!s->ev_lvl[last])
                GN_HIP(hipEventCreate(&s->ev_lvl[last]));
            GN_HIP(hipEventRecord(s->ev_lvl[last], st));
        }
        GN_HIP(hipMemcpyAsync(s->h_hctr, s->d_hctr, (5 * NL + 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(s->h_ctr, s->d_ctr, GN_NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        GN_HIP(hipStreamSynchronize(st));
        // per-level byte counts -> cumulative over the levels (what gn_stream_hibf_levels / _level_lines subtract from each other)
        for (uint32_t l = 1; l < NL - 1; ++l)
        {
            s->h_hctr[3 * NL + l] += s->h_hctr[3 * NL + l - 1];
            s->h_hctr[4 * NL + 2 + l] += s->h_hctr[4 * NL + 2 + l - 1];
        }
        return GN_OK;
    };

    s->hibf_levels_run = 0;
    uint64_t need_cap = 0;       // capacity the caller has to provide if this run does not fit (-> d_ctr[0], see gn_finish)
    uint64_t out_upper = 0;      // matches appended so far, holes of the sorted ranges included (an upper bound of *d_out_base)
    uint64_t done_bytes = 0, done_exact = 0, done_lines = 0; // ctr[2] / ctr[6] / ctr[1] after the ranges that are through (a range that is run again starts from them)
    uint32_t n_ranges = 0;
    uint32_t step = s->hibf_range_reads && s->hibf_range_reads < n ? s->hibf_range_reads : n; // (what fitted the last batch)
    uint32_t lo = 0;
    while (lo < n || (n == 0 && n_ranges == 0))
    {
        const uint32_t cnt = n ? std::min<uint32_t>(step, n - lo) : 0;
        rc = run_levels(lo, cnt, lo == 0 && cnt == n);
        if (rc)
            return rc;
        uint64_t worst = 0;
        for (uint32_t i = 0; i < 3 * NL; ++i) // (the fourth row holds byte counts, not queue lengths)
            worst = std::max<uint64_t>(worst, s->h_hctr[i]);
        const uint64_t nm = s->h_ctr[0];
        auto restore = [&]() -> int { // the range is run again: what it added to the batch totals goes
            GN_HIP(hipMemcpyAsync(s->d_ctr + 2, &done_bytes, 8, hipMemcpyHostToDevice, st));
            GN_HIP(hipMemcpyAsync(s->d_ctr + 1, &done_lines, 8, hipMemcpyHostToDevice, st));
            GN_HIP(hipMemcpyAsync(s->d_ctr + 6, &done_exact, 8, hipMemcpyHostToDevice, st));
            GN_HIP(hipStreamSynchronize(st));
            return GN_OK;
        };
        if (worst > s->work_cap)
        {
            // a queue overflowed (entries past the capacity were dropped): grow the queues and run the range again
            for (uint2** q : { &s->d_work[0], &s->d_work[1], &s->d_hdefer, &s->d_hdefer2 })
            {
                hipFree(*q);
                *q = nullptr;
            }
            s->work_cap = (uint32_t)std::min<uint64_t>(worst + worst / 4 + 1024, 0xFFFFFFF0ull);
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_work[0]), (size_t)s->work_cap * sizeof(uint2)));
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_work[1]), (size_t)s->work_cap * sizeof(uint2)));
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_hdefer), (size_t)s->work_cap * sizeof(uint2)));
            GN_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_hdefer2), (size_t)s->work_cap * sizeof(uint2)));
            if ((rc = restore()))
                return rc;
            continue;
        }
        if (nm > pair_limit && cnt > 1)
        {
            // more pairs than one sort (or the device) takes: this range in two halves
            step = (cnt + 1) / 2;
            if ((rc = restore()))
                return rc;
            continue;
        }
        if (nm > 0x7FFFFFF0ull)
            return gn_fail(GN_ERANGE, "one read has %llu HIBF matches: more than one sort takes", (unsigned long long)nm);
        if (nm > s->match_cap)
        {
            need_cap = std::max(need_cap, nm); // fits after the caller has grown the buffers: nothing more to do in this run
            break;
        }
        // ---- pre-drop, sort, finish of this range ----
        uint64_t ns  = nm; // pairs to sort
        int      src = 0;
        if (may_predrop && nm > 8ull * cnt)
        {
            GN_HIP(hipMemsetAsync(s->d_pf_pre + 1, 0, sizeof(unsigned long long), st));
            const unsigned blocks = (unsigned)std::min<uint64_t>((nm + 255) / 256, (uint64_t)f->n_cu * 8);
            hipLaunchKernelGGL(gn_hibf_premax_kernel, dim3(blocks), dim3(256), 0, st, s->d_keys[0], s->d_vals[0], nm, ub_bits, s->v_nh,
                               s->d_pf_rmax);
            hipLaunchKernelGGL(gn_hibf_predrop_kernel, dim3(blocks), dim3(256), 0, st, s->d_keys[0], s->d_vals[0], nm, ub_bits, s->v_nh,
                               s->d_pf_rmax, s->rel_cutoff, s->pf_joint ? 2u : 1u, s->pf_rel_filter, s->d_keys[1], s->d_vals[1], s->match_cap,
                               s->d_pf_pre + 1, s->d_pf_segmin, s->d_pf_pre);
            GN_HIP(hipGetLastError());
            unsigned long long out_n = 0;
            GN_HIP(hipMemcpyAsync(&out_n, s->d_pf_pre + 1, sizeof(out_n), hipMemcpyDeviceToHost, st));
            GN_HIP(hipStreamSynchronize(st));
            if (out_n <= s->match_cap) // (else: chunk holes pushed it past the buffer -- the raw pairs are sorted as they are)
            {
                ns            = out_n;
                src           = 1;
                s->pf_predrop = true;
            }
            else
            {
                // chunk holes pushed the survivors past the buffer: the batch is run again with more room (everything the
                // pre-drop has noted so far is reset at the start of that run)
                need_cap = std::max<uint64_t>(need_cap, out_n);
                break;
            }
        }
        if (out_upper + ns > s->match_cap)
        {
            need_cap = std::max(need_cap, out_upper + ns + ((uint64_t)(n - lo - cnt) / std::max<uint32_t>(cnt, 1u)) * ns);
            break;
        }
        if (ns)
        {
            size_t tmp = s->sort_tmp_bytes;
            GN_HIP(hipcub::DeviceRadixSort::SortPairs(s->d_sort_tmp, tmp, s->d_keys[src], s->d_keys[1 - src], s->d_vals[src], s->d_vals[1 - src],
                                                      (int)ns, 0, (int)(ub_bits + rd_bits), st));
            if (src == 1) // (the sorted pairs are expected in buffer 1)
            {
                std::swap(s->d_keys[0], s->d_keys[1]);
                std::swap(s->d_vals[0], s->d_vals[1]);
            }
            hipLaunchKernelGGL(gn_hibf_finish_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, s->d_keys[1], s->d_vals[1], ns,
                               ub_bits, s->v_nh, s->d_sorted, s->d_seg_count, d_out_base, s->match_cap);
            hipLaunchKernelGGL(gn_hibf_advance_kernel, dim3(1), dim3(1), 0, st, s->d_keys[1], ns, d_out_base);
            GN_HIP(hipGetLastError());
            out_upper += ns;
        }
        done_bytes = s->h_ctr[2];
        done_lines = s->h_ctr[1];
        done_exact = s->h_ctr[6];
        ++n_ranges;
        if (n_ranges == 1 && cnt == n)
            s->hibf_levels_run = depth;
        lo += cnt;
        if (n == 0)
            break;
    }
    s->hibf_ranges = n_ranges;
    if (n_ranges > 1 || step < n)
        s->hibf_range_reads = step; // the next batch starts with ranges of the size that went through
    if (need_cap)
    {
        // gn_finish() reads the cursor, sees that it exceeds the capacity, grows the buffers and runs the batch again
        const unsigned long long v = need_cap;
        GN_HIP(hipMemcpyAsync(s->d_ctr, &v, sizeof(v), hipMemcpyHostToDevice, st));
        GN_HIP(hipStreamSynchronize(st));
        return GN_OK;
    }
    {
        // the cursor of a batch that went through: within the capacity (the gather / pre-pass kernels look at it)
        const unsigned long long v = out_upper;
        GN_HIP(hipMemcpyAsync(s->d_ctr, &v, sizeof(v), hipMemcpyHostToDevice, st));
        GN_HIP(hipStreamSynchronize(st));
    }
    size_t tmp = s->scan_tmp_bytes;
    GN_HIP(gn_scan_counts(s->d_scan_tmp, tmp, s->d_seg_count, s->d_seg_off, (int)(n + 1), st));
    return GN_OK;
}

// dense tap: uint16[n_user_bins] per read == counting_agent_type::bulk_count(values, T) (raw, uncapped sums)
int gn_hibf_dense(gn_stream* s, uint32_t rb, uint32_t re, uint16_t* counts)
{
    gn_filter*            f  = s->f;
    const uint64_t        nm = s->n_matches;
    const uint32_t        ub_bits = std::max(1u, gn_bits_for(f->n_user_bins ? f->n_user_bins - 1 : 0));
    if (s->hibf_ranges > 1)
        return gn_fail(GN_ERANGE, "dense counts: the batch was run in %u read ranges (the tap reads one sorted pair buffer)", s->hibf_ranges);
    std::vector<uint64_t> keys(nm ? nm : 1);
    std::vector<uint32_t> vals(nm ? nm : 1);
    if (nm)
    {
        GN_HIP(hipMemcpy(keys.data(), s->d_keys[1], nm * 8, hipMemcpyDeviceToHost));
        GN_HIP(hipMemcpy(vals.data(), s->d_vals[1], nm * 4, hipMemcpyDeviceToHost));
    }
    std::fill(counts, counts + (size_t)(re - rb) * f->n_user_bins, (uint16_t)0);
    for (uint64_t i = 0; i < nm; ++i)
    {
        const uint32_t r = (uint32_t)(keys[i] >> ub_bits);
        if (r >= rb && r < re)
            counts[(size_t)(r - rb) * f->n_user_bins + (uint32_t)(keys[i] & ((1ULL << ub_bits) - 1ULL))] = (uint16_t)vals[i];
    }
    return GN_OK;
}

// Per tree level of the last HIBF batch: time of the level's kernels (hipEvents on the stream), algorithmic row bytes
// n*h*W*8 summed over the items of the level, the bytes of the IBFs at that depth (a level whose tables fit the 256 MiB
// Infinity Cache is not HBM bound) and their usual row width.  Levels beyond GN_HIBF_TIMED_LEVELS share the last stamp.
extern "C" int gn_stream_hibf_level_lines(gn_stream* s, uint64_t* line_bytes, uint32_t cap)
{
    if (!s || !line_bytes)
        return gn_fail(GN_EINVAL, "null argument");
    if (!s->f->is_hibf)
        return gn_fail(GN_EINVAL, "gn_stream_hibf_level_lines: the stream's filter is not an HIBF");
    int rc = gn_finish_batch(s);
    if (rc)
        return rc;
    const uint32_t NL = GN_HIBF_MAXDEPTH + 1;
    const uint32_t timed = s->hibf_levels_run < GN_HIBF_TIMED_LEVELS ? s->hibf_levels_run : GN_HIBF_TIMED_LEVELS;
    uint64_t       prev = 0;
    for (uint32_t l = 0; l < timed && l < cap; ++l)
    {
        const uint64_t cum = s->h_hctr[4 * NL + 2 + (l + 1 == timed ? s->hibf_levels_run - 1 : l)];
        line_bytes[l]      = cum - prev;
        prev               = cum;
    }
    return GN_OK;
}

extern "C" int gn_stream_hibf_levels(gn_stream* s, uint32_t* n_levels, float* ms, uint64_t* algo_bytes, uint64_t* table_bytes,
                                     uint32_t* row_bytes, uint32_t cap)
{
    if (!s || !n_levels)
        return gn_fail(GN_EINVAL, "null argument");
    if (!s->f->is_hibf)
        return gn_fail(GN_EINVAL, "gn_stream_hibf_levels: the stream's filter is not an HIBF");
    int rc = gn_finish_batch(s);
    if (rc)
        return rc;
    const uint32_t NL = GN_HIBF_MAXDEPTH + 1;
    const uint32_t timed = s->hibf_levels_run < GN_HIBF_TIMED_LEVELS ? s->hibf_levels_run : GN_HIBF_TIMED_LEVELS;
    *n_levels = timed;
    uint64_t prev = 0;
    for (uint32_t l = 0; l < timed && l < cap; ++l)
    {
        const bool last = l + 1 == timed;
        if (ms)
        {
            ms[l] = 0.f;
            hipEventElapsedTime(&ms[l], s->ev_lvl[l], s->ev_lvl[l + 1]);
        }
        const uint64_t cum = s->h_hctr[3 * NL + (last ? s->hibf_levels_run - 1 : l)];
        if (algo_bytes)
            algo_bytes[l] = cum - prev;
        prev = cum;
        uint64_t tb = 0;
        for (uint32_t x = l; x < (last ? (uint32_t)s->f->level_bytes.size() : l + 1); ++x)
            tb += s->f->level_bytes[x];
        if (table_bytes)
            table_bytes[l] = tb;
        if (row_bytes)
            row_bytes[l] = s->f->level_row_bytes[l];
    }
    return GN_OK;
}
