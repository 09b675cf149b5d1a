// gn_internal.h -- device-side parameter blocks and launchers shared by gn_kernels.hip and gn_capi.hip.
// gfx950 (MI355X) only.  Not part of the public ABI (that is include/ganon_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/ganon_hip.h"
#include "../../include/ganon_ibf_hash.h"

#define GN_MAX_CHUNKS 64 // pipeline chunks per batch (minimiser on the side stream || count on the main stream)
// candidate-driven select of the generic count kernel: targets with more bins than this are scanned from a list
#define GN_CAND_NBIG 4u
#define GN_HIBF_MAXDEPTH 64
#define GN_HIBF_TIMED_LEVELS 8 // tree levels with a time stamp of their own (gn_stream_hibf_levels)
#define GN_PF_MAX_JOINT 16   // filters of one hierarchy level in a joint filter_matches pre-pass, per device
#define GN_PF_MAX_DEVICES 16 // devices a joint pass spans (column parts of a bin-range partitioned filter)
#define GN_LONG_BLOCKS 128u // workgroups (and uint32 count slabs) of the long-read kernel
#define GN_NCTR 73 // device counters: [0] match cursor [2] algo bytes [3] hibf work [4] count-deferred [5] minimiser-deferred
                   // [6] exact match total [8..71] total-hashes shards [72] unit cursor of the fast count kernel

// ---- switches ---------------------------------------------------------------------------------
// The ONE place the library reads its environment: $GANON_HIP_ABLATE, a comma list of the names below, parsed once when
// the library is loaded into this struct (gn_capi.hip).  Everything is off by default = the product path.  Tests and
// bench.py cross-check a fast path against the path it replaces by switching it off in-process with gn_ablate()
// (include/ganon_hip.h); nothing else changes the struct, and no launch path calls getenv().
struct GnSwitches
{
    // count + select (flat IBF)
    bool early_exit = false;      // fast kernel: fetch every row (no exact early exit)
    bool cand_select = false;     // generic kernel: scan every target instead of the candidate-driven select
    bool csr_identity = false;    // split-bin kernel family: treat an ordered CSR map as a general one
    bool uniform_select = false;  // ... no packed select for equal bins-per-target
    bool run_select = false;      // ... no run-sum select for ordered unequal targets
    bool max_first = false;       // ... per-read maximum computed in the second pass
    bool const_nb = false;        // ... bins-per-target read from the table, not from the template constant
    bool split_kernel = false;    // split-bin maps go to the generic kernel
    bool predrop = false;         // with a filter_matches pre-pass: write every match, judge afterwards
    bool deferred_grids = false;  // grids of the deferred-list kernels sized for the whole batch, not from the last batch's list
    bool seg_result = false;      // a flat IBF's matches are copied behind each other after every batch (as up to round 5), not left in their segments
    bool on_demand = false;       // fast kernel and HIBF packed kernel: every wave takes units i, i + waves, ... (no cursor: as up to round 5)
    // HIBF
    bool hibf_reg = false;        // no per-item register kernel (LDS-counter level kernel instead)
    bool hibf_pack = false;       // no packed-items kernel either
    bool hibf_one_pack = false;   // only the level's most common width takes the packed kernel (no sorting by width)
    bool hibf_reread = false;     // A/B: lanes whose item is through read their last row again (as up to round 4)
    bool hibf_nsort = false;      // a level's queue is sorted by row width only, not by (width, number of minimisers)
    bool hibf_stage = false;      // the packed kernel fetches every hash from global memory (no staging in LDS)
    bool hibf_persistent = false; // one launch per width class instead of one persistent launch per level
    bool hibf_dense_rows = false; // (read when an HIBF is created) rows of ceil(bins / 64) words on the device, as up to round 5: no padding to whole lines
    bool hibf_fake_hashes = false; // TIMING EXPERIMENT ONLY (wrong results): the packed kernel loads one hash per item and derives the others
    uint32_t hibf_bpc = 0;        // >0: workgroups per CU of the HIBF register kernels (0: what the occupancy query says)
    // device inflate
    bool     inflate_ahead = false; // no decode launched ahead of the step that needs it
    uint64_t hibf_pair_limit = 0; // >0: (read, user bin) pairs per round of a batch (tests make a batch take several rounds)
    // multi-device
    bool gather_copy = false;     // same-device parts take the peer-copy path too
    bool joint_apart = false;     // joint pre-pass: every stream is treated as a device of its own
    // batch pipeline
    uint32_t chunk = 0;           // >0: reads per pipeline chunk (minimiser || count on two HIP streams)
    int      sync_mode = 0;       // 0 runtime default, 1 spin, 2 yield, 3 block: how host threads wait for the device
    bool     pinned_malloc = false; // gn_pinned_alloc: hipHostMalloc as up to round 4 (no huge-page mapping + hipHostRegister)
    bool     debug = false;       // chatter on stderr
    // host-ceiling measurement
    uint32_t emit_probe = 0;      // MEASUREMENT ONLY (wrong results): fast kernel at low cutoffs -- 64: the listed matches are not stored, 128: nor listed
    bool     fake_count = false;  // MEASUREMENT ONLY (wrong results): upload, record index and minimisers run, the count + select kernels do not; every
                                  // second read of a flat IBF gets one made-up match (read % targets, its number of minimisers) -- the device
                                  // step costs next to nothing and what is left is what the host's reader, workers and post stage sustain
};
const GnSwitches& gn_sw();
struct gn_stream;
int             gn_result_compact(gn_stream* s);  // gn_capi.hip: the contiguous copy of a segmented result, queued on the stream (no-op otherwise)
const gn_match* gn_result_matches(gn_stream* s);  // the batch's final matches, contiguous by read (after gn_result_compact / the pre-pass)

// ---- minimiser kernel -----------------------------------------------------------------------
struct GnMinimiserParams
{
    const uint8_t*      bases;     // ASCII
    const uint64_t*     off1;      // n_reads+1
    const uint64_t*     off2;      // n_reads+1 or nullptr
    const uint64_t*     slot_off;  // n_reads+1: first hash slot of each read (upper bound = #windows)
    uint32_t            n_reads;   // END of the read range (exclusive)
    uint32_t            read_begin; // first read of the range this launch covers
    uint32_t            k, w;
    uint64_t*           hashes;    // slot_off[n_reads] slots
    uint32_t*           n_hashes;  // per read
    uint8_t*            status;    // per read GN_READ_*
    unsigned long long* total_hashes; // 64 shards (indexed by blockIdx & 63): sum of n over GN_READ_OK reads
    // lane-per-read kernel: reads longer than lpr_max_len go to defer_list; wave-per-read kernel: work_list input
    uint32_t                  lpr_max_len;
    uint32_t*                 defer_list;
    unsigned long long*       defer_count;
    const uint32_t*           work_list;  // nullptr = every read
    const unsigned long long* work_count;
    uint32_t                  work_hint;  // launch only: reads the previous batch's list held (~0u: unknown -> full grid)
};

hipError_t gn_launch_minimiser_lpr(const GnMinimiserParams& p, hipStream_t st);

// ---- flat IBF count + select kernel ---------------------------------------------------------
struct GnCountParams
{
    // filter
    const uint64_t* rows;
    uint64_t        S;        // rows
    uint32_t        W;        // words per row
    uint32_t        B;        // bins
    uint32_t        shift;    // hash_shift
    const uint32_t* tgt_off;  // CSR over targets (n_targets+1); nullptr when identity
    const uint32_t* tgt_bins;
    const uint4*    tgt_rec;  // per target: {first CSR entry, #bins, LDS slot of bin 0, LDS slot of bin 1}
    const uint32_t* tgt_lds;  // per CSR entry: (LDS dword of the bin's u16 pair) * 2 + half (see gn_count_lds_index)
    const uint32_t* tgt_ids;  // id reported for CSR target t (nullptr = t)
    uint32_t        n_targets;
    // batch
    const uint64_t* hashes;
    const uint64_t* slot_off;
    const uint32_t* n_hashes;
    const uint8_t*  status;
    uint32_t        n_reads;    // END of the read range (exclusive)
    uint32_t        read_begin; // first read of the range this launch covers
    double          rel_cutoff;
    // geometry (host-chosen, see gn_count_geometry)
    uint32_t wpr;      // waves cooperating on one read = column slices of a row (1..16)
    uint32_t gp_log2;  // lanes per hash group = 1 << gp_log2
    uint32_t slice_dwords; // LDS dwords of one wave's count slice = 32*LW*(Gp+1)
    // output
    gn_match*           matches;
    uint64_t            match_cap;
    unsigned long long* cursor;
    uint64_t*           seg_begin;  // n_reads*wpr
    uint32_t*           seg_count;  // n_reads*wpr
    // debug tap: dense counts of reads [dense_begin, dense_end)
    uint16_t* dense;
    uint32_t  dense_begin, dense_end;
    // work list: generic kernel input (nullptr = every read), fast kernel output (reads it defers)
    const uint32_t*           work_list;
    const unsigned long long* work_count;
    uint32_t*                 work_list_out;
    unsigned long long*       work_count_out;
    uint32_t                  max_blocks; // generic kernel: persistent grid size
    uint32_t                  max_blocks_fast; // fast kernel: persistent grid size
    uint32_t                  nt_loads;        // fast kernel: non-temporal row loads
    // generic kernel, candidate-driven select (split bins): bin -> CSR target, and (bins of the bin's target, capped
    // at 255) as u16 pairs in the layout of the LDS count area; nullptr = scan every target
    const uint32_t*           bin_tgt;
    const uint32_t*           bin_nb2;
    uint32_t                  nbtab_off;       // dword offset of the LDS copy of bin_nb2 (0 = none)
    uint32_t                  candcnt_off;     // dword offset of the per-wave candidate counters in LDS
    const uint32_t*           big_list;        // targets with more than GN_CAND_NBIG bins
    uint32_t                  n_big;
    const uint32_t*           sl_nbr;          // split kernel: bins-per-target bytes in the layout of the byte counters
    uint32_t                  early_exit;      // fast kernel: stop fetching rows of reads that cannot reach the cutoff
    unsigned long long*       skip_ctr;        // row bytes not fetched thanks to early exits
    unsigned long long*       grab;            // fast kernel: the cursor its waves take their later units from (zero at launch; nullptr: units i, i + waves, ...)
    // fast kernel, with a filter_matches pre-pass set on the stream: matches that the --rel-filter rule is bound to drop are
    // not written at all (see the epilogue).  0 off, 1: the read's minimum is at least its cutoff count T (this filter sees all
    // of the read's matches), 2: nothing known about the minimum (other filters of the level may report smaller counts)
    uint32_t            uniform_nb;   // csr_identity and every target owns exactly this many bins (2 or 4): packed select of the split kernel; else 0
    uint32_t            max_first;    // split kernel, pre-pass mode, uniform_nb or run_select: the read's true maximum is found before the first select
    uint32_t            const_nb;     // 0, or the bins-per-target the split kernel's RS variant takes for every bin: uniform_nb, or 4 with run_select
    uint32_t            run_select;   // csr_identity, no target of more than four bins, widths differ: the split kernel judges targets from a running sum over a lane's own bins
    uint32_t            csr_identity; // tgt_bins[i] == i: target t owns the bins [tgt_off[t], tgt_off[t+1]) (what ganon-build writes)
    uint32_t            pre_mode;
    double              pre_rel;
    uint32_t*           seg_min; // n_reads*wpr: smallest count among a unit's unwritten matches (0xFFFFFFFF: none)
    unsigned long long* pre_ctr; // matches not written
};

// size_t threshold_filter = max - size_t(std::ceil((max - min) * rel_filter))   (GanonClassify.cpp:755-757)
// non-decreasing in max and in min for 0 <= rel_filter < 1 (one more in max raises the ceil by at most one)
__device__ __forceinline__ uint32_t gn_pf_threshold(uint32_t mx, uint32_t mn, double rel_filter)
{
    return mx - (uint32_t)(unsigned long long)ceil(__dmul_rn((double)(mx - mn), rel_filter));
}

struct GnCountGeometry
{
    uint32_t lw;       // words per lane per row (1 or 2)
    uint32_t wpr;      // waves per read
    uint32_t gp_log2;  // lanes per group
    uint32_t block;    // threads per block
    uint32_t rpb;      // reads per block
    uint32_t slice_dwords;
    size_t   lds_bytes;
    size_t   nbtab_off;   // dword offset of the bins-per-target table inside the block's LDS (0 = read it from global)
    size_t   candcnt_off; // dword offset of the per-wave candidate counters
};

// returns false (and a message) when the IBF shape is outside what the kernel supports
bool gn_count_geometry(uint64_t W, uint32_t hash_funs, GnCountGeometry* g, const char** why);
// LDS position of bin b inside a read's count area for geometry g: dword index * 2 + u16 half
uint32_t gn_count_lds_index(const GnCountGeometry& g, uint32_t b);

hipError_t gn_launch_minimiser(const GnMinimiserParams& p, int n_cu, hipStream_t st);
hipError_t gn_launch_count(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st);
hipError_t gn_launch_count_fast(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st);
// split-bin maps, reads with <= 127 minimisers, rows of at least one wave (gn_split.hip)
hipError_t gn_launch_count_long(const GnCountParams& p, uint32_t hash_funs, uint32_t* list, unsigned long long* count, uint32_t* scratch,
                                uint32_t blocks, hipStream_t st);
hipError_t gn_launch_count_split(const GnCountParams& p, const GnCountGeometry& g, uint32_t hash_funs, hipStream_t st);
size_t     gn_split_lds_bytes(const GnCountGeometry& g, uint32_t hash_funs);

// ---- HIBF -----------------------------------------------------------------------------------
struct GnHibfIbfDev
{
    const uint64_t* rows;
    uint64_t        S;
    uint32_t        W, B, shift, h; // W: words from one row to the next = the width the kernels count (GnIbfHost::Ws: the padding words are zero)
    uint32_t        Wl;             // ceil(B / 64): the words of a row that hold bins (the algorithmic bytes of a row request)
    const uint4*    runs;   // per run: first bin, n bins, user bin (0xFFFFFFFF = merged), child ibf
    uint32_t        n_runs;
    // register-counter level kernel (W <= 64): per technical bin (64*W entries) what a single-bin run leads to --
    // 0x80000000 | child ibf (merged bin), user bin (leaf), 0xFFFFFFFF (bin of a multi-bin run, or padding) -- and the
    // runs of more than one bin (split user bins), which are summed from an LDS image of the counters
    const uint32_t* bin_tab;
    const uint4*    mruns;
    uint32_t        n_mruns;
};

// ---- misc kernels ---------------------------------------------------------------------------
int  gn_run_postfilter(gn_stream* s);     // gn_postfilter.hip
void gn_postfilter_release(gn_stream* s);
void gn_build_release(gn_stream* s);       // gn_build.hip
void gn_fastq_release(gn_stream* s);       // gn_fastq.hip
void gn_peer_enable(int dst, int src);     // gn_gather.hip: direct device-to-device copies between the two (best effort)
hipError_t gn_launch_emplace(uint64_t* rows, uint64_t S, uint32_t W, uint32_t shift, uint32_t h, const uint64_t* hashes,
                             const uint32_t* bins, uint64_t n, hipStream_t st);

// ---- host-side objects behind the opaque C handles --------------------------------------------
int gn_fail(int code, const char* fmt, ...);

#define GN_HIP(expr)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return gn_fail(e__ == hipErrorOutOfMemory ? GN_ENOMEM : GN_ENODEV, "%s failed: %s (%s:%d)", #expr,         \
                           hipGetErrorString(e__), __FILE__, __LINE__);                                                \
    } while (0)

struct GnIbfHost
{
    uint64_t* d_rows = nullptr;
    uint64_t  S = 0, W = 0, B = 0;
    uint64_t  Ws = 0; // words from one row to the next on the device: W, or -- the IBFs of an HIBF -- W padded so that no row straddles a 128-byte
                      // line (gn_pad_row_words); the words beyond W are zero.  Everything the C ABI takes or returns is W words a row.
    uint32_t  h = 0, shift = 0;
};

// Row stride of an HIBF's IBF.  raptor lays an IBF out with ceil(bins / 64) words a row: 3, 5, 9 ... 15 words for most lower IBFs, and a row
// of 72 .. 120 bytes lies across two 128-byte lines more often than not -- a row request then moves two lines.  Measured on the skewed
// tree (rocprofv3 FETCH_SIZE per launch): level 1 fetched 47.9 GB for 40.1 GB of rows counted in lines of their own, level 2 10.4 GB for
// 6.9 -- and moved them at 0.98 / 0.90 of the gather roof: the levels were short of lines, not of speed.  Padded to the next power of two
// (up to 16 words = one line) or to whole lines beyond, a row never straddles.  Costs memory (at most 2x for an IBF, ~1.3x for a tree).
static inline uint64_t gn_pad_row_words(uint64_t w)
{
    if (w >= 16)
        return (w + 15) & ~15ull;
    uint64_t p = 1;
    while (p < w)
        p <<= 1;
    return p;
}

struct gn_filter
{
    int      device  = 0;
    int      n_cu    = 256;
    bool     is_hibf = false;
    bool     storage_only = false; // flat, created without a bin map: no streams (gn_filter_upload_ibf)
    uint64_t device_bytes = 0;
    hipStream_t load_st = nullptr; // streaming upload (gn_filter_write_rows), created on first use
    uint64_t*   d_emplace_stage = nullptr; // gn_filter_emplace_split's staging buffer
    uint64_t    emplace_stage_cap = 0;     // ... and its capacity in hashes
    // flat
    GnIbfHost       ibf;
    uint32_t*       d_tgt_off  = nullptr;
    uint32_t*       d_tgt_bins = nullptr;
    uint32_t*       d_tgt_lds  = nullptr;
    uint4*          d_tgt_rec  = nullptr;
    uint32_t*       d_bin_tgt  = nullptr; // bin -> CSR target (0xFFFFFFFF = none)
    uint32_t*       d_bin_nb2  = nullptr; // see GnCountParams::bin_nb2
    uint32_t*       d_big_list = nullptr; // targets with more than GN_CAND_NBIG bins
    uint32_t        n_big      = 0;
    uint32_t*       d_sl_nbr   = nullptr; // split kernel (GnCountParams::sl_nbr); nullptr = not applicable
    uint32_t        split_bpc  = 0;       // split kernel: resident blocks per CU
    uint32_t        n_targets  = 0;
    bool            identity   = false;
    uint32_t        uniform_nb = 0;       // see GnCountParams::uniform_nb
    bool            csr_identity = false; // bins of the targets, target after target, are 0, 1, 2, ... (GnCountParams::csr_identity)
    bool            run_ok       = false; // csr_identity, every target id owns at least one bin, n_targets < 2^28: the run select may number
                                          // targets by counting target ends (an id without bins would shift every later id)
    GnCountGeometry geom{};
    // hibf
    std::vector<GnIbfHost> ibfs;
    std::vector<void*>     hibf_allocs;
    GnHibfIbfDev*          d_hibf   = nullptr;
    uint64_t               n_user_bins = 0;
    uint32_t               max_bins = 0;
    uint32_t               max_depth = 0;
    std::vector<uint32_t>  level_gp;  // per tree level: log2(lanes per row) most of its IBFs have (packed kernel)
    std::vector<std::vector<uint32_t>> level_gps; // ... and every width that occurs there, most common first
    std::vector<uint64_t>  level_bytes; // per tree level: bytes of the IBFs at that depth (is the level's table cache resident?)
    std::vector<uint32_t>  level_row_bytes; // per tree level: row bytes most of its IBFs have
};

struct gn_stream
{
    gn_filter*  f  = nullptr;
    int         device = 0;     // f->device, kept here so that destroying a stream never touches the filter
    hipStream_t st = nullptr;   // main stream: uploads, count/select, grouping, downloads
    hipStream_t st2 = nullptr;  // side stream: slot scan + minimiser kernels, one chunk ahead of the main stream
    hipEvent_t  ev[4]{};        // [0] batch start (side) [1] last minimiser end (side) [2] last count end [3] batch end
    hipEvent_t  ev_sync = nullptr, ev_count0 = nullptr;
    hipEvent_t  ev_chunk[GN_MAX_CHUNKS]{};
    uint32_t    n_chunks = 1;
    uint32_t    max_reads = 0;
    uint64_t    max_bases = 0;
    uint64_t    match_cap = 0;
    // device buffers
    uint8_t*            d_bases    = nullptr;
    uint64_t*           d_off1     = nullptr;
    uint64_t*           d_off2     = nullptr;
    uint64_t*           d_slot_cnt = nullptr; // n+1 window counts (scan input)
    uint64_t*           d_slot_off = nullptr; // n+1
    uint64_t*           d_hashes   = nullptr;
    uint32_t*           d_nh       = nullptr;
    uint8_t*            d_status   = nullptr;
    // what the count / select / fetch side READS: this stream's own buffers above, or -- after gn_stream_classify_shared -- those of
    // the stream whose resident batch it shares (same device, same reads, hashed once)
    const uint64_t*     v_hashes   = nullptr;
    const uint64_t*     v_slot_off = nullptr;
    const uint32_t*     v_nh       = nullptr;
    const uint8_t*      v_status   = nullptr;
    gn_stream*          src        = nullptr; // the stream shared from (nullptr: own batch)
    gn_match*           d_matches  = nullptr; // unordered (reservation order)
    gn_match*           d_sorted   = nullptr; // grouped by read
    // A flat IBF whose reads are one unit each (wpr == 1) leaves every read's matches as ONE contiguous segment of d_matches
    // (seg_begin[r], seg_count[r]): that IS the grouped result, written once.  The pre-pass of filter_matches reads the segments
    // where they lie; the contiguous copy (d_sorted, CSR) is made only when a consumer asks for it (gn_result_compact), not per batch.
    bool                segmented  = false;   // the batch's matches lie in segments of d_matches
    bool                compacted  = true;    // ... and d_sorted holds their contiguous copy
    gn_match*           pf_out     = nullptr; // where the pre-pass put its survivors (d_matches, or d_sorted when it read segments)
    hipEvent_t          ev_cmp[2]{};          // around the last contiguous copy (gn_timings.ms_compact)
    bool                cmp_timed  = false;
    unsigned long long* d_ctr      = nullptr; // [0] cursor [1] total_hashes [2] algo_bytes [3] work count
    uint64_t*           d_seg_begin = nullptr;
    uint32_t*           d_seg_count = nullptr;
    uint64_t*           d_seg_off   = nullptr; // n*wpr+1 exclusive scan of seg_count
    uint32_t*           d_deferred  = nullptr; // reads the fast count kernel left to the generic one
    uint32_t*           d_mdeferred = nullptr; // reads the lane-per-read minimiser kernel left to the wave-per-read one
    void*               d_scan_tmp  = nullptr;
    size_t              scan_tmp_bytes = 0;
    // hibf work queues + sort buffers
    uint2*        d_work[2]{ nullptr, nullptr };
    uint2*        d_hdefer = nullptr;  // (read, ibf) items the packed kernel leaves to the per-item register-counter kernel
    uint2*        d_hdefer2 = nullptr; // ... and those that one leaves to the LDS-counter kernel
    unsigned long long* d_hsub = nullptr; // per level 3 x 128: counts, bases, cursors of the (width class, n-bin) keys of the level's sorted queue
    unsigned long long* d_hctr = nullptr; // NL = GN_HIBF_MAXDEPTH+1 per row: [l] queue length of level l, [NL+l] / [2NL+l] deferred items
    unsigned long long* h_hctr = nullptr; // pinned copy
    uint32_t      work_cap = 0;
    uint64_t*     d_keys[2]{ nullptr, nullptr };
    uint32_t*     d_vals[2]{ nullptr, nullptr };
    void*         d_sort_tmp = nullptr;
    size_t        sort_tmp_bytes = 0;
    uint64_t      hibf_cap = 0;
    hipEvent_t    ev_lvl[GN_HIBF_TIMED_LEVELS + 1]{}; // HIBF: start of every tree level's kernels (and the end of the last)
    uint32_t      hibf_levels_run = 0;
    uint32_t      hibf_ranges = 0;       // read ranges the last HIBF batch was run in (1 = the usual case)
    uint32_t      hibf_range_reads = 0;  // reads per range that went through last time (0: whole batches)
    // reads with more than 65 535 minimisers (gn_stream_set_long_reads): list, counter, uint32 count slabs of the long kernel
    bool                long_reads = false;
    uint32_t*           d_long_list = nullptr;
    unsigned long long* d_long_count = nullptr;
    uint32_t*           d_long_scratch = nullptr;
    // build side (gn_build.hip): pack / sort / unique buffers, allocated by the first gn_stream_distinct_hashes
    uint64_t*           d_build[2]{ nullptr, nullptr };
    void*               d_build_tmp = nullptr;
    size_t              build_tmp_bytes = 0;
    unsigned long long* d_build_ctr = nullptr; // [0] packed hashes [1] distinct hashes
    uint64_t            build_cap = 0;
    uint64_t            build_distinct = ~0ull; // result of the last gn_stream_distinct_hashes on the resident hashes (~0: none)
    // device-side pre-pass of filter_matches (gn_postfilter.hip); off unless gn_stream_set_postfilter enabled it
    bool                pf_on = false;
    double              pf_rel_filter = 0, pf_fpr_query = 1;
    uint32_t*           d_pf_keep = nullptr; // survivors per read
    uint32_t*           d_pf_max  = nullptr; // max count per read before filtering
    uint32_t*           d_pf_min  = nullptr; // joint pass: min count per read (this stream's filter)
    uint32_t*           d_pf_gmax = nullptr; // joint pass: the level's max / min per read (filled on the first stream)
    uint32_t*           d_pf_gmin = nullptr;
    bool                pf_joint  = false;   // the pass is run by gn_streams_postfilter_joint, not with the batch
    bool                pf_joint_done = false; // ... and has run for the batch this stream holds
    uint32_t*           d_pf_peer = nullptr; // joint pass over several devices: [this device's max | min] then one such pair per other device
    uint64_t            pf_peer_cap = 0;     // ... in uint32 entries
    bool                pf_predrop = false;  // this batch's count kernel left surely-dropped matches unwritten (seg_min, d_pf_pre valid)
    bool                pf_merge  = false;   // ... in its merging form (filters of the level share targets)
    uint32_t*           d_pf_gid  = nullptr; // merging form: device target -> level-wide target id
    double*             d_pf_fpr  = nullptr; // per target
    uint32_t*           d_pf_segmin = nullptr; // per (read, column slice): see GnCountParams::seg_min
    uint64_t            pf_segmin_cap = 0;
    unsigned long long* d_pf_pre  = nullptr; // [0] matches the count kernel did not write (they count as dropped by rel_filter) [1] HIBF: cursor
    uint32_t*           d_pf_rmax = nullptr; // HIBF: largest count per read over the raw pairs (gn_hibf_premax_kernel)
    unsigned long long* d_pf_ctr  = nullptr; // [0] dropped rel_filter [1] dropped fpr_query [2] survivors
    unsigned long long* h_pf_ctr  = nullptr; // pinned copy
    void*               d_pf_scan = nullptr;
    size_t              pf_scan_bytes = 0;
    // FASTQ text tokenised on the device (gn_fastq.hip); allocated by the first gn_stream_upload_fastq
    uint8_t*            d_text    = nullptr;
    uint32_t*           d_fq_tile = nullptr; // newline count per 4 KiB tile, then its exclusive scan
    uint32_t*           d_fq_nl   = nullptr; // position of every newline (up to four per read the stream holds)
    uint32_t*           d_fq_rec  = nullptr; // per record: first byte / first letter / number of letters
    uint32_t*           d_fq_seq  = nullptr;
    uint32_t*           d_fq_len  = nullptr;
    unsigned long long* d_fq      = nullptr; // see gn_fq_records_kernel
    uint32_t*           d_fq_hoff = nullptr; // gn_stream_fastq_headers: header length per record, then the exclusive sums
    uint8_t*            d_fq_hdr  = nullptr; // ... the header lines back to back
    void*               d_fq_hscan = nullptr;
    size_t              fq_hscan_bytes = 0;
    unsigned long long* h_fq      = nullptr; // pinned copy
    void*               d_fq_scan = nullptr;
    size_t              fq_scan_bytes = 0;
    uint64_t            fq_text_cap = 0, fq_bytes = 0;
    uint32_t            fq_tiles_cap = 0, fq_nl_cap = 0, fq_reads = 0;
    uint32_t *          d_fq2_tile = nullptr, *d_fq2_nl = nullptr, *d_fq2_rec = nullptr, *d_fq2_seq = nullptr, *d_fq2_len = nullptr; // the mates' text
    bool                fq_pair = false; // the resident text batch is a pair of texts (gn_stream_upload_text_pair)
    double              fq_probe[4]{}; // $GANON_HIP_CALL_TIMING: seconds in the calls of gn_stream_upload_fastq, batches
    bool                fq_pending = false;  // text uploaded, gn_stream_fastq_index not yet called
    // pinned host
    unsigned long long* h_ctr = nullptr;
    // state
    uint32_t n_reads = 0;
    uint64_t n_bases = 0;
    bool     paired  = false;
    bool     have_reads = false, classified = false, hashed = false;
    bool     ctr_copied = false; // the batch's counters are on their way to h_ctr / h_pf_ctr (queued behind its last kernel)
    // reads the previous batch left to the deferred launches (count stage, minimiser stage): their lists are usually empty and a
    // chip-filling persistent grid that finds nothing costs 0.1-0.2 ms per launch; ~0: unknown
    uint64_t prev_count_deferred = ~0ull, prev_min_deferred = ~0ull;
    uint32_t k = 0, w = 0;
    double   rel_cutoff = 0;
    uint64_t n_matches = 0;
    gn_timings tm{};
};

