/*
 * ganon_hip.h -- C ABI of libganon_hip.so: ganon's read-classification hot path on MI355X.
 *
 * This is the drop-in boundary.  It replaces exactly the two SeqAn3 call sites
 * of the reference's per-read loop and the per-read target selection that consumes them:
 *
 *   /root/reference/src/ganon-classify/GanonClassify.cpp:693-700
 *        hashes = seq | seqan3::views::minimiser_hash(k, w, adjust_seed(k))       -> gn_submit_batch
 *   /root/reference/src/ganon-classify/GanonClassify.cpp:514   (IBF)  agent.bulk_count(hashes)
 *   /root/reference/src/ganon-classify/GanonClassify.cpp:553   (HIBF) agent.bulk_count(hashes, thr)
 *   /root/reference/src/ganon-classify/GanonClassify.cpp:516-540 / :556-576  per-target sum, cap,
 *        cutoff test of select_matches                                              -> gn_fetch_batch
 *
 * Plain C: caller-owned host buffers, no exceptions, no globals besides a thread-local error
 * string.  Every function returns 0 on success or a negative GN_E* code; gn_last_error() gives
 * the message.  A gn_filter is immutable after upload and may be shared by any number of
 * gn_streams / host threads (mirrors "const member functions are thread-safe, one agent per
 * thread", hierarchical_interleaved_bloom_filter.hpp:79-83,501-504).  One gn_stream per host
 * worker thread; a stream owns a HIP stream, device batch buffers and pinned staging.
 *
 * There is no CPU fallback: every entry point fails (GN_ENODEV) when no HIP device is usable.
 */
#ifndef GANON_HIP_H
#define GANON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_OK 0
#define GN_EINVAL (-22)   /* bad argument */
#define GN_ENOMEM (-12)   /* host/device allocation failed */
#define GN_ENODEV (-19)   /* no usable HIP device / HIP runtime error */
#define GN_ERANGE (-34)   /* configuration outside what the kernels support (message says which) */
#define GN_EOVERFLOW (-75) /* caller's match buffer too small; *n_matches holds the needed size */

typedef struct gn_filter gn_filter;
typedef struct gn_stream gn_stream;

/* One flat interleaved Bloom filter as laid out by seqan3::interleaved_bloom_filter<uncompressed>
 * (the .ibf payload verbatim): word(row r, batch b) = rows[r*bin_words + b],
 * bit j of that word <-> bin 64*b + j. */
typedef struct
{
    const uint64_t* rows;      /* host pointer, bin_size*bin_words words; NULL = allocate zero-filled on device */
    uint64_t        bin_size;  /* S: rows (bin_size_, the number of bits of one Bloom filter) */
    uint64_t        bin_words; /* W = ceil(bins/64) */
    uint64_t        bins;      /* B: bin_count() */
    uint32_t        hash_funs; /* h <= 5 */
    uint32_t        hash_shift; /* countl_zero(bin_size) */
} gn_ibf_desc;

/* sparse result of select_matches for one (read, target): count already summed over the target's
 * bins and capped at n_hashes (GanonClassify.cpp:519-526 / :561-564), and >= the read's cutoff. */
typedef struct
{
    uint32_t read;   /* index in the batch */
    uint32_t target; /* caller's target id (IBF: bin2target[], HIBF: user bin index) */
    uint32_t count;
} gn_match;

/* per-read status written by gn_fetch_batch (GanonClassify.cpp:690,706) */
#define GN_READ_OK 0
#define GN_READ_SMALL 1 /* len(read1) < window_size: skipped, regardless of the mate (:690,743-747) */
#define GN_READ_BIG 2   /* more than 65535 minimisers: skipped (:674,706,737-741) */

/* device timings of the last gn_stream_classify, measured with hipEvents on the stream's HIP stream */
typedef struct
{
    float    ms_minimiser; /* slot scan + minimiser kernels (side stream; overlaps the count kernels of earlier chunks) */
    float    ms_count;     /* first IBF/HIBF count+select kernel start -> last one's end (main stream) */
    float    ms_total;     /* first kernel start -> last kernel end */
    uint64_t n_hashes;     /* minimisers of the batch (sum over reads that were counted) */
    uint64_t algo_bytes;   /* algorithmic row bytes: sum n*h*W*8 over every IBF visited */
    uint64_t n_matches;
    uint32_t n_count_launches; /* count/select launches of the batch (chunks of the minimiser||count pipeline); ms_count spans all */
    uint64_t fetched_bytes;    /* row bytes actually requested: algo_bytes minus the rows skipped by the exact early exit
                                  of reads that can no longer reach their cutoff (flat IBF fast path) */
    float    ms_compact;       /* 0, or the contiguous copy of a segmented result that a consumer asked for after the batch (a flat IBF with
                                  one unit per read leaves a read's matches where the count kernel wrote them; gn_fetch_batch(matches),
                                  gn_stream_device_matches and the merges make the copy on demand): not part of ms_total */
} gn_timings;

int         gn_device_count(int* n);
const char* gn_last_error(void);
/* The constants of seqan3's hash_and_fit this library was BUILT with (include/ganon_ibf_hash.h: five seeds, the multiplier), so that a
 * caller or a test can compare them with its own; needs no device.  Replaces nothing in the reference: the numbers live inside SeqAn3
 * (GanonClassify.cpp:514 -> bulk_count -> hash_and_fit). */
int         gn_ibf_hash_constants(uint64_t seeds[5], uint64_t* multiplier);

/* ---- filters ------------------------------------------------------------------------------- */

/* Flat IBF.  bin2target[b] (b < bins) = caller's target id of technical bin b, or 0xFFFFFFFF for
 * bins that belong to no target (replaces Filter::map, GanonClassify.cpp:279-287,1021-1025);
 * target ids must be < n_targets.  A target may own any set of bins (split bins, :516-523).
 * bin2target == NULL makes a STORAGE-ONLY filter (the builder's use): rows can be written, hashes inserted, rows read
 * back, but no stream can be created on it -- and the bin-count limit of the classify kernels does not apply. */
int gn_filter_upload_ibf(int device, const gn_ibf_desc* ibf, const uint32_t* bin2target, uint32_t n_targets,
                         gn_filter** out);

/* HIBF (raptor 3.0.1 layout, hierarchical_interleaved_bloom_filter.hpp:124-136,188):
 * next_ibf_id[i][b] and bin2userbin[i][b] (-1 = merged bin) for every bin b < ibfs[i].bins.
 * Reported target id = user bin index (ibf_bin_to_filename_position). */
int gn_filter_upload_hibf(int device, uint32_t n_ibf, const gn_ibf_desc* ibfs, const int64_t* const* next_ibf_id,
                          const int64_t* const* bin2userbin, uint64_t n_user_bins, gn_filter** out);

/* OR minimiser hashes into technical bins of a flat IBF on the device (interleaved_bloom_filter::emplace,
 * call site src/ganon-build/GanonBuild.cpp:694).  n (hash, bin) pairs in host memory. */
int gn_filter_emplace(gn_filter* f, const uint64_t* hashes, const uint32_t* bins, uint64_t n);

/* Same for IBF `ibf_idx` of an HIBF (ibf_idx = 0 for a flat filter). */
int gn_filter_emplace_ibf(gn_filter* f, uint32_t ibf_idx, const uint64_t* hashes, const uint32_t* bins, uint64_t n);

/* copy rows [row_begin, row_begin+n_rows) of IBF `ibf_idx` back to the host (tests / sampling parity) */
int gn_filter_download_rows(const gn_filter* f, uint32_t ibf_idx, uint64_t row_begin, uint64_t n_rows, uint64_t* out);
/* gather rows row_idx[0..n) of IBF `ibf_idx` (bin_words words each) into out[n*bin_words]: the sampling parity
 * checks on filters too large to download (BASELINE configs 4/5) fetch only the rows their sample touches */
int gn_filter_download_row_list(const gn_filter* f, uint32_t ibf_idx, const uint64_t* row_idx, uint64_t n, uint64_t* out);

/* ---- streaming load (filter loaders, /root/reference/src/ganon-classify/GanonClassify.cpp:949-986 load_filter:
 * the reference deserialises the whole sdsl bit_vector into host memory; here the payload goes to HBM chunk by chunk).
 * gn_filter_upload_ibf/hibf with rows == NULL allocate a zero-filled matrix; gn_filter_write_rows then copies
 * rows [row_begin, row_begin+n_rows) of IBF `ibf_idx` from a host chunk whose rows are src_row_words wide, keeping
 * words [word_lo, word_lo + bin_words) of every row (word_lo > 0 / src_row_words > bin_words = one column slice of
 * a bin-range partitioned filter).  With `src` from gn_pinned_alloc the copy is asynchronous on the
 * filter's load stream: the caller may refill the OTHER staging buffer meanwhile and must call
 * gn_filter_write_sync before reusing `src`.  gn_filter_finalize clears the padding bins and waits. */
int gn_pinned_alloc(size_t bytes, void** out);
int gn_pinned_free(void* p);
int gn_filter_write_rows(gn_filter* f, uint32_t ibf_idx, uint64_t row_begin, uint64_t n_rows, const uint64_t* src,
                         uint64_t src_row_words, uint64_t word_lo);
int gn_filter_write_sync(gn_filter* f);
int gn_filter_finalize(gn_filter* f);

/* Synthetic content for benchmarks / full-size parity tests (SURVEY 8d "generated on device for >= 8 GiB"):
 *   word(r, j) = AND_{a < and_words} mix64(mix64(seed + a) + (r * row_words_total + word_lo + j) * 0x9E3779B97F4A7C15)
 * with mix64 = the splitmix64 finaliser, i.e. iid Bernoulli(2^-and_words) bits that depend only on the GLOBAL word
 * position -- a column slice (word_lo, row_words_total of the whole filter) holds exactly the bits the unsliced
 * filter has there.  Padding bins are cleared.  Not part of the reference's interface.
 * and_words = GN_FILL_3_OF_8: word = m0 & (m1 | m2) with m_a = the mixes above, density 3/8 -- what a Bloom filter built for a
 * false-positive rate of 0.05 with three hash functions looks like (0.375^3 = 0.053).
 * and_words = GN_FILL_3_OF_16: word = m0 & m1 & (m2 | m3), density 3/16 -- an HIBF at the reference's defaults: `ganon build
 * --filter-type hibf` passes --max-fp 0.001 and --hash-functions 4 to raptor (/root/reference/src/ganon/config.py:140-143,
 * 1258-1260, build_update.py:487-489); 0.1875^4 = 0.0012. */
#define GN_FILL_3_OF_8 0x38u
#define GN_FILL_3_OF_16 0x3Fu
int gn_filter_fill_random(gn_filter* f, uint32_t ibf_idx, uint64_t seed, uint32_t and_words, uint64_t word_lo,
                          uint64_t row_words_total);

int gn_filter_info(const gn_filter* f, int* is_hibf, uint32_t* n_ibf, uint64_t* n_targets, uint64_t* device_bytes);
int gn_filter_free(gn_filter* f);

/* ---- streams ------------------------------------------------------------------------------- */

/* max_matches = capacity of the device match buffer; 0 picks max_reads*4 (it grows on demand). */
int gn_stream_create(gn_filter* f, uint32_t max_reads, uint64_t max_bases, uint64_t max_matches, gn_stream** out);
int gn_stream_destroy(gn_stream* s);

/* Batch layout: `bases` holds ASCII nucleotides (dna15 alphabet, any case; converted to dna4 ranks on
 * the device exactly like seqan3::dna4).  Read i = bases[off1[i] .. off1[i+1]) and, if
 * off2 != NULL, its mate = bases[off2[i] .. off2[i+1]).  off1/off2 have n_reads+1 entries.
 * k, w: minimiser shape; the seed is adjust_seed(k) (src/utils/include/utils/adjust_seed.hpp:33-37).
 * rel_cutoff: per-read cutoff T = max(1, ceil(n_hashes*rel_cutoff)) computed in IEEE double on the
 * device, bit-identical to GanonClassify.cpp:492-495,720-724. */

/* H2D only (asynchronous on the stream). */
int gn_stream_upload_reads(gn_stream* s, const uint8_t* bases, uint64_t n_bases, const uint64_t* off1,
                           const uint64_t* off2, uint32_t n_reads);
/* Minimiser hashes only (asynchronous): the build-side use of the same view
 * (/root/reference/src/ganon-build/GanonBuild.cpp:198-200,236); results via gn_stream_fetch_hashes. */
int gn_stream_minimisers(gn_stream* s, uint32_t k, uint32_t w);
/* Kernels only, on the reads currently resident in the stream (asynchronous). */
int gn_stream_classify(gn_stream* s, uint32_t k, uint32_t w, double rel_cutoff);
/* upload + classify (asynchronous) -- the call the reference's loop body maps to */
int gn_submit_batch(gn_stream* s, const uint8_t* bases, uint64_t n_bases, const uint64_t* off1, const uint64_t* off2,
                    uint32_t n_reads, uint32_t k, uint32_t w, double rel_cutoff);
/* Several filters on ONE device see the same batch -- the filters of a hierarchy level (same k, w: GanonClassify.cpp:1479-1494), or
 * the column parts of a wide / partitioned filter: the reference hashes a read once and hands the hashes to every filter's agent
 * (:693-735).  gn_stream_classify_shared counts the batch that is resident in `source` (uploaded and hashed there by
 * gn_submit_batch / gn_stream_classify / gn_stream_minimisers) against the filter of `s` as well, without uploading or hashing
 * again (asynchronous; k, w are the source's).  Everything that follows -- gn_fetch_batch, the pre-pass, gn_gather -- works on `s`
 * as after gn_stream_classify.  The caller fetches (or syncs) every sharing stream before it puts the next batch into `source`. */
int gn_stream_classify_shared(gn_stream* s, gn_stream* source, double rel_cutoff);

/* Reads as they lie in the file: uncompressed four-line FASTQ text instead of packed bases.  Replaces, for such input, the record
 * parsing of the reference's input side (seqan3::sequence_file_input in parse_reads, GanonClassify.cpp:1220-1287) together with
 * gn_stream_upload_reads: `text` (n_bytes <= the stream's max_bases, first byte = first byte of a record) is copied to the device
 * and the records are found there.  A record is taken when its four lines are  @id / letters / +... / quality  with as many
 * quality characters as letters and dna15 letters only (a '\r' before the '\n' of the letters line does not count); the first
 * record that is not -- or the end of the text inside a record, or the stream's max_reads -- ends the batch.  Asynchronous.
 *   gn_stream_fastq_index   waits for the tokeniser: the batch's reads (records before the first one not taken), their bases,
 *                           and parsed_bytes = offset of the first byte that is not part of them (== n_bytes: all of the text).
 *                           From here on the stream holds the batch as after gn_stream_upload_reads (single-end):
 *                           gn_stream_classify / gn_stream_minimisers / gn_fetch_batch apply, read i = record i.
 *   gn_stream_fastq_keep    the caller takes fewer records than were found (before gn_stream_classify)
 *   gn_stream_fastq_records per read: offset of its record ('@'), of its first letter, and its number of letters, in `text`
 *                           (any may be NULL; waits for the stream) -- ids and letters stay where they are, in the caller's text */
int gn_stream_upload_fastq(gn_stream* s, const uint8_t* text, uint64_t n_bytes);
/* The same for either text format.  GN_TEXT_FASTA: records of two lines,  >id / letters  -- every record's letters on one line and
 * the next line a '>' (or the end of the text); a wrapped sequence, a blank line or a ';' header ends the batch before its record,
 * like a FASTQ record the four-line rule does not cover.  gn_stream_fastq_index / _keep / _records apply unchanged (rec_at = the '>'). */
#define GN_TEXT_FASTQ 0
#define GN_TEXT_FASTA 1
int gn_stream_upload_text(gn_stream* s, const uint8_t* text, uint64_t n_bytes, int format);
/* Paired input as text: a piece of each mate file, both beginning with a record and holding the SAME records by number (the caller
 * cuts the second file where the first file's piece ends, counted in records: parse_reads takes file 2 with take(n_reads),
 * GanonClassify.cpp:1240-1252).  Both texts are tokenised; the batch is the pairs BOTH hold before the first group of lines
 * that is no record in either: read i = record i of text1 with record i of text2 as its mate (its id is text1's, :1244).
 * n_bytes1 rounded up to 16, plus n_bytes2, must fit the stream's max_bases.
 *   gn_stream_text_pair_index     waits; the pairs taken and the bytes of each text they cover (== n_bytes: all of it)
 *   gn_stream_fastq_keep / gn_stream_fastq_records (text1's records) apply; gn_stream_text_pair_records2 = the mates' in text2 */
int gn_stream_upload_text_pair(gn_stream* s, const uint8_t* text1, uint64_t n_bytes1, const uint8_t* text2, uint64_t n_bytes2, int format);
/* The same for a text that is in device memory already (of device src_device; the stream's own or a peer): the piece of a gzip file
 * that gn_inflate_step left there (gn_inflate_text_device + the offsets of gn_inflate_cuts).  The caller does not hold the text, so
 *   gn_stream_fastq_headers   (after gn_stream_fastq_index / _keep) copies the batch's header lines -- '@' or '>' to the newline, both
 *                             included -- back to back to dst, record i's at hdr_off[i] (hdr_off has n_reads + 1 entries; *n_bytes =
 *                             hdr_off[n_reads]; GN_EOVERFLOW with *n_bytes set when cap is too small): ids as in parse_reads
 *                             (GanonClassify.cpp:1244,1262: the whole header line) without the text crossing the link */
int gn_stream_upload_text_device(gn_stream* s, const uint8_t* d_text, uint64_t n_bytes, int format, int src_device);
/* both mate files' pieces in device memory (as gn_stream_upload_text_pair otherwise) */
int gn_stream_upload_text_pair_device(gn_stream* s, const uint8_t* d_text1, uint64_t n_bytes1, const uint8_t* d_text2, uint64_t n_bytes2, int format,
                                      int src_device);
/* ... the two pieces on two devices (the mate files of a pair inflated by different inflaters, gn_inflate_set_turns) */
int gn_stream_upload_text_pair_devices(gn_stream* s, const uint8_t* d_text1, uint64_t n_bytes1, int src_device1, const uint8_t* d_text2, uint64_t n_bytes2,
                                       int src_device2, int format);
int gn_stream_fastq_headers(gn_stream* s, uint8_t* dst, uint64_t cap, uint32_t* hdr_off, uint64_t* n_bytes);
/* The resident batch's letters as the kernels see them (ASCII, mate-1 block then mate-2 block: the layout of gn_stream_upload_reads) with
 * off1 / off2 (n_reads + 1 entries each; off2 only for pairs): for a caller that classified a text it does not hold and needs the letters
 * after all -- reads left unclassified go on to the next hierarchy level (GanonClassify.cpp:811-820).  *n_bytes = bytes copied
 * (GN_EOVERFLOW with *n_bytes set when cap is too small). */
int gn_stream_fetch_letters(gn_stream* s, uint8_t* bases, uint64_t cap, uint64_t* off1, uint64_t* off2, uint64_t* n_bytes);
int gn_stream_text_pair_index(gn_stream* s, uint32_t* n_reads, uint64_t* parsed_bytes1, uint64_t* parsed_bytes2);
int gn_stream_text_pair_records2(gn_stream* s, uint32_t* rec_at, uint32_t* seq_at, uint32_t* seq_len);
int gn_stream_fastq_index(gn_stream* s, uint32_t* n_reads, uint64_t* n_bases, uint64_t* parsed_bytes);
int gn_stream_fastq_keep(gn_stream* s, uint32_t n_reads);
int gn_stream_fastq_records(gn_stream* s, uint32_t* rec_at, uint32_t* seq_at, uint32_t* seq_len);
int gn_stream_sync(gn_stream* s);

/* Wait for the batch and copy results out.  n_hashes[n_reads], status[n_reads], match_off[n_reads+1]
 * (any may be NULL); matches[cap] receives the matches grouped by read (ascending read, then ascending
 * target).  Returns GN_EOVERFLOW with *n_matches = required capacity if cap is too small. */
int gn_fetch_batch(gn_stream* s, uint32_t* n_hashes, uint8_t* status, uint64_t* match_off, gn_match* matches,
                   uint64_t cap, uint64_t* n_matches);

/* Reads with more than 65535 minimisers.  By default they come back with status GN_READ_BIG and no matches, like the
 * reference's default build (TIntCount = uint16_t, GanonClassify.cpp:45-49,674).  on != 0 gives what the reference does when
 * compiled with -DLONGREADS (uint32 counters): such reads are counted with 32-bit counters (flat IBF: by a kernel of their
 * own; HIBF: by the LDS-counter level kernel), come back with status GN_READ_OK, and their match counts may exceed 65535.
 * For an HIBF the option also switches the 16-bit wrap of the per-user-bin sums off for every read, as the uint32 build does
 * (hibf.hpp:438,442: value_t). */
int gn_stream_set_long_reads(gn_stream* s, int on);

/* Optional device-side pre-pass of filter_matches (/root/reference/src/ganon-classify/GanonClassify.cpp:579-613 with the
 * threshold of :755-761) on every following batch of this stream: with max/min = the read's largest/smallest match count
 * (min starts at n_hashes, :704), matches below max - ceil((max-min)*rel_filter) are dropped (exact), and, if fpr_query < 1,
 * matches whose q = 1 - BinomCDF(count; n_hashes, target_fpr[target]) is above fpr_query BY A SAFE MARGIN are dropped too
 * (q > fpr_query*1.001 + 1e-9; the caller applies the exact rule to what is left, so the final result is the reference's).
 * A surviving match whose q is below fpr_query by the same margin (q < fpr_query*0.999 - 1e-9) comes back with
 * GN_MATCH_FPR_OK set in its count: the caller may skip its own evaluation for it (mask the bit off to get the count).
 * gn_fetch_batch / gn_stream_device_matches then return the survivors only.  Use it only where this filter sees all of a
 * read's matches (one filter per hierarchy level, filter not cut into column parts).  target_fpr: n_targets doubles
 * (flat IBF) / n_user_bins doubles (HIBF); may be NULL when fpr_query >= 1.  pf == NULL switches the pass off. */
#define GN_MATCH_FPR_OK 0x80000000u
typedef struct gn_postfilter
{
    double        rel_filter;
    double        fpr_query;
    const double* target_fpr;
    int           joint; /* 0: the pass runs with every batch.  1 / 2: one of several filters of a hierarchy level (see below) */
    const uint32_t* target_gid; /* joint == 2: level-wide id (< 2^28) of every target of this filter (same size as target_fpr) */
} gn_postfilter;
int gn_stream_set_postfilter(gn_stream* s, const gn_postfilter* pf);
/* A hierarchy level with several filters: the reference merges their matches before it thresholds (:716-735,755-761), so a
 * read's max/min are the level's.  If the filters' targets are DISJOINT (no target can be reported by two of them) the
 * pre-pass still works per filter: submit the same batch to one stream per filter (same device, post-filters set with
 * joint = 1), then call this once: it finds every stream's max/min per read, combines them, and applies the rules to every
 * stream with the level's values.  Afterwards gn_fetch_batch / gn_fetch_postfilter work per stream as above (max_count is the
 * level's maximum on every stream).
 * If the filters SHARE targets, set joint = 2 and target_gid on every stream (an id per target name, the same id for the same
 * name in every filter; a filter names a target once): the call then replays the reference's merge per read (:531-537: a
 * target keeps its largest count, the entry of the earliest stream in `streams` on ties; max/min follow the entries that got
 * in), keeps only the winning entry of every target -- on the stream of the filter that reported it -- and applies the rules
 * to the winners.  Reads with more than 4096 matches over all streams are left untouched and come back with bit 31 of
 * max_count set on every stream: the caller merges and thresholds those itself.  The dropped-match totals of a merging pass
 * are reported on streams[0]. */
int gn_streams_postfilter_joint(gn_stream* const* streams, uint32_t n_streams);
/* The streams of a joint pass over DISJOINT targets may sit on several devices -- the column parts of a bin-range
 * partitioned filter (below): every device combines its streams' max/min per read, the devices exchange those two arrays
 * (8 bytes per read and pair of devices, hipMemcpyPeerAsync over xGMI), and every stream applies the rules with the level's
 * values.  Up to 16 streams per device and 16 devices.  A merging pass (joint = 2) needs all its streams on one device. */
/* after a batch with the pass on: per read the largest match count BEFORE filtering (0 = the read had no match; the
 * reference's max_count_read, :753,776,806), and how many matches each rule dropped in this batch */
int gn_fetch_postfilter(gn_stream* s, uint32_t* max_count, uint64_t* dropped_rel_filter, uint64_t* dropped_fpr_query);

/* Device-resident view of the same result for callers that forward it without a host round trip (the sparse-match
 * exchange of a bin-range partitioned filter sends it over RCCL straight from HBM): waits for the batch,
 * then *d_matches points to n_matches records in DEVICE memory, grouped by read (ascending read, then target).
 * The memory belongs to the stream and is valid until its next submit/classify/destroy. */
int gn_stream_device_matches(gn_stream* s, const gn_match** d_matches, uint64_t* n_matches);

/* ---- bin-range partitioned flat IBF (BASELINE config 5) ------------------------------------------------
 * The reference loads a filter of any size into host memory (/root/reference/src/ganon-classify/GanonClassify.cpp:949-986,
 * 1007-1039) and one bulk_count sees all of its bins (:514).  A filter larger than one GPU's HBM is cut by technical-bin
 * range, at boundaries between targets, into column parts: part g is a flat IBF of its own (gn_filter_upload_ibf with the
 * part's bins and a part-local target numbering, rows streamed with gn_filter_write_rows(word_lo = the part's first word)),
 * created on whichever device has room.  Every part classifies every batch (one gn_stream per part, same reads); the cutoff
 * and the per-target sum of :516-527 are complete inside a part because a target's bins never straddle a cut.
 * gn_gather puts a read's matches back together on the batch's OWNER device: the parts' grouped matches and per-read
 * offsets are copied device to device (hipMemcpyPeerAsync: xGMI, not the host) and one kernel concatenates them per read in
 * part order -- ascending target, since targets ascend with the bins -- translating part-local target ids through
 * target_map[i] (n_map[i] entries; NULL = ids are the caller's already).  Parts that live on the owner device are read in
 * place.  With a filter_matches pre-pass set on the parts' streams (joint = 1), call gn_streams_postfilter_joint over them
 * first: then only survivors travel.  The result is what gn_fetch_batch would return for the unpartitioned filter. */
typedef struct gn_gather gn_gather;
int gn_gather_create(int device, uint32_t n_parts, const uint32_t* const* target_map, const uint32_t* n_map, gn_gather** out);
/* streams[i] = part i's stream, all holding the same submitted batch (gn_submit_batch / gn_stream_classify) */
int gn_gather_run(gn_gather* g, gn_stream* const* streams, uint32_t n_streams);
/* The same over parts that are plain DEVICE buffers on the gather's device -- the receive buffers of a multi-process job
 * (one process per GPU, the parts' records arrive over RCCL: ganon_amd/partition.py).  d_off[i]: n_reads+1 offsets of part
 * i, of any origin (read r of part i = d_matches[i][d_off[i][r] - d_off[i][0] .. d_off[i][r+1] - d_off[i][0])), n_matches[i]
 * its record count.  The caller's own stream must be done with the buffers. */
int gn_gather_run_buffers(gn_gather* g, const uint64_t* const* d_off, const gn_match* const* d_matches, const uint64_t* n_matches,
                          uint32_t n_parts, uint32_t n_reads);
/* per-read offsets (n_reads+1, device memory) that go with gn_stream_device_matches; valid as long as those */
int gn_stream_device_offsets(gn_stream* s, const uint64_t** d_match_off);
/* match_off[n_reads+1], matches[cap] grouped by read (ascending read, then ascending target); either may be NULL */
int gn_gather_fetch(gn_gather* g, uint64_t* match_off, gn_match* matches, uint64_t cap, uint64_t* n_matches);
/* device-resident view (valid until the next gn_gather_run) and the bytes the last run moved between devices */
int gn_gather_device_matches(gn_gather* g, const gn_match** d_matches, const uint64_t** d_match_off, uint64_t* n_matches,
                             uint64_t* peer_bytes);
int gn_gather_destroy(gn_gather* g);
/* What gn_gather did between devices so far, per ordered pair (dst = the gathering device): *state = 0 never needed, 1 peer
 * access enabled (hipDeviceEnablePeerAccess: direct copies over the xGMI link of the pair), 2 refused or not available (the
 * runtime stages the copy); *bytes = what travelled src -> dst.  `ganon-classify --verbose` prints it per pair. */
int gn_peer_stats(int dst, int src, int* state, uint64_t* bytes);

/* Switches (test and measurement hook; no counterpart in the reference).  The library reads its environment in ONE place:
 * $GANON_HIP_ABLATE, a comma list parsed once when the library is loaded.  gn_ablate() replaces that list in-process so a
 * test can cross-check a fast path against the path it replaces (NULL or "" = the product path).  Call it only while no
 * launch is in flight.  Names that switch a path OFF: early_exit cand_select csr_identity uniform_select run_select
 * max_first const_nb split_kernel predrop deferred_grids hibf_reg hibf_pack hibf_one_pack hibf_persistent; test set-ups:
 * hibf_stage hibf_reread pinned_malloc gather_copy joint_apart inflate_ahead on_demand hibf_dense_rows chunk=N hibf_pair_limit=N hibf_bpc=N; runtime: sync=spin|yield|block debug; measurement with WRONG results: fake_count (flat IBF: the count +
 * select kernels are not run, every second read gets one made-up match -- what is left is what the host around the device sustains).  Unknown name -> GN_EINVAL and
 * the previous list stays. */
int gn_ablate(const char* list);
/* free / total memory of a device: the host decides with it whether a filter is replicated or partitioned */
int gn_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
/* Words from one row to the next that gn_filter_upload_hibf gives an IBF of `bin_words` = ceil(bins / 64) words a row on the
 * device: rows are padded to whole 128-byte lines there (3 -> 4, 5..8 -> 8, 9..16 -> 16 words, multiples of 16 beyond), so that
 * no row request moves two lines.  An IBF of bin_size rows takes bin_size * this * 8 bytes -- what the host's placement must
 * count instead of the file's payload.  Everything the ABI takes or returns stays bin_words words a row.  (No counterpart in the
 * reference: /root/reference/src/ganon-classify/GanonClassify.cpp:949-986 reads the filter into host memory as it is.) */
uint64_t gn_hibf_row_stride_words(uint64_t bin_words);

/* ---- reassign: the EM over a classification's .all ---------------------------------------------------
 * `ganon classify` runs this after the binary by default (/root/reference/src/ganon/classify.py:76-88, --multiple-matches
 * em); the algorithm is /root/reference/src/ganon/reassign.py:96-145 with get_top_match :226-241.
 * The table is CSR over reads: off[n_reads+1] into target[n_entries]; a read's entries are in the order its lines appear in
 * the .all file, reads in first-appearance order, targets numbered 0..n_targets-1 by first appearance (:76-92) -- the
 * tie-break of get_top_match (first listed entry wins) and the left-to-right double sum of the convergence test depend
 * on exactly that order.  The table is copied to the device once and every iteration runs there.
 * gn_reassign_run: max_iter = 0 iterates until diff <= threshold (:141-145); diffs[i] = the i-th iteration's
 * sum |old prob - new prob| bit for bit as the reference computes it (gn_reassign_diffs copies the first `cap` of them).
 * gn_reassign_fetch (any pointer may be NULL): counts[t] = reassigned_matches of the LAST iteration (:113-121; the new .rep
 * holds counts[t] - unique[t] in its lca column, :189-219), unique[t] = reads listing t and nothing else, prob[t] after the
 * last update, choice[r] = index into target[] of the entry read r is given in the .one file (:153-181).
 * PRECONDITION (since round 5): every read has at least one entry, off[r + 1] > off[r] -- the reference's table holds a read only once a
 * line of the .all file names it (:76-92); a CSR with an empty row is refused with GN_EINVAL (an empty row has no entry choice[r] could
 * index, and the .one writer would have nothing to print).  A caller that holds reads without matches drops them before the call. */
typedef struct gn_reassign gn_reassign;
int gn_reassign_create(int device, uint64_t n_reads, uint64_t n_entries, uint32_t n_targets, const uint64_t* off,
                       const uint32_t* target, gn_reassign** out);
int gn_reassign_run(gn_reassign* g, uint32_t max_iter, double threshold, uint32_t* iterations);
int gn_reassign_diffs(const gn_reassign* g, double* diffs, uint32_t cap);
int gn_reassign_fetch(gn_reassign* g, uint64_t* counts, uint64_t* unique, double* prob, uint64_t* choice);
/* reads with one / several entries, reads long enough to take a wave of their own, device time of the last run (EM +
 * final choice) and the algorithmic bytes one iteration reads (8 per read, 4 per entry of a read with several + 4 for its last choice) */
int gn_reassign_info(const gn_reassign* g, uint64_t* n_unique_reads, uint64_t* n_multi_reads, uint64_t* n_wave_reads, float* ms_em,
                     uint64_t* bytes_per_iteration);
int gn_reassign_free(gn_reassign* g);

/* Build side (/root/reference/src/ganon-build/GanonBuild.cpp).
 * gn_stream_distinct_hashes: after gn_stream_minimisers, the SET of minimiser hashes of all sequences resident in the
 * stream, ascending -- what count_hashes collects per file in a robin_hood::unordered_set (:184-249; the set's own iteration
 * order is not reproducible here, ascending order is this library's definition).  out may be NULL to query *n_distinct.
 * A long sequence is passed as overlapping pieces (consecutive pieces share window_size-1 bases): every window lies in one
 * piece, so the set is the sequence's (emission order and duplicates differ, the set does not).
 * gn_filter_emplace_split: hash i of `hashes` is inserted into technical bin first_bin + i / hashes_per_bin -- a target's
 * run of bins with equal shares, create_bin_map_hash (:619-653) + build (:655-698). */
int gn_stream_distinct_hashes(gn_stream* s, uint64_t* out, uint64_t cap, uint64_t* n_distinct);
int gn_filter_emplace_split(gn_filter* f, const uint64_t* hashes, uint64_t n, uint32_t first_bin, uint64_t hashes_per_bin);
/* The membership check of the reference's build test (validate_elements, /root/reference/tests/ganon-build/GanonBuild.test.cpp:53-98:
 * hash a sequence, bulk_count, add up the counts of its target's bins, compare with the number of hashes) for a flat IBF on the
 * device: *hits = sum over the n hashes (host memory) of the number of `bins` that contain the hash, *missing = hashes no bin
 * contains, *first_missing = the smallest index of one (~0 when none).  `ganon-classify --verify-filter` calls it per target. */
int gn_filter_probe(gn_filter* f, const uint64_t* hashes, uint64_t n, const uint32_t* bins, uint32_t n_bins, uint64_t* hits,
                    uint64_t* missing, uint64_t* first_missing);

/* Parity / debugging taps (tests only): minimiser hashes of the resident batch in emission order
 * (hash_off[n_reads+1]; hashes[cap]) and dense per-bin counts of reads [read_begin, read_end)
 * (flat IBF: uint16[bins] per read == counting_agent::bulk_count; HIBF: uint16[n_user_bins] per read
 * == counting_agent_type::bulk_count(values, T)). */
int gn_stream_fetch_hashes(gn_stream* s, uint64_t* hash_off, uint64_t* hashes, uint64_t cap, uint64_t* n_total);
int gn_stream_dense_counts(gn_stream* s, uint32_t read_begin, uint32_t read_end, uint16_t* counts);

int gn_stream_timings(gn_stream* s, gn_timings* t);
/* HIBF only: the last batch per tree level (level 0 = the top IBF) -- duration of the level's kernels, algorithmic row
 * bytes of the level (they add up to gn_timings.algo_bytes), bytes of the IBFs at that depth and their usual row width.
 * A level whose tables fit the 256 MiB Infinity Cache is not bound by HBM; the roofline is stated per level for that
 * reason.  Arrays of `cap` entries; *n_levels = levels of the filter's tree (at most 8 are timed separately). */
int gn_stream_hibf_levels(gn_stream* s, uint32_t* n_levels, float* ms, uint64_t* algo_bytes, uint64_t* table_bytes,
                          uint32_t* row_bytes, uint32_t cap);

/* ... and the same row requests in the 128-byte lines the rows occupy (a 32-byte row moves a line, a 136-byte row two): with IBFs of
 * different widths on one level this, not row_bytes, is what the level's physical rate is computed from. */
int gn_stream_hibf_level_lines(gn_stream* s, uint64_t* line_bytes, uint32_t cap);

/* ---- gzip input inflated on the device (csrc/gn_inflate.hip) ------------------------------------------------------------------
 * Replaces, for `reads.fq.gz` / `reads.fa.gz`, the zlib stream the reference reads its input through
 * (seqan3::sequence_file_input over a gz stream in parse_reads, /root/reference/src/ganon-classify/GanonClassify.cpp:1220-1287;
 * its one decompression thread, :1433): the caller feeds the file's bytes as they are, the device finds deflate block starts,
 * decodes the chunks between them in parallel and returns the text -- the same bytes zlib's inflate() yields for the file
 * (every member; what follows the last member is ignored like gzip does).  Every member's CRC-32 and ISIZE are checked.
 *   gn_inflate_create   compressed_bytes = size of the file; chunk_bytes = compressed bytes per parallel chunk (0: 32 KiB);
 *                       step_bytes = compressed bytes decoded per gn_inflate_step (0: 128 MiB for files below 1.5 GB, 256 MiB above; at most 8192 chunks).  The whole compressed file
 *                       stays resident in HBM (files of 64 GiB and more: GN_ERANGE).
 *   gn_inflate_feed     appends the next n bytes of the file (host memory; returns when `data` may be reused)
 *   gn_inflate_step     decodes the next step: every chunk whose bytes, and 4 MiB behind them, are fed -- all of the rest once the
 *                       whole file is.  *n_text = bytes of text this step produced (they follow the previous step's), *done = 1
 *                       after the last member's trailer.  Blocks until the text is complete in device memory.
 *   gn_inflate_text     copies [off, off + n) of the LAST step's text to host memory
 *   gn_inflate_text_device   the last step's text in device memory (valid until the step after the next one begins)
 * GN_ERANGE from gn_inflate_step: damaged / truncated data, a wrong CRC-32 or ISIZE, or data this decoder is not made for (more than
 * 8-fold expansion, hardly any dynamic-Huffman blocks): the caller reads the file with its host inflater instead -- nothing
 * that was returned before is wrong, and nothing is returned that was not decoded. */
typedef struct gn_inflate gn_inflate;
typedef struct gn_inflate_stats
{
    uint64_t steps, chunks, fixups, markers, members, text_bytes;
    double   ms_decode, ms_chain, ms_resolve, ms_step_wall;
    /* a library built with -DGI_PROF=1 (scripts/inflate_variants.sh; zeros otherwise), summed over the chunks (wave-milliseconds): block search screen, headers of search candidates, block headers, Huffman decoding,
     * placing the tokens (LZ77 copies), flushes, whole chunk; [7] unused */
    double   prof_ms[8];
} gn_inflate_stats;
int gn_inflate_create(int device, uint64_t compressed_bytes, uint32_t chunk_bytes, uint64_t step_bytes, gn_inflate** out);
int gn_inflate_destroy(gn_inflate* z);
int gn_inflate_feed(gn_inflate* z, const uint8_t* data, uint64_t n);
int gn_inflate_step(gn_inflate* z, uint64_t* n_text, int* done);
int gn_inflate_text(gn_inflate* z, uint8_t* dst, uint64_t off, uint64_t n);
int gn_inflate_text_device(gn_inflate* z, const uint8_t** text, uint64_t* n);
int gn_inflate_get_stats(gn_inflate* z, gn_inflate_stats* out);
/* Batches for the classifier straight from the device: where do records begin in the last step's text?
 *   gn_inflate_cuts       cuts[0 .. *n_cuts): ascending offsets into the last step's text, each the first byte behind a newline whose
 *                         number (from 1, counted from the text's first byte) is a multiple of lines_per_record (4: FASTQ, 2: two-line
 *                         FASTA) -- the first such offset at or behind every multiple of piece_bytes, and the last one of the text.
 *                         The text's first byte must begin a record (true for a file's first step; later: gn_inflate_set_carry).
 *   gn_inflate_set_carry  the last n_tail bytes of the last step's text (the record the step's end cuts) are what the NEXT step's text
 *                         begins with: its n_text counts them, its text holds them in front. */
int gn_inflate_cuts(gn_inflate* z, uint32_t lines_per_record, uint64_t piece_bytes, uint64_t* cuts, uint32_t cap, uint32_t* n_cuts);
/* ... for the two files of paired reads (parse_reads takes file 2 with take(n_reads), GanonClassify.cpp:1240-1252: pairs go by record
 * NUMBER): gn_inflate_cuts_lines also says how many lines lie in front of every cut (cut_lines[i], counted from the text's first byte);
 * gn_inflate_cut_at_lines answers, for the mate file's text, where those many lines end: offsets[i] = first byte behind line number
 * lines[i] (0 for 0 lines, ~0 when the text holds fewer), *total_lines = newlines in the text. */
int gn_inflate_cuts_lines(gn_inflate* z, uint32_t lines_per_record, uint64_t piece_bytes, uint64_t* cuts, uint64_t* cut_lines, uint32_t cap, uint32_t* n_cuts);
int gn_inflate_cut_at_lines(gn_inflate* z, const uint64_t* lines, uint32_t n_lines, uint64_t* offsets, uint64_t* total_lines);
int gn_inflate_set_carry(gn_inflate* z, uint64_t n_tail);
/* One file, several devices.  A step's decode depends on nothing before it; the pieces that put a step's text together need four things
 * of the step before: the stream's bit position, the open member's length and CRC-32 so far, the 32 KiB window, and the text the step's
 * end cut off.  So N inflaters created for the SAME file (same sizes), each fed the whole file, can take its steps in turn:
 *   gn_inflate_set_turns(z_i, N, i)   before the first step: z_i will run steps i, i + N, ... (sizes its decodes-ahead by that; only
 *                                      turn 0's first step is the file's short first step)
 *   gn_inflate_handoff(from, to)      after gn_inflate_step(from) [+ gn_inflate_cuts / gn_inflate_set_carry(from)], before
 *                                      gn_inflate_step(to): copies those four things device to device (32 KiB + the carried record)
 * The text of step k then lies on device k mod N -- where the worker of that device classifies it without a peer copy.  With one
 * inflater the file is bound by that device's decode rate (31-37 GB/s of text); with N by the serial passes (order, windows, resolve,
 * CRC: ~7 ms per 0.9 GB of text).  The reference reads a .gz through one zlib stream (GanonClassify.cpp:1220-1287,1433). */
int gn_inflate_set_turns(gn_inflate* z, uint32_t n_turns, uint32_t my_turn);
int gn_inflate_handoff(gn_inflate* from, gn_inflate* to);

#ifdef __cplusplus
}
#endif
#endif
