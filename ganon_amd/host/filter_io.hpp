// filter_io.hpp -- readers for the reference's on-disk filters (no SeqAn3 / cereal dependency).
//
//   .ibf   ganon-build output: cereal BinaryOutputArchive of version tuple, IBFConfig, hashes_count, bin_map and a
//          seqan3::interleaved_bloom_filter<uncompressed>  (writer /root/reference/src/ganon-build/GanonBuild.cpp:251-288,
//          reader /root/reference/src/ganon-classify/GanonClassify.cpp:949-986, IBFConfig.hpp:28-40)
//   .hibf  raptor 3.0.1 index (reader GanonClassify.cpp:875-938, hibf.hpp:163-169,293-298)
//
// A file is read in two parts: its metadata (FilterMeta: everything but the bits) and its bit matrices, which are
// STREAMED into a FilterSink in large row chunks -- double-buffered, read with several threads straight into the
// sink's (pinned) staging memory -- so that the host never holds a whole matrix.
//
// The byte layout of the SeqAn3/sdsl parts is restated from the published serialisation code and is NOT pinned by
// any binary fixture in the reference tree, so the readers self-check every redundant quantity (technical_bins ==
// 64*bin_words, hash_shift == clz(bin_size), bit_vector size == technical_bins*bin_size, file size arithmetic),
// accept the plausible variants of the sdsl bit_vector header (see filter_io.cpp) and fail loudly on a mismatch
// instead of classifying against a misread filter.
#pragma once

#include <cstdint>
#include <iosfwd>
#include <map>
#include <string>
#include <vector>

namespace gnhost
{

struct IbfShape
{
    uint64_t bins = 0, technical_bins = 0, bin_size = 0, hash_shift = 0, bin_words = 0, hash_funs = 0;
    uint64_t payload_bytes() const { return bin_size * bin_words * 8; }
};

// IBFConfig.hpp:3-41
struct IBFConfig
{
    uint64_t n_bins = 0, max_hashes_bin = 0;
    uint8_t  hash_functions = 0, kmer_size = 0;
    uint16_t window_size = 0;
    uint64_t bin_size_bits = 0;
    double   max_fp = 0, true_max_fp = 0, true_avg_fp = 0;
};

// What load_files() (GanonClassify.cpp:1007-1039) knows about one --ibf argument, minus the bit matrices.
struct FilterMeta
{
    bool      is_hibf = false;
    IBFConfig ibf_config;
    // targets in first-appearance order of (bin number ascending); bins per target
    std::vector<std::string>           targets;
    std::vector<std::vector<uint64_t>> target_bins; // IBF: technical bins; HIBF: user bin index (one)
    std::vector<double>                target_fpr;  // :968-982 (IBF) / :932 (HIBF)
    uint64_t                           bin_count = 0;
    std::vector<IbfShape>              shapes;      // one (flat) or one per IBF of the HIBF
    // hibf
    std::vector<std::vector<int64_t>> next_ibf_id;
    std::vector<std::vector<int64_t>> bin_to_user;
    uint64_t                          n_user_bins = 0;
};

// Receives a filter: begin(meta) -> rows(...)* -> end().  rows() may return before `src` has been consumed (an
// asynchronous copy out of pinned staging); drain() returns when every rows() so far is done with its source.
class FilterSink
{
public:
    virtual ~FilterSink() = default;
    virtual bool begin(const FilterMeta& meta, std::string& err) = 0;
    // staging buffer `which` (0 or 1) of at least `bytes` for the loader to fill; nullptr = use your own memory
    virtual uint64_t* staging(int which, size_t bytes) = 0;
    // rows [row_begin, row_begin + n_rows) of IBF `ibf`, bin_words words each, contiguous at src
    virtual bool rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string& err) = 0;
    virtual bool drain(std::string& err) = 0;
    virtual bool end(std::string& err)   = 0;
};

// Where the last load_filter_file() of this thread spent its time (`ganon-classify --verbose` prints it under [startup])
struct LoadTiming
{
    double   parse_s = 0, staging_s = 0, pread_s = 0, sink_s = 0, begin_s = 0, end_s = 0;
    uint64_t payload_bytes = 0;
};
const LoadTiming& last_load_timing();

// Parse `path` (.ibf, or .hibf when `hibf`), fill `meta`, stream the bits into `sink`.  Throws std::runtime_error
// with a descriptive message on malformed input or when the sink reports an error.
void load_filter_file(const std::string& path, bool hibf, FilterMeta& meta, FilterSink& sink);

// `ganon-classify --inspect-filter`: parse the file's metadata only (no device, no bits), print every header field with its
// offset and every redundancy check (technical_bins == 64*bin_words, hash_shift == countl_zero(bin_size), the bit_vector
// header variant, S*W*8 against the bytes left, map sizes); false -- with the first inconsistent field named -- when the file
// is not what load_filter_file would accept.  First contact with a file written by SeqAn3 / raptor starts here.
bool inspect_filter_file(const std::string& path, bool hibf, std::ostream& out);

// GanonClassify.cpp:940-947
double false_positive(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t n_hashes);

// node -> (parent, rank, name); GanonClassify.cpp:988-1005
struct TaxNode
{
    std::string parent, rank, name;
};
std::map<std::string, TaxNode> load_tax(const std::string& path);

} // namespace gnhost
