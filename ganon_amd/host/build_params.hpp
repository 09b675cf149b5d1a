// build_params.hpp -- how ganon-build sizes a flat IBF and lays targets out over its technical bins
// (/root/reference/src/ganon-build/GanonBuild.cpp:290-653).  Plain IEEE double / integer arithmetic; the operations are
// the reference's, in its order, so that the numbers in the file header (IBFConfig) are the ones the reference writes for
// the same minimiser counts.  oracle/build_params.py is the independent restatement the tests compare this with.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace gnbuild
{

constexpr uint8_t kMaxHashFunctions = 5; // Config.hpp:27

struct IbfParams // src/utils/include/utils/IBFConfig.hpp
{
    uint64_t n_bins = 0, max_hashes_bin = 0;
    uint8_t  hash_functions = 0, kmer_size = 0;
    uint16_t window_size = 0;
    uint64_t bin_size_bits = 0;
    double   max_fp = 0, true_max_fp = 0, true_avg_fp = 0;
};

// Bloom filter arithmetic (:290-314,365-380)
uint64_t bits_for(double max_fp, uint64_t elements);                        // optimal number of hash functions
uint64_t bits_for(double max_fp, uint64_t elements, uint8_t hash_functions); // given number of hash functions
uint8_t  hash_functions_for(uint64_t bin_size_bits, uint64_t elements, uint8_t requested);
double   bloom_fp(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t elements);
uint64_t padded_bins(uint64_t n_bins); // next multiple of 64

// One technical bin: `target` owns hashes [first, last] of its (sorted) hash list
struct BinSpan
{
    uint32_t target;
    uint64_t first, last;
};

// optimal_hashes (:427-616): simulate every 100th bin capacity from the largest target downwards and keep the capacity with
// the best trade-off between filter size (or false-positive rate, with --filter-size) and number of bins.
// counts: distinct minimisers per target, in target order (zero counts allowed).  Fills everything but kmer/window size
// and the true fp figures.  n_bins stays 0 when nothing can be built.
void choose_capacity(double max_fp, double filter_size_mb, const std::vector<uint64_t>& counts, uint8_t hash_functions,
                     const std::string& mode, IbfParams& out);
// true_false_positive (:382-412)
void true_fp(const std::vector<uint64_t>& counts, IbfParams& p);
// create_bin_map_hash (:619-653), targets in the order given; shares[t] = hashes per bin of target t
std::vector<BinSpan> lay_out_bins(const IbfParams& p, const std::vector<uint64_t>& counts, std::vector<uint64_t>* shares = nullptr);

} // namespace gnbuild
