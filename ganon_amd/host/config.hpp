// config.hpp -- run configuration of the drop-in ganon-classify binary.
// Field meaning, defaults, validation messages and the broadcast rules for per-filter / per-hierarchy values are
// those of the reference's Config (/root/reference/src/ganon-classify/include/ganon-classify/Config.hpp:18-246);
// the flag table is /root/reference/src/ganon-classify/CommandLineParser.cpp:14-46.  Logic lives in config.cpp.
#pragma once

#include <cstddef>
#include <cstdint>
#include <iosfwd>
#include <optional>
#include <string>
#include <vector>

namespace gnhost
{

constexpr const char* kVersion = "2.1.1-mi355x";

using StringList = std::vector<std::string>;
using DoubleList = std::vector<double>;

struct Config
{
    // inputs
    StringList ibf, tax;
    StringList single_reads, paired_reads, batch_reads;
    // outputs
    std::string output_prefix;
    bool        output_all = false, output_lca = false, output_unclassified = false, output_stats = false;
    bool        output_single = false;
    // classification parameters (one per filter / one per hierarchy label after validate())
    StringList hierarchy_labels{ "H1" };
    DoubleList rel_cutoff{ 0.2 };
    DoubleList rel_filter{ 0.0 };
    DoubleList fpr_query{ 1.0 };
    bool        hibf          = false;
    bool        reference_order = false; // extension: .all / .rep in the reference's robin_hood iteration order (robin_order.hpp)
    bool        long_reads    = false; // extension: the reference's compile-time -DLONGREADS (GanonClassify.cpp:45-49) as a flag
    bool        skip_lca      = false;
    std::string tax_root_node = "1";
    // execution
    uint16_t threads   = 1;    // classify workers (:1579-1597) when --device is not given: 2..4 of them on GPU 0, sharing the filters
    size_t   n_batches = 1000; // accepted for compatibility
    size_t   n_reads   = 400;  // accepted for compatibility
    bool     verbose = false, quiet = false;
    std::vector<int> devices{ 0 }; // extension: --device 0,1,.. | all (default $GANON_DEVICE or 0); empty = every visible GPU
    // extensions for a first contact with files written by the reference (filter_io.hpp, verify.cpp); both end the run
    std::string inspect_filter; // --inspect-filter F [--hibf]: parse the metadata only, print every field and check, no device
    std::string verify_filter;  // --ibf F --verify-filter refs.tsv: every minimiser of every reference file is in its target's bins
    bool             devices_given = false; // --device / $GANON_DEVICE was used (else --threads decides the workers on GPU 0)

    // checks + broadcasting; prints the reference's message to stderr and returns false on the first violation
    bool validate();
};

std::ostream& operator<<(std::ostream& stream, const Config& config); // the --verbose dump

// Returns the parsed config, or nullopt when the program should exit (exit_code set like main.cpp:7-17:
// 0 after -h/-v, 1 with no arguments or on a parse error).
std::optional<Config> parse_command_line(int argc, char** argv, int& exit_code);

} // namespace gnhost
