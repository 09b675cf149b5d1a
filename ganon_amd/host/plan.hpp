// plan.hpp -- what a run consists of, derived from the command line before any data is touched:
//   ReadPlan   which files belong to which read-set prefix (--single-reads / --paired-reads / --batch-reads)
//   LevelPlan  which filters form which hierarchy level, with the per-level and per-filter thresholds and the
//              suffixes of the per-level output files
// Behavioural contract: /root/reference/src/ganon-classify/GanonClassify.cpp:289-401 (what is grouped how, the
// error texts) and :403-473 (the --verbose listings); levels and prefixes are processed in sorted order (:1461).
#pragma once

#include "config.hpp"

#include <iosfwd>
#include <map>
#include <string>
#include <vector>

namespace gnhost
{

struct ReadFiles
{
    std::string mate1, mate2; // mate2 empty = single-end
    bool        paired() const { return !mate2.empty(); }
};

// prefix ("" without --batch-reads) -> files, in sorted prefix order
using ReadPlan = std::map<std::string, std::vector<ReadFiles>>;

// false + message on stderr when a --batch-reads table is malformed or names a missing / empty file
bool make_read_plan(const Config& config, ReadPlan& plan);

struct FilterSpec
{
    std::string ibf_file, tax_file;
    double      rel_cutoff = 0;
};

struct Level
{
    std::string             label;
    std::vector<FilterSpec> filters;
    double                  rel_filter = 0, fpr_query = 1;
    std::string             suffix_one, suffix_all; // "one"/"all", or "<label>.one"/"<label>.all"
    uint8_t                 kmer_size   = 0;        // filled when the level's filters are loaded
    uint32_t                window_size = 0;
};

// levels in sorted label order; config must have passed Config::validate() (one label / cutoff per filter,
// one rel-filter / fpr-query per distinct label in order of first appearance)
std::vector<Level> make_level_plan(const Config& config);

// --verbose listings
void list_levels(std::ostream& os, const std::vector<Level>& levels);
void list_reads(std::ostream& os, const ReadPlan& plan);
void list_outputs(std::ostream& os, const Config& config, const std::vector<Level>& levels, const ReadPlan& plan);

} // namespace gnhost
