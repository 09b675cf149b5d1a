// report.hpp -- run counters and the text they are reported in: the stderr summary, the `.sta` table and the
// --verbose timing block.
//
// What is counted and how each line / column is formatted is the reference's output contract
// (/root/reference/src/ganon-classify/GanonClassify.cpp: counters :153-246, stderr summary :1053-1128, `.sta`
// :1130-1218, timing block :1041-1051).  The code is organised around that contract, not around the reference's
// functions: one set of derived figures (Digest) feeds both renderers.
#pragma once

#include <chrono>
#include <cstddef>
#include <iosfwd>
#include <map>
#include <string>
#include <vector>

namespace gnhost
{

// Wall-clock stopwatch that can be started and stopped several times; remembers the first start and the last stop.
class Stopwatch
{
public:
    using Clock = std::chrono::system_clock;
    void   start();
    void   stop();
    double seconds() const { return accumulated_.count(); }
    // "YYYY-MM-DD HH:MM:SS" in local time
    std::string first_start_text() const { return stamp(first_start_); }
    std::string last_stop_text() const { return stamp(last_stop_); }

private:
    static std::string            stamp(Clock::time_point t);
    bool                          running_once_ = false;
    Clock::time_point             first_start_{}, lap_start_{}, last_stop_{};
    std::chrono::duration<double> accumulated_{ 0.0 };
};

// Per (hierarchy level, target): what `.rep` reports and what the level totals absorb.
struct TargetTally
{
    size_t matches = 0, lca_reads = 0, unique_reads = 0, dropped_by_rel_filter = 0, dropped_by_fpr_query = 0;
    bool   reported() const { return matches || lca_reads || unique_reads; }
    void   add(const TargetTally& o)
    {
        matches += o.matches, lca_reads += o.lca_reads, unique_reads += o.unique_reads;
        dropped_by_rel_filter += o.dropped_by_rel_filter, dropped_by_fpr_query += o.dropped_by_fpr_query;
    }
};

// Per (hierarchy level, read-set prefix).
struct ReadSetTally
{
    size_t reads_in = 0;                         // records parsed from the files (classified or not)
    size_t reads_seen = 0, bases_seen = 0, minimisers_seen = 0; // first level only
    size_t too_short = 0, too_many_minimisers = 0;              // first level only
    size_t reads_classified = 0, best_match_minimisers = 0, minimisers_of_classified = 0;
    size_t matches = 0, unique_reads = 0, dropped_by_rel_filter = 0, dropped_by_fpr_query = 0;

    void add(const ReadSetTally& o);            // every field (two partial tallies of the same level and prefix)
    void absorb_reads(const ReadSetTally& o);   // everything a level counts per read
    void absorb_targets(const TargetTally& t);  // what a level's `.rep` rows add up to
};

class RunReport
{
public:
    // a finished level hands in its per-prefix read tallies and per-prefix target tallies
    void add_level(const std::string& label, const std::map<std::string, ReadSetTally>& reads,
                   const std::map<std::string, std::vector<TargetTally>>& targets);
    void count_input(const std::string& prefix, size_t n_records);
    void touch(const std::string& prefix) { overall_[prefix]; }

    const std::map<std::string, ReadSetTally>& overall() const { return overall_; }

    // `<output_prefix><prefix>.sta`: one row per level (+ "-total-" with several levels), 18 tab-separated columns
    void write_sta(const std::string& output_prefix, const std::vector<std::string>& level_labels) const;
    // stderr summary
    void print(std::ostream& os, double classify_seconds, const std::vector<std::string>& level_labels) const;

private:
    std::map<std::string, ReadSetTally>                        overall_;   // prefix -> sum over levels
    std::map<std::string, std::map<std::string, ReadSetTally>> per_level_; // label -> prefix -> tally
    size_t all_reads_ = 0, all_bases_ = 0, all_minimisers_ = 0;
};

void print_timing_block(std::ostream& os, const Stopwatch& whole_run, const Stopwatch& loading, const Stopwatch& classifying);

} // namespace gnhost
