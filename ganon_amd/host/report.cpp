// report.cpp -- see report.hpp.
#include "report.hpp"

#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace gnhost
{

// ---- Stopwatch -------------------------------------------------------------------------------------------------
void Stopwatch::start()
{
    lap_start_ = Clock::now();
    if (!running_once_)
    {
        first_start_  = lap_start_;
        running_once_ = true;
    }
}

void Stopwatch::stop()
{
    last_stop_ = Clock::now();
    accumulated_ += last_stop_ - lap_start_;
}

std::string Stopwatch::stamp(Clock::time_point t)
{
    const std::time_t  tt = Clock::to_time_t(t);
    std::ostringstream os;
    os << std::put_time(std::localtime(&tt), "%F %T");
    return os.str();
}

// ---- tallies ---------------------------------------------------------------------------------------------------
void ReadSetTally::absorb_reads(const ReadSetTally& o)
{
    reads_seen += o.reads_seen;
    bases_seen += o.bases_seen;
    minimisers_seen += o.minimisers_seen;
    too_short += o.too_short;
    too_many_minimisers += o.too_many_minimisers;
    reads_classified += o.reads_classified;
    best_match_minimisers += o.best_match_minimisers;
    minimisers_of_classified += o.minimisers_of_classified;
    // (matches a backend dropped before they reached the host's per-target tallies)
    dropped_by_rel_filter += o.dropped_by_rel_filter;
    dropped_by_fpr_query += o.dropped_by_fpr_query;
}

void ReadSetTally::add(const ReadSetTally& o)
{
    reads_in += o.reads_in;
    reads_seen += o.reads_seen, bases_seen += o.bases_seen, minimisers_seen += o.minimisers_seen;
    too_short += o.too_short, too_many_minimisers += o.too_many_minimisers;
    reads_classified += o.reads_classified, best_match_minimisers += o.best_match_minimisers;
    minimisers_of_classified += o.minimisers_of_classified;
    matches += o.matches, unique_reads += o.unique_reads;
    dropped_by_rel_filter += o.dropped_by_rel_filter, dropped_by_fpr_query += o.dropped_by_fpr_query;
}

void ReadSetTally::absorb_targets(const TargetTally& t)
{
    matches += t.matches;
    unique_reads += t.unique_reads;
    dropped_by_rel_filter += t.dropped_by_rel_filter;
    dropped_by_fpr_query += t.dropped_by_fpr_query;
}

void RunReport::count_input(const std::string& prefix, size_t n_records)
{
    overall_[prefix].reads_in += n_records;
}

void RunReport::add_level(const std::string& label, const std::map<std::string, ReadSetTally>& reads,
                          const std::map<std::string, std::vector<TargetTally>>& targets)
{
    auto& level = per_level_[label];
    for (const auto& [prefix, tally] : reads)
    {
        all_reads_ += tally.reads_seen;
        all_bases_ += tally.bases_seen;
        all_minimisers_ += tally.minimisers_seen;
        overall_[prefix].absorb_reads(tally);
        level[prefix].absorb_reads(tally);
    }
    for (const auto& [prefix, rows] : targets)
        for (const auto& row : rows)
        {
            overall_[prefix].absorb_targets(row);
            level[prefix].absorb_targets(row);
        }
}

namespace
{

// the figures both renderers show, derived once from a tally
struct Digest
{
    size_t classified, unique, multiple, matches, dropped_filter, dropped_fpr, matched, of_classified;
    double pct_classified, pct_unique, pct_multiple, matches_per_read, pct_matched;

    Digest(const ReadSetTally& t, double denominator)
      : classified(t.reads_classified), unique(t.unique_reads), multiple(t.reads_classified - t.unique_reads),
        matches(t.matches), dropped_filter(t.dropped_by_rel_filter), dropped_fpr(t.dropped_by_fpr_query),
        matched(t.best_match_minimisers), of_classified(t.minimisers_of_classified),
        pct_classified(t.reads_classified / denominator * 100), pct_unique(t.unique_reads / denominator * 100),
        pct_multiple((t.reads_classified - t.unique_reads) / denominator * 100),
        matches_per_read(t.reads_classified ? t.matches / static_cast<double>(t.reads_classified) : 0),
        pct_matched(t.best_match_minimisers ? t.best_match_minimisers / static_cast<double>(t.minimisers_of_classified) * 100 : 0)
    {
    }
};

// percentages are taken over the reads of the whole read set; an empty set divides by one
double denominator_of(const ReadSetTally& whole)
{
    return whole.reads_seen > 0 ? static_cast<double>(whole.reads_seen) : 1.0;
}

const ReadSetTally& tally_or_empty(const std::map<std::string, std::map<std::string, ReadSetTally>>& per_level,
                                   const std::string& label, const std::string& prefix)
{
    static const ReadSetTally empty{};
    auto                      l = per_level.find(label);
    if (l == per_level.end())
        return empty;
    auto p = l->second.find(prefix);
    return p == l->second.end() ? empty : p->second;
}

// the block of the stderr summary that describes one tally; `unclassified` = 0 leaves its lines out
void print_block(std::ostream& os, const ReadSetTally& t, double denom, size_t unclassified)
{
    const Digest d(t, denom);
    os << d.classified << " sequences classified (" << d.pct_classified << "%)\n"
       << "  " << d.unique << " with unique matches (" << d.pct_unique << "%)\n"
       << "  " << d.multiple << " with multiple matches (" << d.pct_multiple << "%)\n";
    if (unclassified > 0)
    {
        os << unclassified << " sequences unclassified (" << unclassified / denom * 100 << "%)\n";
        if (t.too_short)
            os << "  " << t.too_short << " sequences skipped (shorter than window size)\n";
        if (t.too_many_minimisers)
            os << "  " << t.too_many_minimisers << " sequences skipped (larger than allowed, check compilation with -DLONGREADS)\n";
    }
    os << "matches: " << d.matches << " (avg. " << d.matches_per_read << " reference/sequence), " << d.dropped_filter
       << " discarded (--rel-filter), " << d.dropped_fpr << " discarded (--fpr-query)\n"
       << "k-mers: " << d.matched << "/" << d.of_classified << " k-mers matched/k-mers from classified sequences (" << d.pct_matched
       << "%)\n";
    os.flush();
}

const char* const kStaColumns[] = { "prefix", "hierarchy_label", "seq_processed", "seq_unclassified", "seq_classified",
                                    "seq_classified_perc", "seq_unique_matches", "seq_unique_matches_perc", "seq_multiple_matches",
                                    "seq_multiple_matches_perc", "matches", "avg_matches_ref_seq", "dis_matches_rel_filter",
                                    "dis_matches_fpr_query", "kmers_proccessed", "kmers_matched", "kmers_from_classified_seqs",
                                    "kmers_matched_perc" };

void sta_row(std::ostream& os, const std::string& prefix, const std::string& level, const ReadSetTally& t, const ReadSetTally& whole)
{
    const double denom = denominator_of(whole);
    const Digest d(t, denom);
    os << prefix << '\t' << level << '\t' << static_cast<size_t>(denom) << '\t' << whole.reads_seen - whole.reads_classified << '\t'
       << d.classified << '\t' << d.pct_classified << '\t' << d.unique << '\t' << d.pct_unique << '\t' << d.multiple << '\t'
       << d.pct_multiple << '\t' << d.matches << '\t' << d.matches_per_read << '\t' << d.dropped_filter << '\t' << d.dropped_fpr
       << '\t' << whole.minimisers_seen << '\t' << d.matched << '\t' << d.of_classified << '\t' << d.pct_matched << '\n';
}

} // namespace

void RunReport::write_sta(const std::string& output_prefix, const std::vector<std::string>& level_labels) const
{
    for (const auto& [prefix, whole] : overall_)
    {
        std::ofstream out(output_prefix + prefix + ".sta");
        out << std::fixed << std::setprecision(6);
        const char* sep = "";
        for (const char* c : kStaColumns)
        {
            out << sep << c;
            sep = "\t";
        }
        out << '\n';
        for (const auto& label : level_labels)
            sta_row(out, prefix, label, tally_or_empty(per_level_, label, prefix), whole);
        if (level_labels.size() > 1)
            sta_row(out, prefix, "-total-", whole, whole);
    }
}

void RunReport::print(std::ostream& os, double classify_seconds, const std::vector<std::string>& level_labels) const
{
    const double mbp = all_bases_ / 1000000.0;
    os << "ganon-classify processed " << all_reads_ << " sequences (" << mbp << " Mbp) with " << all_minimisers_ << " k-mers in "
       << classify_seconds << " seconds (" << mbp / (classify_seconds / 60.0) << " Mbp/m)" << std::endl;
    for (const auto& [prefix, whole] : overall_)
    {
        if (overall_.size() > 1)
            os << "\n[" << prefix << "] " << whole.reads_seen << " sequences (" << whole.bases_seen / 1000000.0 << " Mbp) with "
               << whole.minimisers_seen << " k-mers" << std::endl;
        const double denom = denominator_of(whole);
        print_block(os, whole, denom, whole.reads_seen - whole.reads_classified);
        if (level_labels.size() > 1)
        {
            os << "\nBy database hierarchical level:" << std::endl;
            for (const auto& label : level_labels)
            {
                os << label << ":" << std::endl;
                print_block(os, tally_or_empty(per_level_, label, prefix), denom, 0);
            }
        }
    }
}

void print_timing_block(std::ostream& os, const Stopwatch& whole_run, const Stopwatch& loading, const Stopwatch& classifying)
{
    os << "ganon-classify        start time: " << whole_run.first_start_text() << "\n"
       << "ganon-classify          end time: " << whole_run.last_stop_text() << "\n"
       << "loading filter(s)    elapsed (s): " << loading.seconds() << " seconds\n"
       << "classifying+printing elapsed (s): " << classifying.seconds() << " seconds\n"
       << "total                elapsed (s): " << whole_run.seconds() << " seconds\n"
       << "----------------------------------------------------------------------\n"
       << std::endl;
}

} // namespace gnhost
