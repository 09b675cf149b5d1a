// config.cpp -- validation and printing of gnhost::Config.  Behavioural contract (messages, order of checks,
// broadcast of single values) = /root/reference/src/ganon-classify/include/ganon-classify/Config.hpp:50-288.
#include "config.hpp"

#include <algorithm>
#include <filesystem>
#include <iostream>
#include <set>

namespace gnhost
{

namespace
{

bool complain(const std::string& msg)
{
    std::cerr << msg << std::endl;
    return false;
}

// every listed file must exist and be non-empty (Config.hpp:50-67)
bool usable_files(const StringList& files)
{
    namespace fs = std::filesystem;
    for (const auto& f : files)
    {
        if (!fs::exists(f))
            return complain("file not found: " + f);
        if (fs::file_size(f) == 0)
            return complain("file is empty: " + f);
    }
    return true;
}

bool all_in_unit_interval(const DoubleList& v)
{
    return std::all_of(v.begin(), v.end(), [](double x) { return x >= 0 && x <= 1; });
}

// "one value or one per item": a single value is repeated `wanted` times, anything else must already match
template <typename T>
bool spread(std::vector<T>& values, size_t wanted, const char* msg)
{
    if (values.size() == 1 && wanted > 1)
    {
        values.resize(wanted, values.front());
        return true;
    }
    return values.size() == wanted ? true : complain(msg);
}

} // namespace

bool Config::validate()
{
    // ---- presence and shape of the arguments (Config.hpp:72-101)
    if (output_prefix.empty())
        return complain("--output-prefix is mandatory");
    const bool direct_reads = !single_reads.empty() || !paired_reads.empty();
    if (!direct_reads && batch_reads.empty())
        return complain("At least one of --[single|paired|batch]-reads is mandatory");
    if (ibf.empty())
        return complain("--ibf is mandatory");
    if (direct_reads && !batch_reads.empty())
        return complain("--batch-reads cannot be used together with --[single|paired]-reads");
    if (paired_reads.size() % 2 != 0)
        return complain("--paired-reads should be an even number of files (pairs)");

    // ---- files (:103-112)
    for (const StringList* lst : { &single_reads, &paired_reads, &batch_reads, &ibf, &tax })
        if (!usable_files(*lst))
            return false;

    // ---- value ranges (:114-156)
    if (!all_in_unit_interval(rel_cutoff))
        return complain("--rel-cutoff values should be set between 0 and 1 (0 to disable)");
    if (!all_in_unit_interval(rel_filter))
        return complain("--rel-filter values should be set between 0 and 1 (1 to disable)");
    if (!all_in_unit_interval(fpr_query))
        return complain("--fpr-query values should be set between 0 and 1 (1 to disable)");
    n_batches = std::max<size_t>(n_batches, 1);
    n_reads   = std::max<size_t>(n_reads, 1);

    // ---- per-hierarchy and per-filter values (:175-245)
    const size_t levels = std::set<std::string>(hierarchy_labels.begin(), hierarchy_labels.end()).size();
    if (!spread(rel_filter, levels, "Please provide a single or one-per-hierarchy --rel-filter value[s]"))
        return false;
    if (!spread(fpr_query, levels, "Please provide a single or one-per-hierarchy --fpr-query value[s]"))
        return false;
    if (!tax.empty() && tax.size() != ibf.size())
        return complain("The number of files provided with --ibf and --tax should match");
    if (!spread(hierarchy_labels, ibf.size(), "--hierarchy does not match with the number of --ibf and --tax"))
        return false;
    if (!spread(rel_cutoff, ibf.size(), "Please provide a single or one-per-filter --rel-cutoff value[s]"))
        return false;

    if (tax.empty()) // no taxonomy -> no LCA (:168-170)
        skip_lca = true;
    if (reference_order && threads != 1)
        return complain("--reference-order needs --threads 1 (with more threads the reference's own line order is not deterministic)");
    return true;
}

std::ostream& operator<<(std::ostream& os, const Config& c)
{
    static const char* rule = "----------------------------------------------------------------------";
    os << rule << '\n';
    const std::pair<const char*, const StringList*> lists[] = { { "--single-reads        ", &c.single_reads },
                                                                { "--paired-reads        ", &c.paired_reads },
                                                                { "--batch-reads        ", &c.batch_reads } };
    for (const auto& [label, files] : lists)
    {
        if (files->empty())
            continue;
        os << label << '\n';
        for (const auto& f : *files)
            os << "                      " << f << '\n';
    }
    os << "--output-prefix       " << c.output_prefix << '\n';
    const std::pair<const char*, long> scalars[] = {
        { "--output-lca          ", c.output_lca },   { "--output-all          ", c.output_all },
        { "--output-unclassified ", c.output_unclassified }, { "--output-stats        ", c.output_stats },
        { "--output-single       ", c.output_single }, { "--hibf                ", c.hibf },
        { "--threads             ", c.threads },      { "--n-batches           ", (long)c.n_batches },
        { "--n-reads             ", (long)c.n_reads }, { "--skip-lca            ", c.skip_lca },
        { "--verbose             ", c.verbose },      { "--quiet               ", c.quiet },
    };
    for (const auto& [label, value] : scalars)
        os << label << value << '\n';
    return os << rule << '\n';
}

} // namespace gnhost
