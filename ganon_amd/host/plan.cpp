// plan.cpp -- see plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <set>

namespace gnhost
{

namespace
{

const char* const kRule = "----------------------------------------------------------------------\n";

std::vector<std::string> tab_fields(const std::string& line)
{
    std::vector<std::string> out;
    size_t                   from = 0;
    if (line.empty())
        return out;
    for (;;)
    {
        const size_t tab = line.find('\t', from);
        if (tab == std::string::npos)
        {
            if (from < line.size()) // (a trailing tab does not open an empty last field)
                out.push_back(line.substr(from));
            return out;
        }
        out.push_back(line.substr(from, tab - from));
        from = tab + 1;
    }
}

bool present_and_filled(const std::string& path)
{
    std::error_code ec;
    return std::filesystem::exists(path, ec) && std::filesystem::file_size(path, ec) > 0 && !ec;
}

} // namespace

bool make_read_plan(const Config& config, ReadPlan& plan)
{
    if (config.batch_reads.empty())
    {
        auto& direct = plan[""];
        for (const auto& f : config.single_reads)
            direct.push_back({ f, "" });
        for (size_t i = 0; i + 1 < config.paired_reads.size(); i += 2)
            direct.push_back({ config.paired_reads[i], config.paired_reads[i + 1] });
        if (direct.empty())
            plan.erase("");
        return true;
    }
    // table rows: prefix <tab> file1 [<tab> file2]; a prefix may repeat
    for (const auto& table : config.batch_reads)
    {
        std::ifstream in(table);
        std::string   line;
        while (std::getline(in, line))
        {
            const auto cols = tab_fields(line);
            if (cols.size() < 2)
            {
                std::cerr << "ERROR: invalid --batch-reads file (prefix <tab> file1 [<tab> file2])" << std::endl;
                return false;
            }
            const bool with_mate = cols.size() == 3;
            for (size_t c = 1; c <= (with_mate ? 2u : 1u); ++c)
                if (!present_and_filled(cols[c]))
                {
                    std::cerr << "ERROR: file not found/empty: " << cols[c] << std::endl;
                    return false;
                }
            plan[cols[0]].push_back({ cols[1], with_mate ? cols[2] : std::string() });
        }
    }
    return true;
}

std::vector<Level> make_level_plan(const Config& config)
{
    const size_t distinct = std::set<std::string>(config.hierarchy_labels.begin(), config.hierarchy_labels.end()).size();
    const bool   label_in_names = distinct > 1 && !config.output_single;

    std::vector<Level> levels; // in order of first appearance for now: that is how rel-filter / fpr-query are indexed
    for (size_t i = 0; i < config.hierarchy_labels.size(); ++i)
    {
        const std::string& label = config.hierarchy_labels[i];
        auto at = std::find_if(levels.begin(), levels.end(), [&](const Level& l) { return l.label == label; });
        if (at == levels.end())
        {
            Level fresh;
            fresh.label      = label;
            fresh.rel_filter = config.rel_filter[levels.size()];
            fresh.fpr_query  = config.fpr_query[levels.size()];
            fresh.suffix_one = label_in_names ? label + ".one" : "one";
            fresh.suffix_all = label_in_names ? label + ".all" : "all";
            levels.push_back(std::move(fresh));
            at = levels.end() - 1;
        }
        at->filters.push_back({ config.ibf[i], config.tax.empty() ? std::string() : config.tax[i], config.rel_cutoff[i] });
    }
    std::sort(levels.begin(), levels.end(), [](const Level& a, const Level& b) { return a.label < b.label; });
    return levels;
}

void list_levels(std::ostream& os, const std::vector<Level>& levels)
{
    os << "Database(s):\n";
    for (const auto& level : levels)
    {
        os << level.label << ":\n--rel-filter " << level.rel_filter << "\n--fpr-query " << level.fpr_query << "\n";
        for (const auto& f : level.filters)
        {
            if (f.rel_cutoff > -1)
                os << "--rel-cutoff " << f.rel_cutoff;
            os << " " << f.ibf_file;
            if (!f.tax_file.empty())
                os << ", " << f.tax_file;
            os << "\n";
        }
    }
    os << kRule;
}

void list_reads(std::ostream& os, const ReadPlan& plan)
{
    os << "Sequence(s):\n";
    for (const auto& [prefix, files] : plan)
    {
        if (!prefix.empty())
            os << prefix << ":\n";
        for (const auto& f : files)
            os << f.mate1 << (f.paired() ? ", " + f.mate2 : std::string()) << "\n";
    }
    os << kRule;
}

void list_outputs(std::ostream& os, const Config& config, const std::vector<Level>& levels, const ReadPlan& plan)
{
    os << "Output file(s):\n";
    for (const auto& [prefix, files] : plan)
    {
        (void)files;
        const std::string stem = config.output_prefix + prefix;
        if (!prefix.empty())
            os << prefix << ":\n";
        os << stem << ".rep\n";
        if (config.output_unclassified)
            os << stem << ".unc\n";
        if (config.output_stats)
            os << stem << ".sta\n";
        for (const auto& level : levels)
        {
            if (config.output_lca)
                os << stem << "." << level.suffix_one << "\n";
            if (config.output_all)
                os << stem << "." << level.suffix_all << "\n";
        }
    }
    os << kRule;
}

} // namespace gnhost
