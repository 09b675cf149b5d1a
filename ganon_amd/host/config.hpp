// config.hpp -- ganon-classify configuration: same fields, defaults, validation messages and broadcast rules as the
// reference (/root/reference/src/ganon-classify/include/ganon-classify/Config.hpp:18-246), and the same flag table
// as /root/reference/src/ganon-classify/CommandLineParser.cpp:14-46 (parsed without cxxopts).
#pragma once

#include <algorithm>
#include <cstdint>
#include <filesystem>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

namespace gnhost
{

constexpr const char* kVersion = "2.1.1-mi355x";

struct Config
{
    std::vector<std::string> single_reads;
    std::vector<std::string> paired_reads;
    std::vector<std::string> batch_reads;

    std::vector<std::string> ibf;
    std::vector<std::string> tax;
    std::string              output_prefix;

    std::vector<std::string> hierarchy_labels{ "H1" };

    std::vector<double> rel_cutoff{ 0.2 };
    std::vector<double> rel_filter{ 0.0 };
    std::vector<double> fpr_query{ 1.0 };

    bool output_lca          = false;
    bool output_all          = false;
    bool output_unclassified = false;
    bool output_stats        = false;
    bool output_single       = false;

    bool        hibf          = false;
    bool        skip_lca      = false;
    std::string tax_root_node = "1";
    uint16_t    threads       = 1;
    size_t      n_batches     = 1000;
    size_t      n_reads       = 400;
    bool        verbose       = false;
    bool        quiet         = false;

    // device selection is an extension (not a reference flag): --device N, default $GANON_DEVICE or 0
    int device = 0;

    bool check_files(std::vector<std::string> const& files) const
    {
        for (auto const& file : files)
        {
            if (!std::filesystem::exists(file))
            {
                std::cerr << "file not found: " << file << std::endl;
                return false;
            }
            else if (std::filesystem::file_size(file) == 0)
            {
                std::cerr << "file is empty: " << file << std::endl;
                return false;
            }
        }
        return true;
    }

    static bool in_unit(std::vector<double> const& v)
    {
        for (double x : v)
            if (x < 0 || x > 1)
                return false;
        return true;
    }

    // Config.hpp:70-172
    bool validate()
    {
        if (output_prefix.size() == 0)
        {
            std::cerr << "--output-prefix is mandatory" << std::endl;
            return false;
        }
        if (paired_reads.size() == 0 && single_reads.size() == 0 && batch_reads.size() == 0)
        {
            std::cerr << "At least one of --[single|paired|batch]-reads is mandatory" << std::endl;
            return false;
        }
        if (ibf.size() == 0)
        {
            std::cerr << "--ibf is mandatory" << std::endl;
            return false;
        }
        if ((paired_reads.size() >= 1 || single_reads.size() >= 1) && batch_reads.size() >= 1)
        {
            std::cerr << "--batch-reads cannot be used together with --[single|paired]-reads" << std::endl;
            return false;
        }
        if (paired_reads.size() % 2 != 0)
        {
            std::cerr << "--paired-reads should be an even number of files (pairs)" << std::endl;
            return false;
        }
        if (!check_files(single_reads) || !check_files(paired_reads) || !check_files(batch_reads) || !check_files(ibf)
            || !check_files(tax))
            return false;
        if (!in_unit(rel_cutoff))
        {
            std::cerr << "--rel-cutoff values should be set between 0 and 1 (0 to disable)" << std::endl;
            return false;
        }
        if (!in_unit(rel_filter))
        {
            std::cerr << "--rel-filter values should be set between 0 and 1 (1 to disable)" << std::endl;
            return false;
        }
        if (!in_unit(fpr_query))
        {
            std::cerr << "--fpr-query values should be set between 0 and 1 (1 to disable)" << std::endl;
            return false;
        }
        if (n_batches < 1)
            n_batches = 1;
        if (n_reads < 1)
            n_reads = 1;
        if (!validate_hierarchy())
            return false;
        if (tax.size() == 0) // Disable LCA without tax files (:168-170)
            skip_lca = true;
        return true;
    }

    // Config.hpp:175-245
    bool validate_hierarchy()
    {
        std::vector<std::string> sorted_hierarchy = hierarchy_labels;
        std::sort(sorted_hierarchy.begin(), sorted_hierarchy.end());
        uint16_t unique_hierarchy = std::unique(sorted_hierarchy.begin(), sorted_hierarchy.end()) - sorted_hierarchy.begin();

        if (rel_filter.size() == 1 && unique_hierarchy > 1)
        {
            for (uint16_t b = 1; b < unique_hierarchy; ++b)
                rel_filter.push_back(rel_filter[0]);
        }
        else if (rel_filter.size() != unique_hierarchy)
        {
            std::cerr << "Please provide a single or one-per-hierarchy --rel-filter value[s]" << std::endl;
            return false;
        }
        if (fpr_query.size() == 1 && unique_hierarchy > 1)
        {
            for (uint16_t b = 1; b < unique_hierarchy; ++b)
                fpr_query.push_back(fpr_query[0]);
        }
        else if (fpr_query.size() != unique_hierarchy)
        {
            std::cerr << "Please provide a single or one-per-hierarchy --fpr-query value[s]" << std::endl;
            return false;
        }
        if (tax.size() > 0 && ibf.size() != tax.size())
        {
            std::cerr << "The number of files provided with --ibf and --tax should match" << std::endl;
            return false;
        }
        if (hierarchy_labels.size() == 1 && ibf.size() > 1)
        {
            for (uint16_t b = 1; b < ibf.size(); ++b)
                hierarchy_labels.push_back(hierarchy_labels[0]);
        }
        else if (hierarchy_labels.size() != ibf.size())
        {
            std::cerr << "--hierarchy does not match with the number of --ibf and --tax" << std::endl;
            return false;
        }
        if (rel_cutoff.size() == 1 && ibf.size() > 1)
        {
            for (uint16_t b = 1; b < ibf.size(); ++b)
                rel_cutoff.push_back(rel_cutoff[0]);
        }
        else if (rel_cutoff.size() != ibf.size())
        {
            std::cerr << "Please provide a single or one-per-filter --rel-cutoff value[s]" << std::endl;
            return false;
        }
        return true;
    }
};

// Config.hpp:248-288
inline std::ostream& operator<<(std::ostream& stream, const Config& config)
{
    constexpr auto newl{ "\n" };
    constexpr auto separator{ "----------------------------------------------------------------------" };
    stream << separator << newl;
    if (config.single_reads.size())
    {
        stream << "--single-reads        " << newl;
        for (const auto& s : config.single_reads)
            stream << "                      " << s << newl;
    }
    if (config.paired_reads.size())
    {
        stream << "--paired-reads        " << newl;
        for (const auto& s : config.paired_reads)
            stream << "                      " << s << newl;
    }
    if (config.batch_reads.size())
    {
        stream << "--batch-reads        " << newl;
        for (const auto& s : config.batch_reads)
            stream << "                      " << s << newl;
    }
    stream << "--output-prefix       " << config.output_prefix << newl;
    stream << "--output-lca          " << config.output_lca << newl;
    stream << "--output-all          " << config.output_all << newl;
    stream << "--output-unclassified " << config.output_unclassified << newl;
    stream << "--output-stats        " << config.output_stats << newl;
    stream << "--output-single       " << config.output_single << newl;
    stream << "--hibf                " << config.hibf << newl;
    stream << "--threads             " << config.threads << newl;
    stream << "--n-batches           " << config.n_batches << newl;
    stream << "--n-reads             " << config.n_reads << newl;
    stream << "--skip-lca            " << config.skip_lca << newl;
    stream << "--verbose             " << config.verbose << newl;
    stream << "--quiet               " << config.quiet << newl;
    stream << separator << newl;
    return stream;
}

// Returns the parsed config, or nullopt when the program should exit (exit_code set like main.cpp:7-17:
// 0 after -h/-v, 1 with no arguments or on a parse error).
std::optional<Config> parse_command_line(int argc, char** argv, int& exit_code);

} // namespace gnhost
