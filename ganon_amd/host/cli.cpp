// cli.cpp -- command line of the drop-in ganon-classify binary.
// Flag table, short names, help texts and exit behaviour follow
// /root/reference/src/ganon-classify/CommandLineParser.cpp:14-121 and main.cpp:7-17 (cxxopts there; a small
// hand-written parser here: "--name value", "--name=value", "-n value", comma-separated vectors that append when
// an option is repeated, boolean switches with an optional "=true/false").
#include "config.hpp"
#include "tunables.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace gnhost
{

namespace
{

enum class Kind { VecStr, VecDouble, Str, Bool, U16, Size, Devices, Help, Version };

struct Opt
{
    char        short_name; // 0 = none
    const char* long_name;
    Kind        kind;
    const char* help;
};

const Opt kOpts[] = {
    { 'r', "single-reads", Kind::VecStr, "single-end reads file[s] (comma-separated; flat, gzipped or bzip2-compressed)" },
    { 'p', "paired-reads", Kind::VecStr, "paired-end reads file[s] (comma-separated; flat, gzipped or bzip2-compressed)" },
    { 'b', "batch-reads", Kind::VecStr,
      "file describing several files of single- or paired-end reads to be processed in one run: prefix <tab> file1 "
      "[<tab> file2]. Prefixes can be repeated for multiple files." },
    { 'i', "ibf", Kind::VecStr, "ibf file[s] from ganon-build (comma-separated)" },
    { 'x', "tax", Kind::VecStr, "tax file[s] from ganon-build for LCA calculation (comma-separated)" },
    { 'y', "hierarchy-labels", Kind::VecStr,
      "Hierarchy labels to define level for classification. Hierarchy follows order of the sorted labels. Default: H1" },
    { 'c', "rel-cutoff", Kind::VecDouble,
      "Relative cutoff (i.e. percentage of minimizers). 0 for no cutoff. One or one per filter (comma-separated). "
      "Default: 0.2" },
    { 'd', "rel-filter", Kind::VecDouble,
      "Relative filter. Additional percentage of matches allowed (relative to the best match). 1 for no filtering. one "
      "or one per hierarchy label (comma-separated). Default: 0.0" },
    { 'f', "fpr-query", Kind::VecDouble,
      "Min. False positive for a query. 1 for no filtering. one or one per hierarchy label (comma-separated). Default: "
      "1.0" },
    { 'o', "output-prefix", Kind::Str,
      "Output prefix (prefix.rep, [prefix.one, prefix.all, prefix.unc]). With multi-level --hierarchy-labels a '.label' "
      "is added to the output. With many sequence prefixes, a '.prefix' is added to the output." },
    { 'l', "output-lca", Kind::Bool, "Runs and outputs file with lca classification (prefix.one)" },
    { 'a', "output-all", Kind::Bool, "Outputs file with all matches (prefix.all)" },
    { 'u', "output-unclassified", Kind::Bool, "Outputs unclassified read ids (prefix.unc)" },
    { 'z', "output-stats", Kind::Bool, "Outputs classification statistics (prefix.sta)" },
    { 's', "output-single", Kind::Bool, "Do not split output files (one and all) with multi-level --hierarchy-labels" },
    { 0, "hibf", Kind::Bool, "Input is an Hierarchical IBF (.hibf) generated from raptor." },
    { 0, "long-reads", Kind::Bool, "Classify reads with more than 65535 minimisers too (what the reference does when it is built with -DLONGREADS=ON)." },
    { 0, "reference-order", Kind::Bool,
      "Write .all lines and .rep rows in the order the reference's robin_hood hash maps iterate, instead of ascending target (needs "
      "--threads 1, where the reference's own order is deterministic). Restated from robin_hood 3.11 and libstdc++; UNVERIFIED against a "
      "run of the reference." },
    { 0, "inspect-filter", Kind::Str,
      "Parse this .ibf (or .hibf with --hibf) without loading it: print every header field with its offset and every consistency check "
      "(S*W*8 against the file size, hash_shift, technical bins, map sizes), exit 1 naming the first inconsistent field. No GPU needed." },
    { 0, "verify-filter", Kind::Str,
      "With one --ibf: a ganon-build input file (file <tab> target). Every minimiser of every listed file must be found in its target's bins "
      "(the check of the reference's build test); the first false negative is printed with hash, rows and bits. Exit 1 on any." },
    { 0, "skip-lca", Kind::Bool, "Skip LCA step." },
    { 0, "tax-root-node", Kind::Str, "Define alternative root node for LCA. Default: 1" },
    { 't', "threads", Kind::U16, "Number of threads" },
    { 0, "n-batches", Kind::Size, "Number of batches of n-reads to hold in memory. Default: 1000" },
    { 0, "n-reads", Kind::Size, "Number of reads for each batch. Default: 400" },
    { 0, "verbose", Kind::Bool, "Verbose output mode" },
    { 0, "quiet", Kind::Bool, "Quiet output mode (only outputs errors and warnings to the STDERR)" },
    { 0, "device", Kind::Devices,
      "MI355X device(s): one index, a comma-separated list or 'all' -- one classify worker per entry, one copy of the filters "
      "per GPU (an index listed twice: two workers sharing it), read batches shared out among the workers (extension; default "
      "$GANON_DEVICE, else GPU 0 with --threads workers, at least 2, at most 4)" },
    { 'h', "help", Kind::Help, "Print help" },
    { 'v', "version", Kind::Version, "Show version" },
};

const Opt* find_long(const std::string& n)
{
    for (auto const& o : kOpts)
        if (n == o.long_name)
            return &o;
    return nullptr;
}
const Opt* find_short(char c)
{
    for (auto const& o : kOpts)
        if (o.short_name && o.short_name == c)
            return &o;
    return nullptr;
}

std::vector<std::string> split_commas(const std::string& s)
{
    std::vector<std::string> out;
    std::string              cur;
    std::istringstream       is(s);
    while (std::getline(is, cur, ','))
        out.push_back(cur);
    return out;
}

void print_help()
{
    std::cerr << "Ganon classifier\nUsage:\n  ganon-classify [OPTION...]\n\n";
    for (auto const& o : kOpts)
    {
        std::string left = "  ";
        if (o.short_name)
        {
            left += '-';
            left += o.short_name;
            left += ", ";
        }
        else
            left += "    ";
        left += "--";
        left += o.long_name;
        if (o.kind != Kind::Bool && o.kind != Kind::Help && o.kind != Kind::Version)
            left += " arg";
        while (left.size() < 30)
            left += ' ';
        std::cerr << left << " " << o.help << "\n";
    }
    std::cerr << std::endl;
}

// "all" -> empty list (= every visible device); otherwise non-negative indices
std::vector<int> parse_devices(const std::string& v)
{
    std::vector<int> out;
    if (v == "all")
        return out;
    for (auto const& s : split_commas(v))
    {
        size_t pos = 0;
        int    d   = std::stoi(s, &pos);
        if (pos != s.size() || d < 0)
            throw std::invalid_argument("Argument '" + s + "' failed to parse");
        out.push_back(d);
    }
    if (out.empty())
        throw std::invalid_argument("Argument '" + v + "' failed to parse");
    return out;
}

bool parse_bool(const std::string& v)
{
    if (v == "1" || v == "true" || v == "True" || v == "t" || v == "T")
        return true;
    if (v == "0" || v == "false" || v == "False" || v == "f" || v == "F")
        return false;
    throw std::invalid_argument("Argument '" + v + "' failed to parse");
}

} // namespace

std::optional<Config> parse_command_line(int argc, char** argv, int& exit_code)
{
    exit_code = 0;
    if (argc == 1)
    {
        std::cerr << "Try 'ganon-classify -h/--help' for more information." << std::endl;
        exit_code = 1;
        return std::nullopt;
    }
    Config cfg;
    bool want_help = false, want_version = false;
    try
    {
        if (const std::string* d = tun().str(Knob::device))
        {
            cfg.devices       = parse_devices(*d);
            cfg.devices_given = true;
        }
        for (int i = 1; i < argc; ++i)
        {
            std::string arg = argv[i];
            const Opt*  opt = nullptr;
            std::string value;
            bool        has_value = false;
            if (arg.rfind("--", 0) == 0)
            {
                std::string name = arg.substr(2);
                auto        eq   = name.find('=');
                if (eq != std::string::npos)
                {
                    value     = name.substr(eq + 1);
                    name      = name.substr(0, eq);
                    has_value = true;
                }
                opt = find_long(name);
                if (!opt)
                    throw std::invalid_argument("Option '" + name + "' does not exist");
            }
            else if (arg.size() >= 2 && arg[0] == '-')
            {
                opt = find_short(arg[1]);
                if (!opt)
                    throw std::invalid_argument(std::string("Option '") + arg[1] + "' does not exist");
                if (arg.size() > 2)
                {
                    value     = arg.substr(arg[2] == '=' ? 3 : 2);
                    has_value = true;
                }
            }
            else
            {
                throw std::invalid_argument("Unexpected positional argument '" + arg + "'");
            }
            const bool needs_value = !(opt->kind == Kind::Bool || opt->kind == Kind::Help || opt->kind == Kind::Version);
            if (needs_value && !has_value)
            {
                if (i + 1 >= argc)
                    throw std::invalid_argument(std::string("Option '") + opt->long_name + "' is missing an argument");
                value     = argv[++i];
                has_value = true;
            }
            const std::string n = opt->long_name;
            switch (opt->kind)
            {
                case Kind::Help: want_help = true; break;
                case Kind::Version: want_version = true; break;
                case Kind::VecStr:
                {
                    auto  v = split_commas(value);
                    auto& dst = n == "single-reads"   ? cfg.single_reads
                                : n == "paired-reads" ? cfg.paired_reads
                                : n == "batch-reads"  ? cfg.batch_reads
                                : n == "ibf"          ? cfg.ibf
                                : n == "tax"          ? cfg.tax
                                                      : cfg.hierarchy_labels;
                    static std::vector<std::string> seen;
                    if (std::find(seen.begin(), seen.end(), n) == seen.end())
                    {
                        dst.clear(); // first use replaces the default
                        seen.push_back(n);
                    }
                    dst.insert(dst.end(), v.begin(), v.end());
                    break;
                }
                case Kind::VecDouble:
                {
                    auto& dst = n == "rel-cutoff" ? cfg.rel_cutoff : n == "rel-filter" ? cfg.rel_filter : cfg.fpr_query;
                    static std::vector<std::string> seen;
                    if (std::find(seen.begin(), seen.end(), n) == seen.end())
                    {
                        dst.clear();
                        seen.push_back(n);
                    }
                    for (auto const& s : split_commas(value))
                    {
                        size_t pos = 0;
                        double x   = std::stod(s, &pos);
                        if (pos != s.size())
                            throw std::invalid_argument("Argument '" + s + "' failed to parse");
                        dst.push_back(x);
                    }
                    break;
                }
                case Kind::Str:
                    if (n == "output-prefix")
                        cfg.output_prefix = value;
                    else if (n == "inspect-filter")
                        cfg.inspect_filter = value;
                    else if (n == "verify-filter")
                        cfg.verify_filter = value;
                    else
                        cfg.tax_root_node = value;
                    break;
                case Kind::Bool:
                {
                    const bool b = has_value ? parse_bool(value) : true;
                    if (n == "output-lca") cfg.output_lca = b;
                    else if (n == "output-all") cfg.output_all = b;
                    else if (n == "output-unclassified") cfg.output_unclassified = b;
                    else if (n == "output-stats") cfg.output_stats = b;
                    else if (n == "output-single") cfg.output_single = b;
                    else if (n == "hibf") cfg.hibf = b;
                    else if (n == "long-reads") cfg.long_reads = b;
                    else if (n == "reference-order") cfg.reference_order = b;
                    else if (n == "skip-lca") cfg.skip_lca = b;
                    else if (n == "verbose") cfg.verbose = b;
                    else if (n == "quiet") cfg.quiet = b;
                    break;
                }
                case Kind::U16:
                {
                    long x = std::stol(value);
                    if (x < 0 || x > 65535)
                        throw std::invalid_argument("Argument '" + value + "' failed to parse");
                    cfg.threads = (uint16_t)x;
                    break;
                }
                case Kind::Size:
                {
                    unsigned long long x = std::stoull(value);
                    if (n == "n-batches")
                        cfg.n_batches = (size_t)x;
                    else
                        cfg.n_reads = (size_t)x;
                    break;
                }
                case Kind::Devices:
                    cfg.devices       = parse_devices(value);
                    cfg.devices_given = true;
                    break;
            }
        }
    }
    catch (std::exception const& e)
    {
        std::cerr << "ERROR: " << e.what() << std::endl;
        exit_code = 1;
        return std::nullopt;
    }
    if (want_help)
    {
        print_help();
        return std::nullopt;
    }
    if (want_version)
    {
        std::cerr << "version: " << kVersion << std::endl;
        return std::nullopt;
    }
    return cfg;
}

} // namespace gnhost
