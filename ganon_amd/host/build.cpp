// build.cpp -- `ganon-build` on MI355X: the flat-IBF builder behind the reference's own command line
// (/root/reference/src/ganon-build/CommandLineParser.cpp:14-33, Config.hpp, GanonBuild.cpp).
//
//   input file (file [<tab> target]) -> per target: distinct minimiser hashes of its files   (count_hashes :184-249)
//                                    -> bin capacity / filter size / hash functions          (optimal_hashes :427-616)
//                                    -> technical bins, equal shares per target              (create_bin_map_hash :619-653)
//                                    -> bits set                                             (build :655-698)
//                                    -> .ibf                                                 (save_filter :251-288)
//
// The device does what is data-parallel: sequences are cut into overlapping pieces (every window of a sequence lies in one
// piece), the classify-side minimiser kernels hash them, a radix sort + unique gives the file's hash SET
// (gn_stream_distinct_hashes), the filter is created empty in HBM, filled by an atomic-OR scatter
// (gn_filter_emplace_split) and streamed to the file.  The host parses FASTA, keeps the hash sets (RAM, not `.min` files:
// --tmp-output-folder is accepted and validated, nothing is written there) and does the sizing arithmetic.
// There is no CPU implementation of the hashing or the filter: without a HIP device the program fails.
//
// Differences from the reference that cannot be avoided here: targets are laid out in the order of
// their first appearance in the input file and a target's hashes in ascending order -- the reference uses the iteration
// order of robin_hood maps/sets, which is not reproducible without that library; which bin of a split target holds
// which hash therefore differs, the set of hashes per target, the sizing and every IBFConfig value do not.
// Sequences shorter than the window (but at least one k-mer long) yield the minimum over all their k-mers, which is what
// seqan3::views::minimiser does when the range is shorter than its window (recollection of SeqAn3 3.3.0, unpinned).
#include "build_params.hpp"
#include "hasher.hpp"
#include "hostmem.hpp"
#include "seq_io.hpp"
#include "tunables.hpp"

#include "ganon_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fcntl.h>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <thread>
#include <unistd.h>

namespace fs = std::filesystem;
using gnbuild::IbfParams;

namespace
{

constexpr const char* kVersion = "2.1.1-mi355x"; // as ganon-classify of this build (config.hpp)
constexpr int         kVersionTuple[3] = { 2, 1, 1 };

struct Config // Config.hpp:10-27
{
    std::string input_file, output_file, tmp_output_folder, mode = "avg";
    double      max_fp = 0.05, filter_size = 0;
    uint8_t     kmer_size = 19;
    uint16_t    window_size = 31;
    uint8_t     hash_functions = 0; // (parsed into an int first: 0..255)
    uint64_t    min_length = 0;
    uint16_t    threads = 1;
    bool        verbose = false, quiet = false;
    int         device = 0; // (not in the reference: which GPU)
};

bool validate(Config& c) // Config.hpp:29-107, same messages
{
    auto say = [&](const char* m) {
        if (!c.quiet)
            std::cerr << m << std::endl;
        return false;
    };
    if (c.input_file.empty())
        return say("--input-file is mandatory");
    if (!fs::exists(c.input_file))
    {
        if (!c.quiet)
            std::cerr << "--input-file not found: " << c.input_file << std::endl;
        return false;
    }
    if (fs::file_size(c.input_file) == 0)
    {
        if (!c.quiet)
            std::cerr << "--input-file is empty: " << c.input_file << std::endl;
        return false;
    }
    if (c.output_file.empty())
        return say("--output-file is mandatory");
    if (c.tmp_output_folder != "" && !fs::exists(c.tmp_output_folder))
        return say("--tmp-output-folder not found");
    if (c.hash_functions > gnbuild::kMaxHashFunctions)
        return say("--hash-functions must be <=5");
    if (c.filter_size == 0 && c.max_fp == 0)
        return say("--max-fp or --filter-size is mandatory");
    if (c.filter_size > 0)
        c.max_fp = 0;
    if (c.window_size < c.kmer_size)
        return say("--window-size has to be >= --kmer-size");
    if (c.mode != "avg" && c.mode != "smaller" && c.mode != "smallest" && c.mode != "faster" && c.mode != "fastest")
        return say("Invalid --mode");
    if (c.kmer_size > 32)
        return say("--kmer-size has to be <= 32");
    return true;
}

void print_config(const Config& c) // Config.hpp:110-133
{
    const char* sep = "----------------------------------------------------------------------";
    std::cerr << sep << '\n'
              << "--input-file        " << c.input_file << '\n'
              << "--output-file       " << c.output_file << '\n'
              << "--tmp-output-folder " << c.tmp_output_folder << '\n'
              << "--max-fp            " << c.max_fp << '\n'
              << "--filter-size       " << c.filter_size << '\n'
              << "--kmer-size         " << unsigned(c.kmer_size) << '\n'
              << "--window-size       " << c.window_size << '\n'
              << "--hash-functions    " << unsigned(c.hash_functions) << '\n'
              << "--mode              " << c.mode << '\n'
              << "--min-length        " << c.min_length << '\n'
              << "--threads           " << c.threads << '\n'
              << "--verbose           " << c.verbose << '\n'
              << "--quiet             " << c.quiet << '\n'
              << sep << '\n';
}

const char* kHelp =
    "Ganon builder (MI355X)\n"
    "Usage:\n"
    "  ganon-build [OPTION...]\n\n"
    "  -i, --input-file arg         Define sequences to use. Tabular file with the fields: file [<tab> target]\n"
    "  -o, --output-file arg        Filter output file\n"
    "  -k, --kmer-size arg          k-mer size. Default: 19\n"
    "  -w, --window-size arg        window size. Default: 31\n"
    "  -s, --hash-functions arg     number of hash functions. 0 to auto-detect. Default: 0\n"
    "  -p, --max-fp arg             Maximum false positive rate per target. Used to define filter size [mutually exclusive\n"
    "                               --filter-size]. Default: 0.05\n"
    "  -f, --filter-size arg        Filter size (MB) [mutually exclusive --max-fp]\n"
    "  -j, --mode arg               mode to build filter [avg, smaller, smallest, faster, fastest]. Default: avg\n"
    "  -y, --min-length arg         min. sequence length (bp) to keep. 0 to keep all. Default: 0\n"
    "  -m, --tmp-output-folder arg  Folder to write temporary files (accepted; this build keeps the hashes in memory)\n"
    "  -t, --threads arg            Number of threads (parser threads, one device stream each)\n"
    "      --device arg             HIP device index. Default: 0\n"
    "      --verbose                Verbose output mode\n"
    "      --quiet                  Quiet output mode\n"
    "  -h, --help                   Show help commands\n"
    "  -v, --version                Show current version\n";

// returns 0 = run, 1 = exit success, 2 = exit failure
int parse_args(int argc, char** argv, Config& c)
{
    if (argc == 1)
    {
        std::cerr << "Try 'ganon-build -h/--help' for more information." << std::endl;
        return 2;
    }
    static const std::map<std::string, std::string> shorts = {
        { "-i", "--input-file" },  { "-o", "--output-file" }, { "-k", "--kmer-size" },  { "-w", "--window-size" },
        { "-s", "--hash-functions" }, { "-p", "--max-fp" },    { "-f", "--filter-size" }, { "-j", "--mode" },
        { "-y", "--min-length" },  { "-m", "--tmp-output-folder" }, { "-t", "--threads" }, { "-h", "--help" }, { "-v", "--version" }
    };
    std::map<std::string, std::string> vals;
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i], v;
        bool        has = false;
        if (a.size() > 2 && a[0] == '-' && a[1] != '-') // attached short form, -k19 (what cxxopts takes as well)
        {
            v   = a.substr(a[2] == '=' ? 3 : 2);
            a   = a.substr(0, 2);
            has = true;
        }
        if (a.rfind("--", 0) == 0)
        {
            const size_t eq = a.find('=');
            if (eq != std::string::npos)
            {
                v   = a.substr(eq + 1);
                a   = a.substr(0, eq);
                has = true;
            }
        }
        auto s = shorts.find(a);
        if (s != shorts.end())
            a = s->second;
        if (a == "--help" || a == "--version" || a == "--verbose" || a == "--quiet")
        {
            vals[a] = has ? v : "true";
            continue;
        }
        static const std::set<std::string> known = { "--input-file", "--output-file", "--kmer-size", "--window-size",
                                                     "--hash-functions", "--max-fp", "--filter-size", "--mode", "--min-length",
                                                     "--tmp-output-folder", "--threads", "--device" };
        if (!known.count(a))
        {
            std::cerr << "Option '" << a << "' does not exist" << std::endl;
            return 2;
        }
        if (!has)
        {
            if (i + 1 >= argc)
            {
                std::cerr << "Option '" << a << "' is missing an argument" << std::endl;
                return 2;
            }
            v = argv[++i];
        }
        vals[a] = v;
    }
    if (vals.count("--help"))
    {
        std::cerr << kHelp << std::endl;
        return 1;
    }
    if (vals.count("--version"))
    {
        std::cerr << "version: " << kVersion << std::endl;
        return 1;
    }
    try
    {
        auto u = [&](const char* k, uint64_t hi) -> uint64_t {
            size_t             pos = 0;
            const std::string& s   = vals.at(k);
            const uint64_t     x   = std::stoull(s, &pos);
            if (pos != s.size() || s[0] == '-' || x > hi)
                throw std::invalid_argument(k);
            return x;
        };
        auto d = [&](const char* k) -> double {
            size_t             pos = 0;
            const std::string& s   = vals.at(k);
            const double       x   = std::stod(s, &pos);
            if (pos != s.size())
                throw std::invalid_argument(k);
            return x;
        };
        if (vals.count("--input-file"))
            c.input_file = vals["--input-file"];
        if (vals.count("--output-file"))
            c.output_file = vals["--output-file"];
        if (vals.count("--kmer-size"))
            c.kmer_size = (uint8_t)u("--kmer-size", 255);
        if (vals.count("--window-size"))
            c.window_size = (uint16_t)u("--window-size", 65535);
        if (vals.count("--hash-functions"))
            c.hash_functions = (uint8_t)u("--hash-functions", 255);
        if (vals.count("--max-fp"))
            c.max_fp = d("--max-fp");
        if (vals.count("--filter-size"))
            c.filter_size = d("--filter-size");
        if (vals.count("--mode"))
            c.mode = vals["--mode"];
        if (vals.count("--min-length"))
            c.min_length = u("--min-length", ~0ull);
        if (vals.count("--tmp-output-folder"))
            c.tmp_output_folder = vals["--tmp-output-folder"];
        if (vals.count("--threads"))
            c.threads = (uint16_t)u("--threads", 65535);
        if (vals.count("--device"))
            c.device = (int)u("--device", 1 << 20);
        c.verbose = vals.count("--verbose") && vals["--verbose"] != "false";
        c.quiet   = vals.count("--quiet") && vals["--quiet"] != "false";
    }
    catch (const std::exception& e)
    {
        std::cerr << "Argument '" << e.what() << "' failed to parse" << std::endl;
        return 2;
    }
    return 0;
}

struct Target
{
    std::string              name;
    std::vector<std::string> files;
    std::vector<uint64_t>    hashes; // per file: its distinct hashes, ascending; files behind each other (:236-238)
};

struct Totals // :52-59
{
    uint64_t files = 0, invalid_files = 0, sequences = 0, skipped_sequences = 0, length_bp = 0;
};

// parse_input_file (:88-140); targets in first-appearance order
std::vector<Target> read_input_file(const Config& c, Totals& totals)
{
    std::vector<Target>           targets;
    std::map<std::string, size_t> index;
    std::set<std::string>         files;
    std::ifstream                 in(c.input_file);
    std::string                   line;
    while (std::getline(in, line, '\n'))
    {
        if (line.empty())
            continue;
        std::vector<std::string> fields;
        std::istringstream       ls(line);
        std::string              f;
        while (std::getline(ls, f, '\t'))
            fields.push_back(f);
        if (fields.empty())
            continue;
        const std::string& file = fields[0];
        files.insert(file);
        std::error_code ec;
        if (!fs::exists(file, ec) || fs::file_size(file, ec) == 0)
        {
            if (!c.quiet)
                std::cerr << "WARNING: input file not found/empty: " << file << std::endl;
            totals.invalid_files++;
            continue;
        }
        std::string target;
        if (fields.size() == 1)
            target = fs::path(file).filename().string();
        else if (fields.size() == 2)
            target = fields[1];
        else
            continue; // (the reference handles one or two columns only)
        auto it = index.find(target);
        if (it == index.end())
        {
            it = index.emplace(target, targets.size()).first;
            targets.push_back(Target{ target, {}, {} });
        }
        targets[it->second].files.push_back(file);
    }
    totals.files = files.size();
    return targets;
}

// count_hashes (:184-249) for the targets this thread draws from the shared cursor
void hash_targets(const Config& c, std::vector<Target>& targets, std::atomic<size_t>& next, Totals& totals, std::string& fatal,
                  std::mutex& log_mutex)
{
    try
    {
        gnhost::Hasher    hasher(c.device, c.kmer_size, c.window_size);
        std::string       ids;
        gnhost::ByteBuf   seq;
        for (;;)
        {
            const size_t t = next.fetch_add(1);
            if (t >= targets.size())
                break;
            Target& tg = targets[t];
            for (const std::string& file : tg.files)
            {
                std::vector<uint64_t> file_hashes;
                unsigned              flushes = 0;
                try
                {
                    gnhost::SeqReader reader(file);
                    for (;;)
                    {
                        ids.clear();
                        seq.clear();
                        if (!reader.next(ids, seq))
                            break;
                        if (seq.size() < c.min_length)
                        {
                            totals.skipped_sequences++;
                            continue;
                        }
                        totals.sequences++;
                        totals.length_bp += seq.size();
                        hasher.add(seq.data(), seq.size(), file_hashes, flushes);
                    }
                    hasher.flush(file_hashes, flushes);
                    hasher.flush_short(file_hashes, flushes);
                }
                catch (const gnhost::ParseError& e)
                {
                    // the reference's catch (:242-246): the file contributes nothing, the next file goes on
                    std::lock_guard<std::mutex> lk(log_mutex);
                    std::cerr << "Error parsing file [" << file << "]. " << e.what() << std::endl;
                    unsigned dummy = 0;
                    std::vector<uint64_t> discard;
                    hasher.flush(discard, dummy);
                    hasher.flush_short(discard, dummy);
                    continue;
                }
                if (flushes > 1) // several device batches: their sets still have to be united
                {
                    std::sort(file_hashes.begin(), file_hashes.end());
                    file_hashes.erase(std::unique(file_hashes.begin(), file_hashes.end()), file_hashes.end());
                }
                tg.hashes.insert(tg.hashes.end(), file_hashes.begin(), file_hashes.end());
            }
        }
    }
    catch (const std::exception& e)
    {
        std::lock_guard<std::mutex> lk(log_mutex);
        fatal = e.what();
    }
}

// cereal BinaryOutputArchive encodings
struct Writer
{
    std::string buf;
    template <typename T>
    void raw(const T& v)
    {
        buf.append(reinterpret_cast<const char*>(&v), sizeof(T));
    }
    void str(const std::string& s)
    {
        raw<uint64_t>(s.size());
        buf.append(s);
    }
};

bool pwrite_all(int fd, const void* p, size_t n, uint64_t at)
{
    const char* c = static_cast<const char*>(p);
    while (n)
    {
        const ssize_t w = ::pwrite(fd, c, n, (off_t)at);
        if (w <= 0)
            return false;
        c += w;
        n -= (size_t)w;
        at += (uint64_t)w;
    }
    return true;
}

// save_filter (:251-288): header from the host, the bit matrix streamed out of HBM
bool save_filter(const Config& c, gn_filter* flt, const IbfParams& p, const std::vector<Target>& targets,
                 const std::vector<gnbuild::BinSpan>& bins, std::string& err)
{
    const uint64_t W = (p.n_bins + 63) >> 6;
    Writer         w;
    for (int v : kVersionTuple)
        w.raw<int32_t>(v);
    w.raw<uint64_t>(p.n_bins);
    w.raw<uint64_t>(p.max_hashes_bin);
    w.raw<uint8_t>(p.hash_functions);
    w.raw<uint8_t>(p.kmer_size);
    w.raw<uint16_t>(p.window_size);
    w.raw<uint64_t>(p.bin_size_bits);
    w.raw<double>(p.max_fp);
    w.raw<double>(p.true_max_fp);
    w.raw<double>(p.true_avg_fp);
    w.raw<uint64_t>(targets.size()); // hashes_count_std
    for (const Target& t : targets)
    {
        w.str(t.name);
        w.raw<uint64_t>(t.hashes.size());
    }
    w.raw<uint64_t>(bins.size()); // bin_map
    for (uint64_t b = 0; b < bins.size(); ++b)
    {
        w.raw<uint64_t>(b);
        w.str(targets[bins[b].target].name);
    }
    // seqan3::interleaved_bloom_filter: bins, technical_bins, bin_size, hash_shift, bin_words, hash_funs, sdsl bit_vector
    w.raw<uint64_t>(p.n_bins);
    w.raw<uint64_t>(W * 64);
    w.raw<uint64_t>(p.bin_size_bits);
    w.raw<uint64_t>((uint64_t)__builtin_clzll(p.bin_size_bits));
    w.raw<uint64_t>(W);
    w.raw<uint64_t>(p.hash_functions);
    w.raw<uint8_t>(1);      // sdsl int_vector<1>: width
    w.raw<float>(1.5f);     //                    growth factor
    w.raw<uint64_t>(W * 64 * p.bin_size_bits); // size in bits
    const int fd = ::open(c.output_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0)
    {
        err = "cannot write " + c.output_file;
        return false;
    }
    bool ok = pwrite_all(fd, w.buf.data(), w.buf.size(), 0);
    const uint64_t payload_at = w.buf.size();
    const uint64_t row_bytes  = W * 8;
    const uint64_t per        = std::max<uint64_t>(1, std::min<uint64_t>(256ull << 20, p.bin_size_bits * row_bytes) / row_bytes);
    void*          stage      = nullptr;
    if (ok && gn_pinned_alloc(per * row_bytes, &stage) != GN_OK)
    {
        err = gnhost::hip_error();
        ok  = false;
    }
    for (uint64_t row = 0; ok && row < p.bin_size_bits; row += per)
    {
        const uint64_t n = std::min<uint64_t>(per, p.bin_size_bits - row);
        if (gn_filter_download_rows(flt, 0, row, n, static_cast<uint64_t*>(stage)) != GN_OK)
        {
            err = gnhost::hip_error();
            ok  = false;
            break;
        }
        // a few writers per chunk: one pwrite stream does not fill a fast disk
        const unsigned           nt = (unsigned)std::min<uint64_t>(8, std::max<uint64_t>(1, n * row_bytes >> 24));
        std::vector<std::thread> th;
        std::atomic<bool>        good{ true };
        const uint64_t           bytes = n * row_bytes, share = (bytes + nt - 1) / nt;
        for (unsigned i = 0; i < nt; ++i)
            th.emplace_back([&, i] {
                const uint64_t lo = std::min<uint64_t>(bytes, i * share), hi = std::min<uint64_t>(bytes, lo + share);
                if (hi > lo && !pwrite_all(fd, static_cast<const char*>(stage) + lo, hi - lo, payload_at + row * row_bytes + lo))
                    good = false;
            });
        for (auto& t : th)
            t.join();
        if (!good)
        {
            err = "write error on " + c.output_file;
            ok  = false;
        }
    }
    if (stage)
        gn_pinned_free(stage);
    ::close(fd);
    return ok;
}

std::string stamp(std::chrono::system_clock::time_point t)
{
    const std::time_t tt = std::chrono::system_clock::to_time_t(t);
    char              b[64];
    std::strftime(b, sizeof(b), "%Y-%m-%d %H:%M:%S", std::localtime(&tt));
    return b;
}

struct Lap
{
    std::chrono::system_clock::time_point b, e;
    void   start() { b = std::chrono::system_clock::now(); }
    void   stop() { e = std::chrono::system_clock::now(); }
    double seconds() const { return std::chrono::duration<double>(e - b).count(); }
};

bool run(Config c)
{
    if (!validate(c))
        return false;
    if (c.verbose)
        print_config(c);
    Lap whole, counting, sizing, filling, writing;
    whole.start();

    int n_dev = 0;
    if (gn_device_count(&n_dev) != GN_OK || n_dev <= 0)
    {
        std::cerr << "no usable MI355X/HIP device (" << gn_last_error() << "); ganon-build has no CPU fallback" << std::endl;
        return false;
    }
    if (c.device >= n_dev)
    {
        std::cerr << "--device " << c.device << " does not exist (" << n_dev << " visible)" << std::endl;
        return false;
    }

    Totals              totals;
    std::vector<Target> targets = read_input_file(c, totals);
    if (targets.empty())
    {
        std::cerr << "No valid input files" << std::endl;
        return false;
    }

    counting.start();
    {
        // every hasher page-locks 64 MiB and owns a device stream with GBs of hash and sort buffers: --threads (the wrapper
        // forwards 64 or 128 gladly) buys parser threads only up to what eight such streams keep busy
        const unsigned           nt = std::max<unsigned>(1, std::min<unsigned>(std::min<unsigned>(c.threads, 8u), (unsigned)targets.size()));
        std::vector<Totals>      per(nt);
        std::vector<std::thread> th;
        std::atomic<size_t>      next{ 0 };
        std::string              fatal;
        std::mutex               log_mutex;
        for (unsigned i = 0; i < nt; ++i)
            th.emplace_back(hash_targets, std::cref(c), std::ref(targets), std::ref(next), std::ref(per[i]), std::ref(fatal),
                            std::ref(log_mutex));
        for (auto& t : th)
            t.join();
        if (!fatal.empty())
        {
            std::cerr << fatal << std::endl;
            return false;
        }
        for (const Totals& t : per)
        {
            totals.sequences += t.sequences;
            totals.skipped_sequences += t.skipped_sequences;
            totals.length_bp += t.length_bp;
        }
    }
    counting.stop();

    sizing.start();
    IbfParams p;
    p.kmer_size   = c.kmer_size;
    p.window_size = c.window_size;
    std::vector<uint64_t> counts;
    for (const Target& t : targets)
        counts.push_back(t.hashes.size());
    gnbuild::choose_capacity(c.max_fp, c.filter_size, counts, c.hash_functions, c.mode, p);
    if (p.n_bins != 0)
        gnbuild::true_fp(counts, p);
    sizing.stop();

    if (c.verbose) // :793-802
    {
        const char* sep = "----------------------------------------------------------------------";
        std::cerr << "ibf_config:" << '\n'
                  << "n_bins         " << p.n_bins << '\n'
                  << "max_hashes_bin " << p.max_hashes_bin << '\n'
                  << "hash_functions " << unsigned(p.hash_functions) << '\n'
                  << "kmer_size      " << unsigned(p.kmer_size) << '\n'
                  << "window_size    " << p.window_size << '\n'
                  << "bin_size_bits  " << p.bin_size_bits << '\n'
                  << "max_fp         " << p.max_fp << '\n'
                  << "true_max_fp    " << p.true_max_fp << '\n'
                  << "true_avg_fp    " << p.true_avg_fp << '\n'
                  << sep << '\n';
        std::cerr << "Filter size: " << (gnbuild::padded_bins(p.n_bins) * p.bin_size_bits) << " Bits";
        std::cerr << " (" << (gnbuild::padded_bins(p.n_bins) * p.bin_size_bits) / static_cast<double>(8388608u) << " Megabytes)"
                  << std::endl;
    }
    if (p.n_bins == 0)
    {
        std::cerr << "No valid sequences to build" << std::endl;
        return false;
    }

    std::vector<uint64_t>             shares;
    const std::vector<gnbuild::BinSpan> bins = gnbuild::lay_out_bins(p, counts, &shares);
    if (bins.size() != p.n_bins)
    {
        std::cerr << "internal error: " << bins.size() << " bins laid out, " << p.n_bins << " expected" << std::endl;
        return false;
    }

    if (p.bin_size_bits == 0 || p.hash_functions < 1 || p.hash_functions > 5)
    {
        // (the reference fails in the seqan3 IBF constructor: "The size of a bin must be > 0" / "hash functions must be > 0 and <= 5")
        std::cerr << "ERROR: the parameters leave a filter of " << p.bin_size_bits << " bits per bin with " << (unsigned)p.hash_functions
                  << " hash function(s): --filter-size / --max-fp do not fit " << targets.size() << " target(s)" << std::endl;
        return false;
    }
    filling.start();
    gn_filter*  flt = nullptr;
    gn_ibf_desc d{};
    d.bins = p.n_bins, d.bin_words = (p.n_bins + 63) >> 6, d.bin_size = p.bin_size_bits, d.hash_funs = p.hash_functions, d.rows = nullptr;
    d.hash_shift = (uint32_t)__builtin_clzll(p.bin_size_bits);
    if (gn_filter_upload_ibf(c.device, &d, nullptr, 0, &flt) != GN_OK) // storage only: no bin map
    {
        std::cerr << gn_last_error() << std::endl;
        return false;
    }
    {
        uint32_t first_bin = 0;
        for (size_t t = 0; t < targets.size(); ++t)
        {
            const uint64_t n = targets[t].hashes.size();
            if (n == 0)
                continue;
            if (gn_filter_emplace_split(flt, targets[t].hashes.data(), n, first_bin, shares[t]) != GN_OK)
            {
                std::cerr << gn_last_error() << std::endl;
                gn_filter_free(flt);
                return false;
            }
            first_bin += (uint32_t)((n + shares[t] - 1) / shares[t]);
        }
    }
    filling.stop();

    writing.start();
    std::string err;
    const bool  saved = save_filter(c, flt, p, targets, bins, err);
    gn_filter_free(flt);
    if (!saved)
    {
        std::cerr << err << std::endl;
        return false;
    }
    writing.stop();
    whole.stop();

    if (!c.quiet)
    {
        if (c.verbose) // print_stats_verbose (:730-757)
        {
            auto block = [](const char* a, const char* pad, const Lap& l) {
                std::cerr << a << stamp(l.b) << '\n' << pad << "    end: " << stamp(l.e) << '\n' << pad << "elapsed (s): " << l.seconds() << '\n';
            };
            block("Count/save hashes start: ", "                ", counting);
            block("Estimate params   start: ", "                ", sizing);
            block("Building filter   start: ", "                ", filling);
            block("Saving filer      start: ", "                ", writing);
            block("ganon-build       start: ", "                ", whole);
            std::cerr << std::endl;
        }
        const double elapsed = whole.seconds(); // print_stats (:706-728)
        std::cerr << "ganon-build processed " << totals.sequences << " sequences / " << totals.files << " files ("
                  << totals.length_bp / 1000000.0 << " Mbp) in " << elapsed << " seconds ("
                  << (totals.length_bp / 1000000.0) / (elapsed / 60.0) << " Mbp/m)" << std::endl;
        if (totals.invalid_files > 0)
            std::cerr << " - " << totals.invalid_files << " invalid files skipped" << std::endl;
        if (totals.skipped_sequences > 0)
            std::cerr << " - " << totals.skipped_sequences << " sequences skipped" << std::endl;
        std::cerr << std::fixed << std::setprecision(4) << " - max. false positive: " << p.true_max_fp;
        std::cerr << std::fixed << std::setprecision(4) << " (avg.: " << p.true_avg_fp << ")" << std::endl;
        std::cerr << std::fixed << std::setprecision(2)
                  << " - filter size: " << (gnbuild::padded_bins(p.n_bins) * p.bin_size_bits) / static_cast<double>(8388608u) << "MB"
                  << std::endl;
    }
    return true;
}

} // namespace

int main(int argc, char** argv)
{
    gnhost::HostTunables::init();
    Config    c;
    const int r = parse_args(argc, argv, c);
    if (r == 1)
        return EXIT_SUCCESS;
    if (r == 2)
        return EXIT_FAILURE;
    return run(c) ? EXIT_SUCCESS : EXIT_FAILURE;
}
