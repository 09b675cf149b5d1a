// classify.cpp -- host pipeline of the drop-in ganon-classify binary.
//
// Orchestration, per-read post-processing and every output format follow the reference
// (/root/reference/src/ganon-classify/GanonClassify.cpp; line references inline).  What is different by design:
//   * the two SeqAn3 calls of the hot loop (minimiser_hash :693-700, agent.bulk_count :514/:553) and the per-target
//     sum / cap / cutoff of select_matches (:516-540, :556-576) run on the MI355X behind the C ABI (backend.hpp);
//     the host merges the sparse per-filter results with the reference's insert rule (:531-537, :567-573)
//   * reads travel in large batches (not --n-reads = 400) so that one kernel launch covers ~10^6 reads; --n-reads,
//     --n-batches and --threads are accepted for command-line compatibility
//   * output order is deterministic: reads in input order, the matches of a read in filter order then ascending
//     target index, .rep rows in target order (the reference iterates robin_hood maps and, with --threads > 1,
//     interleaves reads arbitrarily; its own tests compare order-insensitively, tests/aux/Aux.hpp:56-68)
#include "backend.hpp"
#include "config.hpp"
#include "filter_io.hpp"
#include "lca.hpp"
#include "seq_io.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <ctime>
#include <deque>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace gnhost
{

namespace
{

// ---- StopClock (src/utils/include/utils/StopClock.hpp) ---------------------------------------------------
class StopClock
{
public:
    using Clock = std::chrono::system_clock;
    void start()
    {
        begin_round_ = Clock::now();
        if (first_)
        {
            begin_ = begin_round_;
            first_ = false;
        }
    }
    void stop()
    {
        end_ = Clock::now();
        run_ += end_ - begin_round_;
    }
    double            elapsed() const { return run_.count(); }
    Clock::time_point begin() const { return begin_; }
    Clock::time_point end() const { return end_; }

private:
    bool                          first_ = true;
    Clock::time_point             begin_, begin_round_, end_;
    std::chrono::duration<double> run_{ 0.0 };
};
auto datetime(const StopClock::Clock::time_point& tp)
{
    const auto t = std::chrono::system_clock::to_time_t(tp);
    return std::put_time(std::localtime(&t), "%F %T");
}

// ---- counters (GanonClassify.cpp:153-177) ----------------------------------------------------------------
struct Rep
{
    size_t matches = 0, seqs_lca = 0, seqs_unique = 0, discarded_matches_filter = 0, discarded_matches_fprquery = 0;
};
struct Total
{
    size_t input_seqs = 0, seqs_processed = 0, seqs_skipped_big = 0, seqs_skipped_small = 0, length_processed = 0,
           kmers_processed = 0, seqs_classified = 0, kmers_matches = 0, kmers_from_classified_seqs = 0, matches = 0,
           seqs_unique = 0, discarded_matches_filter = 0, discarded_matches_fprquery = 0;
};
using TTotal = std::map<std::string, Total>;

struct Stats
{
    TTotal                        total;
    std::map<std::string, TTotal> hierarchy_total;
    size_t                        total_seqs_processed = 0, total_length_processed = 0, total_kmers_processed = 0;

    void add_totals(const std::string& label, const TTotal& t_level) // :197-227
    {
        for (auto const& [prefix, t] : t_level)
        {
            total_seqs_processed += t.seqs_processed;
            total_length_processed += t.length_processed;
            total_kmers_processed += t.kmers_processed;
            for (Total* d : { &total[prefix], &hierarchy_total[label][prefix] })
            {
                d->seqs_processed += t.seqs_processed;
                d->seqs_skipped_big += t.seqs_skipped_big;
                d->seqs_skipped_small += t.seqs_skipped_small;
                d->length_processed += t.length_processed;
                d->kmers_processed += t.kmers_processed;
                d->seqs_classified += t.seqs_classified;
                d->kmers_matches += t.kmers_matches;
                d->kmers_from_classified_seqs += t.kmers_from_classified_seqs;
            }
        }
    }
    void add_report(const std::string& label, const std::string& prefix, const Rep& rep) // :229-246
    {
        for (Total* d : { &total[prefix], &hierarchy_total[label][prefix] })
        {
            d->matches += rep.matches;
            d->seqs_unique += rep.seqs_unique;
            d->discarded_matches_filter += rep.discarded_matches_filter;
            d->discarded_matches_fprquery += rep.discarded_matches_fprquery;
        }
    }
};

struct FilterConfig
{
    std::string ibf_file, tax_file;
    double      rel_cutoff = 0;
};
struct HierarchyConfig
{
    std::vector<FilterConfig> filters;
    uint8_t                   kmer_size   = 0;
    uint32_t                  window_size = 0;
    double                    rel_filter = 0, fpr_query = 1;
    std::string               output_file_lca, output_file_all;
};
using TReadConfig = std::map<std::string, std::vector<std::pair<std::string, std::string>>>;

// :289-351
bool parse_reads_config(Config& config, TReadConfig& reads_config)
{
    if (config.batch_reads.size() > 0)
    {
        std::string line;
        for (auto const& batch_file : config.batch_reads)
        {
            std::ifstream infile(batch_file);
            while (std::getline(infile, line, '\n'))
            {
                std::istringstream       stream_line(line);
                std::vector<std::string> fields;
                std::string              field;
                while (std::getline(stream_line, field, '\t'))
                    fields.push_back(field);
                if (fields.size() <= 1)
                {
                    std::cerr << "ERROR: invalid --batch-reads file (prefix <tab> file1 [<tab> file2])" << std::endl;
                    return false;
                }
                if (!std::filesystem::exists(fields[1]) || std::filesystem::file_size(fields[1]) == 0)
                {
                    std::cerr << "ERROR: file not found/empty: " << fields[1] << std::endl;
                    return false;
                }
                if (fields.size() == 3)
                {
                    if (!std::filesystem::exists(fields[2]) || std::filesystem::file_size(fields[2]) == 0)
                    {
                        std::cerr << "ERROR: file not found/empty: " << fields[2] << std::endl;
                        return false;
                    }
                    reads_config[fields[0]].push_back({ fields[1], fields[2] });
                }
                else
                    reads_config[fields[0]].push_back({ fields[1], "" });
            }
        }
    }
    else
    {
        for (auto const& reads_file : config.single_reads)
            reads_config[""].push_back({ reads_file, "" });
        for (size_t pair_cnt = 0; pair_cnt < config.paired_reads.size(); pair_cnt += 2)
            reads_config[""].push_back({ config.paired_reads[pair_cnt], config.paired_reads[pair_cnt + 1] });
    }
    return true;
}

// :353-401
std::map<std::string, HierarchyConfig> parse_hierarchy(Config& config)
{
    std::map<std::string, HierarchyConfig> parsed;
    std::vector<std::string>               sorted = config.hierarchy_labels;
    std::sort(sorted.begin(), sorted.end());
    const size_t unique_hierarchy = std::unique(sorted.begin(), sorted.end()) - sorted.begin();
    size_t       hierarchy_count  = 0;
    for (size_t h = 0; h < config.hierarchy_labels.size(); ++h)
    {
        FilterConfig fc{ config.ibf[h], "", config.rel_cutoff[h] };
        if (config.tax.size() > 0)
            fc.tax_file = config.tax[h];
        const std::string& label = config.hierarchy_labels[h];
        if (parsed.find(label) == parsed.end())
        {
            std::string lca = "one", all = "all";
            if (unique_hierarchy > 1 && !config.output_single)
            {
                lca = label + "." + lca;
                all = label + "." + all;
            }
            HierarchyConfig hc;
            hc.filters.push_back(fc);
            hc.rel_filter      = config.rel_filter[hierarchy_count];
            hc.fpr_query       = config.fpr_query[hierarchy_count];
            hc.output_file_lca = lca;
            hc.output_file_all = all;
            parsed[label]      = hc;
            ++hierarchy_count;
        }
        else
            parsed[label].filters.push_back(fc);
    }
    return parsed;
}

// :403-473
void print_hierarchy(const std::map<std::string, HierarchyConfig>& parsed)
{
    std::cerr << "Database(s):\n";
    for (auto const& [label, hc] : parsed)
    {
        std::cerr << label << ":\n";
        std::cerr << "--rel-filter " << hc.rel_filter << "\n";
        std::cerr << "--fpr-query " << hc.fpr_query << "\n";
        for (auto const& fc : hc.filters)
        {
            if (fc.rel_cutoff > -1)
                std::cerr << "--rel-cutoff " << fc.rel_cutoff;
            std::cerr << " " << fc.ibf_file;
            if (!fc.tax_file.empty())
                std::cerr << ", " << fc.tax_file;
            std::cerr << "\n";
        }
    }
    std::cerr << "----------------------------------------------------------------------\n";
}
void print_reads_config(const TReadConfig& rc)
{
    std::cerr << "Sequence(s):\n";
    for (auto const& [prefix, files] : rc)
    {
        if (!prefix.empty())
            std::cerr << prefix << ":\n";
        for (auto const& [f1, f2] : files)
        {
            std::cerr << f1;
            if (!f2.empty())
                std::cerr << ", " << f2;
            std::cerr << "\n";
        }
    }
    std::cerr << "----------------------------------------------------------------------\n";
}
void print_output_files(const Config& config, const std::map<std::string, HierarchyConfig>& parsed, const TReadConfig& rc)
{
    std::cerr << "Output file(s):\n";
    for (auto& [prefix, files] : rc)
    {
        if (!prefix.empty())
            std::cerr << prefix << ":\n";
        std::cerr << config.output_prefix + prefix + ".rep\n";
        if (config.output_unclassified)
            std::cerr << config.output_prefix + prefix + ".unc\n";
        if (config.output_stats)
            std::cerr << config.output_prefix + prefix + ".sta\n";
        for (auto& [label, hc] : parsed)
        {
            if (config.output_lca)
                std::cerr << config.output_prefix + prefix + "." + hc.output_file_lca << "\n";
            if (config.output_all)
                std::cerr << config.output_prefix + prefix + "." + hc.output_file_all << "\n";
        }
    }
    std::cerr << "----------------------------------------------------------------------\n";
}

inline size_t threshold_rel(size_t n_hashes, double p) // :492-495
{
    return std::ceil(n_hashes * p);
}
inline double binom(double n, double k) noexcept // :498-501
{
    return std::exp(std::lgamma(n + 1) - std::lgamma(n - k + 1) - std::lgamma(k + 1));
}

// ---- stats output (:1053-1218) ---------------------------------------------------------------------------
void print_stats_db(const Total& total, double seq_processed, size_t seq_unclassified)
{
    const size_t seq_multiple_matches = total.seqs_classified - total.seqs_unique;
    const double avg_seq_matches = total.seqs_classified ? (total.matches / static_cast<double>(total.seqs_classified)) : 0;
    const double kmers_matched_perc =
        total.kmers_matches ? (total.kmers_matches / static_cast<double>(total.kmers_from_classified_seqs)) * 100 : 0;
    std::cerr << "" << total.seqs_classified << " sequences classified (" << (total.seqs_classified / seq_processed) * 100
              << "%)" << std::endl;
    std::cerr << "  " << total.seqs_unique << " with unique matches (" << (total.seqs_unique / seq_processed) * 100 << "%)"
              << std::endl;
    std::cerr << "  " << seq_multiple_matches << " with multiple matches (" << (seq_multiple_matches / seq_processed) * 100
              << "%)" << std::endl;
    if (seq_unclassified > 0)
    {
        std::cerr << "" << seq_unclassified << " sequences unclassified (" << (seq_unclassified / seq_processed) * 100 << "%)"
                  << std::endl;
        if (total.seqs_skipped_small)
            std::cerr << "  " << total.seqs_skipped_small << " sequences skipped (shorter than window size)" << std::endl;
        if (total.seqs_skipped_big)
            std::cerr << "  " << total.seqs_skipped_big
                      << " sequences skipped (larger than allowed, check compilation with -DLONGREADS)" << std::endl;
    }
    std::cerr << "matches: " << total.matches << " (avg. " << avg_seq_matches << " reference/sequence), "
              << total.discarded_matches_filter << " discarded (--rel-filter), " << total.discarded_matches_fprquery
              << " discarded (--fpr-query)" << std::endl;
    std::cerr << "k-mers: " << total.kmers_matches << "/" << total.kmers_from_classified_seqs
              << " k-mers matched/k-mers from classified sequences" << " (" << kmers_matched_perc << "%)" << std::endl;
}

void print_stats(Stats& stats, double elapsed_classification, const std::map<std::string, HierarchyConfig>& parsed)
{
    std::cerr << "ganon-classify processed " << stats.total_seqs_processed << " sequences ("
              << stats.total_length_processed / 1000000.0 << " Mbp) with " << stats.total_kmers_processed << " k-mers in "
              << elapsed_classification << " seconds ("
              << (stats.total_length_processed / 1000000.0) / (elapsed_classification / 60.0) << " Mbp/m)" << std::endl;
    for (auto const& [prefix, total] : stats.total)
    {
        if (stats.total.size() > 1)
        {
            std::cerr << std::endl;
            std::cerr << "[" << prefix << "] " << total.seqs_processed << " sequences (" << total.length_processed / 1000000.0
                      << " Mbp) with " << total.kmers_processed << " k-mers" << std::endl;
        }
        const size_t seq_unclassified = total.seqs_processed - total.seqs_classified;
        const double seq_processed    = total.seqs_processed > 0 ? static_cast<double>(total.seqs_processed) : 1;
        print_stats_db(total, seq_processed, seq_unclassified);
        if (parsed.size() > 1)
        {
            std::cerr << std::endl;
            std::cerr << "By database hierarchical level:" << std::endl;
            for (auto const& h : parsed)
            {
                std::cerr << h.first << ":" << std::endl;
                print_stats_db(stats.hierarchy_total[h.first][prefix], seq_processed, 0);
            }
        }
    }
}

void write_stats_db(const Total& total, double seq_processed, size_t seq_unclassified, size_t kmers_processed,
                    const std::string& prefix, const std::string& level, std::ofstream& out)
{
    const size_t seq_multiple_matches = total.seqs_classified - total.seqs_unique;
    const double avg_seq_matches = total.seqs_classified ? (total.matches / static_cast<double>(total.seqs_classified)) : 0;
    const double kmers_matched_perc =
        total.kmers_matches ? (total.kmers_matches / static_cast<double>(total.kmers_from_classified_seqs)) * 100 : 0;
    out << std::fixed << std::setprecision(6);
    out << prefix << '\t' << level << '\t' << static_cast<size_t>(seq_processed) << '\t' << seq_unclassified << '\t'
        << total.seqs_classified << '\t' << (total.seqs_classified / seq_processed) * 100 << '\t' << total.seqs_unique << '\t'
        << (total.seqs_unique / seq_processed) * 100 << '\t' << seq_multiple_matches << '\t'
        << (seq_multiple_matches / seq_processed) * 100 << '\t' << total.matches << '\t' << avg_seq_matches << '\t'
        << total.discarded_matches_filter << '\t' << total.discarded_matches_fprquery << '\t' << kmers_processed << '\t'
        << total.kmers_matches << '\t' << total.kmers_from_classified_seqs << '\t' << kmers_matched_perc << '\n';
}

void write_stats(const std::string& output_prefix, Stats& stats, const std::map<std::string, HierarchyConfig>& parsed)
{
    for (auto const& [prefix, total] : stats.total)
    {
        std::ofstream out{ output_prefix + prefix + ".sta" };
        out << "prefix\thierarchy_label\tseq_processed\tseq_unclassified\tseq_classified\tseq_classified_perc\t"
               "seq_unique_matches\tseq_unique_matches_perc\tseq_multiple_matches\tseq_multiple_matches_perc\tmatches\t"
               "avg_matches_ref_seq\tdis_matches_rel_filter\tdis_matches_fpr_query\tkmers_proccessed\tkmers_matched\t"
               "kmers_from_classified_seqs\tkmers_matched_perc\n";
        const size_t seq_unclassified = total.seqs_processed - total.seqs_classified;
        const double seq_processed    = total.seqs_processed > 0 ? static_cast<double>(total.seqs_processed) : 1;
        for (auto const& h : parsed)
            write_stats_db(stats.hierarchy_total[h.first][prefix], seq_processed, seq_unclassified, total.kmers_processed,
                           prefix, h.first, out);
        if (parsed.size() > 1)
            write_stats_db(total, seq_processed, seq_unclassified, total.kmers_processed, prefix, "-total-", out);
    }
}

// ---- read side (:1220-1287): a producer thread turns files into large batches ------------------------------
// bounded producer/consumer queue; consumed items can be handed back so that their (already faulted-in) buffers
// are reused by the producer
template <typename T>
class BoundedQueue
{
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T&& b)
    {
        std::unique_lock<std::mutex> lk(m_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(b));
        not_empty_.notify_one();
    }
    void done()
    {
        std::lock_guard<std::mutex> lk(m_);
        done_ = true;
        not_empty_.notify_all();
    }
    bool pop(T& b)
    {
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty() || done_; });
        if (q_.empty())
            return false;
        b = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void recycle(T&& b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.size() < cap_ + 2)
            free_.push_back(std::move(b));
    }
    bool take_free(T& b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.empty())
            return false;
        b = std::move(free_.back());
        free_.pop_back();
        return true;
    }

private:
    std::mutex              m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T>           q_;
    std::vector<T>          free_;
    size_t                  cap_;
    bool                    done_ = false;
};
using BatchQueue = BoundedQueue<ReadBatch>;

// a batch together with what the device said about it
struct ClassifiedBatch
{
    ReadBatch   rb;
    BatchResult res;
};

constexpr size_t kBatchReads = 1u << 20;   // reads per device batch
constexpr size_t kBatchBases = 1ull << 28; // bases per device batch

// appends the mates-2 region behind the mates-1 region and rebases its offsets
void finalize_batch(ReadBatch& rb, std::vector<uint8_t>& bases2)
{
    if (!rb.paired)
        return;
    const uint64_t base = rb.bases.size();
    for (auto& o : rb.off2)
        o += base;
    rb.bases.insert(rb.bases.end(), bases2.begin(), bases2.end());
    bases2.clear();
}

// Second file of a pair, parsed on its own thread (for gzip input the reader is inflate-bound: two files, two inflate
// streams).  Only sequences are kept (ids come from file 1, :1243-1252); records arrive in blocks through a bounded
// queue; a parse error is delivered in place, after the records that precede it.
class MateStream
{
public:
    // the file is opened here, on the caller's thread: a file that cannot be opened fails before any record is read
    explicit MateStream(const std::string& path) : q_(8), in_(new SeqReader(path)), worker_([this] { run(); }) {}
    ~MateStream()
    {
        stop_ = true;
        Block b;
        while (q_.pop(b)) {} // unblock the producer
        worker_.join();
    }
    // appends the next mate to `bases`; false at end of file; throws the file's ParseError where it occurred
    bool next(std::vector<uint8_t>& bases)
    {
        while (pos_ == cur_.off.size() - 1)
        {
            if (cur_.last)
            {
                if (!cur_.error.empty())
                {
                    std::string e;
                    e.swap(cur_.error);
                    throw ParseError(e);
                }
                return false;
            }
            if (!q_.pop(cur_))
                return false;
            pos_ = 0;
        }
        bases.insert(bases.end(), cur_.bases.begin() + cur_.off[pos_], cur_.bases.begin() + cur_.off[pos_ + 1]);
        ++pos_;
        return true;
    }

private:
    struct Block
    {
        std::vector<uint8_t>  bases;
        std::vector<uint64_t> off{ 0 };
        bool                  last = false;
        std::string           error; // with last: the ParseError that ended the file
    };
    void run()
    {
        Block b;
        try
        {
            std::string id;
            while (!stop_)
            {
                id.clear();
                if (!in_->next(id, b.bases))
                    break;
                b.off.push_back(b.bases.size());
                if (b.off.size() > 65536 || b.bases.size() >= (16u << 20))
                {
                    q_.push(std::move(b));
                    b = Block();
                }
            }
        }
        catch (ParseError const& e)
        {
            b.error = e.what();
        }
        b.last = true;
        q_.push(std::move(b));
        q_.done();
    }
    BoundedQueue<Block>        q_;
    std::atomic<bool>          stop_{ false };
    Block                      cur_;
    size_t                     pos_ = 0;
    std::unique_ptr<SeqReader> in_;
    std::thread                worker_; // last member: everything above exists when it starts
};

void parse_reads(BatchQueue& queue, Stats& stats, std::mutex& stats_mutex, const TReadConfig& reads_config)
{
    for (auto const& [prefix, files] : reads_config)
    {
        for (auto const& [filename1, filename2] : files)
        {
            const bool           paired = !filename2.empty();
            ReadBatch            rb;
            std::vector<uint8_t> bases2; // mates 2 of the current batch
            auto                 fresh = [&]() {
                if (queue.take_free(rb))
                {
                    rb.id_buf.clear();
                    rb.id_off.assign(1, 0);
                    rb.bases.clear();
                }
                else
                    rb = ReadBatch();
                rb.paired = paired;
                rb.prefix = prefix;
                rb.off1.assign(1, 0);
                if (paired)
                    rb.off2.assign(1, 0);
                else
                    rb.off2.clear();
            };
            auto flush = [&]() {
                if (rb.size() == 0)
                    return;
                {
                    std::lock_guard<std::mutex> lk(stats_mutex);
                    stats.total[prefix].input_seqs += rb.size(); // :1253,1272
                }
                finalize_batch(rb, bases2);
                queue.push(std::move(rb));
                fresh();
            };
            fresh();
            try
            {
                SeqReader                   fin1(filename1);
                std::unique_ptr<MateStream> fin2;
                if (paired)
                    fin2.reset(new MateStream(filename2));
                while (fin1.next(rb.id_buf, rb.bases))
                {
                    rb.id_off.push_back(rb.id_buf.size());
                    rb.off1.push_back(rb.bases.size());
                    if (paired)
                    {
                        try
                        {
                            fin2->next(bases2); // at EOF the mate stays empty
                        }
                        catch (ParseError const&)
                        {
                            rb.off2.push_back(bases2.size());
                            throw;
                        }
                        rb.off2.push_back(bases2.size());
                    }
                    if (rb.size() >= kBatchReads || rb.bases.size() + bases2.size() >= kBatchBases)
                        flush();
                }
                flush();
            }
            catch (ParseError const& ext) // :1278-1283: report, keep what was read, go on with the next file
            {
                flush();
                std::cerr << "Error parsing file(s) [" << filename1 << ", " << filename2 << "]" << ext.what() << std::endl;
                continue;
            }
        }
    }
    queue.done();
}

struct Node
{
    std::string parent, rank, name;
};

} // namespace

// ---- the classifier (GanonClassify.cpp:1375-1674) -------------------------------------------------------------
static bool ganon_classify(Config config)
{
    StopClock timeGanon;
    timeGanon.start();

    auto        parsed_hierarchy = parse_hierarchy(config);
    TReadConfig reads_config;
    if (!parse_reads_config(config, reads_config))
        return false;

    for (auto& [prefix, files] : reads_config) // :1390-1397
    {
        std::filesystem::path filepath = std::string(config.output_prefix + prefix);
        if (!std::filesystem::is_directory(filepath) && !filepath.parent_path().empty())
            std::filesystem::create_directories(filepath.parent_path());
    }
    if (config.verbose)
    {
        print_hierarchy(parsed_hierarchy);
        print_reads_config(reads_config);
        print_output_files(config, parsed_hierarchy, reads_config);
    }

    std::string err;
    auto        backend = make_backend(config.device, err);
    if (!backend)
    {
        std::cerr << "ERROR: " << err << std::endl;
        return false;
    }
    if (config.verbose)
        std::cerr << "Backend: " << backend->describe() << "\n";

    StopClock timeLoadFilters, timeClassPrint;
    Stats     stats;
    std::mutex stats_mutex;
    std::map<std::string, std::ofstream> out_rep, out_all, out_lca, out_unc;
    for (auto& [prefix, files] : reads_config)
        out_rep[prefix].open(config.output_prefix + prefix + ".rep");
    if (config.output_unclassified)
        for (auto& [prefix, files] : reads_config)
            out_unc[prefix].open(config.output_prefix + prefix + ".unc");

    BatchQueue  queue1(4);
    std::thread read_task(parse_reads, std::ref(queue1), std::ref(stats), std::ref(stats_mutex), std::cref(reads_config));
    struct Joiner
    {
        std::thread& t;
        BatchQueue&  q;
        ~Joiner()
        {
            if (t.joinable())
            {
                ReadBatch b; // drain so that a blocked producer can finish after an early error return
                while (q.pop(b)) {}
                t.join();
            }
        }
    } joiner{ read_task, queue1 };

    std::vector<ReadBatch> carried; // unclassified reads kept for the next hierarchy level (:811-830)
    double sec_device = 0, sec_post = 0; // where the host's classify time goes ($GANON_HOST_TIMING=1 prints it)

    size_t       hierarchy_id   = 0;
    const size_t hierarchy_size = parsed_hierarchy.size();
    for (auto& [hierarchy_label, hierarchy_config] : parsed_hierarchy)
    {
        ++hierarchy_id;
        const bool hierarchy_first = hierarchy_id == 1;
        const bool hierarchy_last  = hierarchy_id == hierarchy_size;

        // ---- load_files (:1007-1039) + upload
        timeLoadFilters.start();
        std::vector<LoadedFilter> filters(hierarchy_config.filters.size());
        std::vector<std::map<std::string, TaxNode>> filter_tax(filters.size());
        backend->clear_filters();
        for (size_t i = 0; i < filters.size(); ++i)
        {
            try
            {
                if (config.hibf)
                    load_hibf_file(hierarchy_config.filters[i].ibf_file, filters[i]);
                else
                    load_ibf_file(hierarchy_config.filters[i].ibf_file, filters[i]);
                if (!hierarchy_config.filters[i].tax_file.empty())
                    filter_tax[i] = load_tax(hierarchy_config.filters[i].tax_file);
            }
            catch (std::exception const& e)
            {
                std::cerr << "ERROR: loading ibf or tax files: " << e.what() << std::endl;
                return false;
            }
            if (!backend->add_filter(filters[i], err))
            {
                std::cerr << "ERROR: loading ibf or tax files: " << err << std::endl;
                return false;
            }
        }
        timeLoadFilters.stop();

        hierarchy_config.kmer_size   = filters[0].ibf_config.kmer_size;
        hierarchy_config.window_size = filters[0].ibf_config.window_size;
        for (auto const& f : filters) // :1481-1494
            if (f.ibf_config.kmer_size != hierarchy_config.kmer_size || f.ibf_config.window_size != hierarchy_config.window_size)
            {
                std::cerr << "ERROR: databases on the same hierarchy should share same k-mer and window sizes" << std::endl;
                return false;
            }

        // ---- level-wide node namespace: targets of every filter, then tax nodes
        std::vector<std::string>                node_names;
        std::unordered_map<std::string, uint32_t> node_ids;
        auto nid = [&](const std::string& s) -> uint32_t {
            auto it = node_ids.find(s);
            if (it != node_ids.end())
                return it->second;
            const uint32_t id = (uint32_t)node_names.size();
            node_ids.emplace(s, id);
            node_names.push_back(s);
            return id;
        };
        std::vector<std::vector<uint32_t>> target_gid(filters.size());
        for (size_t i = 0; i < filters.size(); ++i)
            for (auto const& t : filters[i].targets)
                target_gid[i].push_back(nid(t));

        // tax: merge first-wins (:1324-1341), missing targets -> root (:1343-1362)
        std::map<std::string, TaxNode> tax;
        if (!hierarchy_config.filters[0].tax_file.empty())
        {
            tax = filter_tax[0];
            for (size_t i = 1; i < filters.size(); ++i)
                tax.insert(filter_tax[i].begin(), filter_tax[i].end());
            for (auto const& f : filters)
                for (auto const& target : f.targets)
                    if (tax.count(target) == 0)
                    {
                        tax[target] = TaxNode{ config.tax_root_node, "no rank", target };
                        if (!config.quiet)
                            std::cerr << "WARNING: target [" << target << "] without tax entry, setting parent as root node ["
                                      << config.tax_root_node << "]" << std::endl;
                    }
        }
        LCA lca;
        if (!config.skip_lca) // :1506-1515
        {
            if (tax.count(config.tax_root_node) == 0)
            {
                std::cerr << "Root node [" << config.tax_root_node << "] not found (--tax-root-node)" << std::endl;
                return false;
            }
            for (auto const& [target, node] : tax)
                lca.addEdge(node.parent, target);
            lca.doEulerWalk(config.tax_root_node);
        }

        const auto file_mode = hierarchy_first || !config.output_single ? std::ofstream::out : std::ofstream::app; // :1542
        if (config.output_lca && !config.skip_lca)
            for (auto& [prefix, files] : reads_config)
                out_lca[prefix].open(config.output_prefix + prefix + "." + hierarchy_config.output_file_lca, file_mode);
        if (config.output_all)
            for (auto& [prefix, files] : reads_config)
                out_all[prefix].open(config.output_prefix + prefix + "." + hierarchy_config.output_file_all, file_mode);

        // per-level report: prefix -> node id -> Rep
        std::map<std::string, std::vector<Rep>> rep; // dense by node id, grown on demand
        auto rep_at = [](std::vector<Rep>& v, uint32_t gid) -> Rep& {
            if (gid >= v.size())
                v.resize((size_t)gid + 1);
            return v[gid];
        };
        TTotal                                                   totals;
        std::vector<double>                                      rel_cutoffs;
        for (auto const& fc : hierarchy_config.filters)
            rel_cutoffs.push_back(fc.rel_cutoff);

        std::vector<ReadBatch> next_carried;
        timeClassPrint.start();

        struct MatchEntry
        {
            uint32_t gid;
            size_t   count;
            double   fpr;
        };
        std::vector<MatchEntry> matches;
        std::vector<uint32_t>    kept_gids;
        std::vector<std::string> kept_targets;
        std::string              buf_all, buf_lca, buf_unc; // one write per batch and file
        auto append_num = [](std::string& dst, size_t v) {
            char  tmp[24];
            char* e = tmp + sizeof(tmp);
            char* q = e;
            do
            {
                *--q = char('0' + v % 10);
                v /= 10;
            } while (v);
            dst.append(q, e - q);
        };

        std::mutex dev_mutex; // sec_device / err are written by the device stage
        auto device_stage = [&](const ReadBatch& rb, BatchResult& res) -> bool {
            const auto  t_dev0 = std::chrono::steady_clock::now();
            std::string e;
            const bool  ok = backend->classify(rb, hierarchy_config.kmer_size, hierarchy_config.window_size, rel_cutoffs, res, e);
            std::lock_guard<std::mutex> lk(dev_mutex);
            sec_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev0).count();
            if (!ok)
                err = e;
            return ok;
        };

        auto post_stage = [&](ReadBatch& rb, const BatchResult& res) -> bool {
            const auto t_dev1 = std::chrono::steady_clock::now();
            Total&    total  = totals[rb.prefix];
            auto&     prep   = rep[rb.prefix];
            buf_all.clear();
            buf_lca.clear();
            buf_unc.clear();
            ReadBatch left;
            left.paired = rb.paired;
            left.prefix = rb.prefix;
            std::vector<uint8_t> left2;
            left.off1.assign(1, 0);
            if (left.paired)
                left.off2.assign(1, 0);
            std::ofstream*           o_all = config.output_all ? &out_all[rb.prefix] : nullptr;
            std::ofstream*           o_lca = (config.output_lca && !config.skip_lca) ? &out_lca[rb.prefix] : nullptr;
            std::ofstream*           o_unc = config.output_unclassified ? &out_unc[rb.prefix] : nullptr;

            for (size_t r = 0; r < rb.size(); ++r)
            {
                const size_t read1_len = rb.len1(r), read2_len = rb.len2(r);
                const size_t n_hashes  = res.n_hashes[r];
                size_t       max_count_read = 0, min_count_read = n_hashes; // :686-688,704
                matches.clear();
                if (res.status[r] == 1) // :743-747
                {
                    if (hierarchy_first)
                        total.seqs_skipped_small++;
                }
                else if (res.status[r] == 2) // :737-741
                {
                    if (hierarchy_first)
                        total.seqs_skipped_big++;
                }
                else
                {
                    if (hierarchy_first) // :709-714
                    {
                        total.seqs_processed++;
                        total.length_processed += read1_len + read2_len;
                        total.kmers_processed += n_hashes;
                    }
                    for (size_t i = 0; i < filters.size(); ++i) // select_matches insert rule (:531-537)
                    {
                        const FilterResult& fr = res.per_filter[i];
                        for (uint64_t x = fr.match_off[r]; x < fr.match_off[r + 1]; ++x)
                        {
                            const Match&   m   = fr.matches[x];
                            const uint32_t gid = target_gid[i][m.target];
                            MatchEntry*    e   = nullptr;
                            for (auto& me : matches)
                                if (me.gid == gid)
                                {
                                    e = &me;
                                    break;
                                }
                            const size_t existing = e ? e->count : 0;
                            if (m.count > existing)
                            {
                                if (e)
                                {
                                    e->count = m.count;
                                    e->fpr   = filters[i].target_fpr[m.target];
                                }
                                else
                                    matches.push_back(MatchEntry{ gid, m.count, filters[i].target_fpr[m.target] });
                                if (m.count > max_count_read)
                                    max_count_read = m.count;
                                if (m.count < min_count_read)
                                    min_count_read = m.count;
                            }
                        }
                    }
                }

                bool classified = false;
                if (max_count_read > 0) // :753-808
                {
                    const size_t threshold_filter =
                        max_count_read - threshold_rel(max_count_read - min_count_read, hierarchy_config.rel_filter);
                    // filter_matches (:579-613)
                    size_t       kept = 0;
                    uint32_t     first_kept = 0;
                    const size_t all_mark = buf_all.size(); // lines of a read that ends up unclassified are dropped
                    kept_gids.clear();
                    for (auto const& me : matches)
                    {
                        if (me.count >= (double)threshold_filter)
                        {
                            if (hierarchy_config.fpr_query < 1.0)
                            {
                                double q = 1;
                                for (size_t i = 0; i <= me.count; i++)
                                    q -= binom(n_hashes, i) * pow(me.fpr, i) * pow(1 - me.fpr, n_hashes - i);
                                if (q > hierarchy_config.fpr_query)
                                {
                                    rep_at(prep, me.gid).discarded_matches_fprquery++;
                                    continue;
                                }
                            }
                            rep_at(prep, me.gid).matches++;
                            if (kept == 0)
                                first_kept = me.gid;
                            ++kept;
                            kept_gids.push_back(me.gid);
                            if (o_all)
                            {
                                buf_all += rb.id(r);
                                buf_all += '\t';
                                buf_all += node_names[me.gid];
                                buf_all += '\t';
                                append_num(buf_all, me.count);
                                buf_all += '\n';
                            }
                        }
                        else
                            rep_at(prep, me.gid).discarded_matches_filter++;
                    }
                    if (kept > 0)
                    {
                        classified = true;
                        total.seqs_classified++;
                        total.kmers_from_classified_seqs += n_hashes;
                        total.kmers_matches += max_count_read;
                        if (!config.skip_lca)
                        {
                            if (kept == 1) // :773-778
                            {
                                rep_at(prep, first_kept).seqs_unique++;
                                if (o_lca)
                                {
                                    // read_out_lca = read_out: the single kept match with its own count
                                    size_t c = 0;
                                    for (auto const& me : matches)
                                        if (me.gid == first_kept)
                                            c = me.count;
                                    buf_lca += rb.id(r);
                                    buf_lca += '\t';
                                    buf_lca += node_names[first_kept];
                                    buf_lca += '\t';
                                    append_num(buf_lca, c);
                                    buf_lca += '\n';
                                }
                            }
                            else // lca_matches :615-627
                            {
                                kept_targets.clear();
                                for (uint32_t g : kept_gids)
                                    kept_targets.push_back(node_names[g]);
                                const std::string target_lca = lca.getLCA(kept_targets);
                                rep_at(prep, nid(target_lca)).seqs_lca++;
                                if (o_lca)
                                {
                                    buf_lca += rb.id(r);
                                    buf_lca += '\t';
                                    buf_lca += target_lca;
                                    buf_lca += '\t';
                                    append_num(buf_lca, max_count_read);
                                    buf_lca += '\n';
                                }
                            }
                        }
                        else
                        {
                            if (kept == 1) // :790-793
                                rep_at(prep, first_kept).seqs_unique++;
                            else           // :794-799
                                rep_at(prep, nid(config.tax_root_node)).seqs_lca++;
                        }
                    }
                    else
                        buf_all.resize(all_mark);
                }
                if (classified)
                    continue;
                if (!hierarchy_last) // :811-820
                {
                    const std::string_view id = rb.id(r);
                    left.id_buf.append(id);
                    left.id_off.push_back(left.id_buf.size());
                    left.bases.insert(left.bases.end(), rb.bases.begin() + rb.off1[r], rb.bases.begin() + rb.off1[r] + read1_len);
                    left.off1.push_back(left.bases.size());
                    if (rb.paired)
                    {
                        left2.insert(left2.end(), rb.bases.begin() + rb.off2[r], rb.bases.begin() + rb.off2[r] + read2_len);
                        left.off2.push_back(left2.size());
                    }
                }
                else if (o_unc) // :821-825
                {
                    buf_unc += rb.id(r);
                    buf_unc += '\n';
                }
            }
            if (o_all)
                o_all->write(buf_all.data(), (std::streamsize)buf_all.size());
            if (o_lca)
                o_lca->write(buf_lca.data(), (std::streamsize)buf_lca.size());
            if (o_unc)
                o_unc->write(buf_unc.data(), (std::streamsize)buf_unc.size());
            if (!hierarchy_last && left.size() != 0)
            {
                finalize_batch(left, left2);
                next_carried.push_back(std::move(left));
            }
            sec_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev1).count();
            return true;
        };

        if (hierarchy_first)
        {
            // three stages on three threads: parse_reads -> queue1 -> device stage -> queue2 -> post stage (here)
            BoundedQueue<ClassifiedBatch> queue2(2);
            std::atomic<bool>             device_failed{ false };
            std::thread                   device_task([&] {
                ClassifiedBatch cb;
                while (queue2.take_free(cb), queue1.pop(cb.rb))
                {
                    if (!device_stage(cb.rb, cb.res))
                    {
                        device_failed = true;
                        break;
                    }
                    queue2.push(std::move(cb));
                }
                queue2.done();
            });
            struct DeviceJoiner
            {
                std::thread&                   t;
                BoundedQueue<ClassifiedBatch>& q;
                BatchQueue&                    q_in;
                ~DeviceJoiner()
                {
                    if (!t.joinable())
                        return;
                    ClassifiedBatch cb; // unblock the stage whichever queue it waits on, then join
                    while (q.pop(cb)) {}
                    t.join();
                }
            } device_joiner{ device_task, queue2, queue1 };
            ClassifiedBatch cb;
            while (queue2.pop(cb))
            {
                if (!post_stage(cb.rb, cb.res))
                    return false;
                queue1.recycle(std::move(cb.rb));
                cb.rb = ReadBatch();
                queue2.recycle(std::move(cb));
            }
            device_task.join();
            if (device_failed)
            {
                std::cerr << "ERROR: " << err << std::endl;
                return false;
            }
            read_task.join();
        }
        else
        {
            BatchResult res;
            for (auto& rb : carried)
            {
                if (!device_stage(rb, res))
                {
                    std::cerr << "ERROR: " << err << std::endl;
                    return false;
                }
                if (!post_stage(rb, res))
                    return false;
            }
        }
        carried.swap(next_carried);

        // reports (:1609-1617)
        stats.add_totals(hierarchy_label, totals);
        for (auto const& [prefix, pr] : rep)
            for (auto const& rp : pr)
                stats.add_report(hierarchy_label, prefix, rp);
        for (auto& [prefix, pr] : rep) // write_report :834-853, rows in node order
        {
            for (uint32_t gid = 0; gid < pr.size(); ++gid)
            {
                const Rep& report = pr[gid];
                if (report.matches || report.seqs_lca || report.seqs_unique)
                {
                    out_rep[prefix] << hierarchy_label << '\t' << node_names[gid] << '\t' << report.matches << '\t'
                                    << report.seqs_unique << '\t' << report.seqs_lca;
                    if (!tax.empty())
                    {
                        auto it = tax.find(node_names[gid]);
                        if (it == tax.end())
                        {
                            std::cerr << "ERROR: node [" << node_names[gid] << "] not found in tax" << std::endl;
                            return false;
                        }
                        out_rep[prefix] << '\t' << it->second.rank << '\t' << it->second.name;
                    }
                    out_rep[prefix] << '\n';
                }
            }
        }
        timeClassPrint.stop();
        if (config.output_lca)
            for (auto& [prefix, file] : out_lca)
                file.close();
        if (config.output_all)
            for (auto& [prefix, file] : out_all)
                file.close();
    }
    backend->clear_filters();

    if (config.output_unclassified)
        for (auto& [prefix, file] : out_unc)
            file.close();

    // write_report_totals :855-863
    for (auto const& [prefix, files] : reads_config)
        stats.total[prefix]; // make sure every prefix has a (possibly empty) total
    for (auto const& [prefix, total] : stats.total)
    {
        out_rep[prefix] << "#total_classified\t" << total.seqs_classified << '\n';
        out_rep[prefix] << "#total_unclassified\t" << total.input_seqs - total.seqs_classified << '\n';
    }
    for (auto& [prefix, file] : out_rep)
        file.close();
    if (config.output_stats)
        write_stats(config.output_prefix, stats, parsed_hierarchy);
    timeGanon.stop();
    if (std::getenv("GANON_HOST_TIMING"))
        std::cerr << "[host timing] backend (upload+kernels+fetch) " << sec_device << " s, post-processing+writing " << sec_post
                  << " s, classify+print wall " << timeClassPrint.elapsed() << " s" << std::endl;
    if (!config.quiet)
    {
        if (config.verbose) // print_time :1041-1051
        {
            std::cerr << "ganon-classify        start time: " << datetime(timeGanon.begin()) << std::endl;
            std::cerr << "ganon-classify          end time: " << datetime(timeGanon.end()) << std::endl;
            std::cerr << "loading filter(s)    elapsed (s): " << timeLoadFilters.elapsed() << " seconds" << std::endl;
            std::cerr << "classifying+printing elapsed (s): " << timeClassPrint.elapsed() << " seconds" << std::endl;
            std::cerr << "total                elapsed (s): " << timeGanon.elapsed() << " seconds" << std::endl;
            std::cerr << "----------------------------------------------------------------------" << std::endl;
            std::cerr << std::endl;
        }
        print_stats(stats, timeClassPrint.elapsed(), parsed_hierarchy);
    }
    return true;
}

// GanonClassify.cpp:1676-1691
bool run(Config config)
{
    if (!config.validate())
        return false;
    if (config.verbose)
        std::cerr << config;
    return ganon_classify(config);
}

} // namespace gnhost
