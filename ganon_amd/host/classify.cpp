// classify.cpp -- host pipeline of the drop-in ganon-classify binary.
//
// What a run does and what it writes follows the reference (/root/reference/src/ganon-classify/GanonClassify.cpp;
// line references inline); how it is organised does not:
//   * the two SeqAn3 calls of the hot loop (minimiser_hash :693-700, agent.bulk_count :514/:553) and the per-target
//     sum / cap / cutoff of select_matches (:516-540, :556-576) run on the MI355X behind the C ABI (backend.hpp);
//     the host merges the sparse per-filter results with the reference's insert rule (:531-537, :567-573)
//   * filters are streamed from disk into the HBM of every selected GPU (filter_io.hpp), never held in host memory
//   * reads travel in large batches (not --n-reads = 400) so that one kernel launch covers ~10^6 reads; --n-reads,
//     --n-batches and --threads are accepted for command-line compatibility
//   * one classify worker per GPU (--device 0,1,.. | all), each with its own replica of the filters -- the role the
//     reference gives its --threads classify workers over one reader (:1579-1597); batches are handed out as the
//     workers become free, results are post-processed in input order, counters are summed (:197-246,475-490)
//   * output order is deterministic: reads in input order, the matches of a read in filter order then ascending
//     target index, .rep rows in target order (the reference iterates robin_hood maps and, with --threads > 1,
//     interleaves reads arbitrarily; its own tests compare order-insensitively, tests/aux/Aux.hpp:56-68)
#include "backend.hpp"
#include "pipeline.hpp"
#include "post.hpp"
#include "cpu_tally.hpp"
#include "robin_order.hpp"
#include "config.hpp"
#include "filter_io.hpp"
#include "lca.hpp"
#include "plan.hpp"
#include "report.hpp"
#include "seq_io.hpp"
#include "startup.hpp"
#include "tunables.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <sys/stat.h>
#include <deque>
#include <functional>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <sched.h>
#include <thread>
#include <unordered_map>

namespace gnhost
{

// ---- the classifier (GanonClassify.cpp:1375-1674) -------------------------------------------------------------
static bool ganon_classify(Config config)
{
    struct JoinCleanups
    {
        ~JoinCleanups() { g_cleanups.join_all(); }
    } join_cleanups;
    Stopwatch whole_run, loading, classifying;
    whole_run.start();

    std::vector<Level> levels = make_level_plan(config);
    ReadPlan           reads;
    if (!make_read_plan(config, reads))
        return false;

    // every output prefix names files in a directory that exists (:1390-1397: a prefix that is a directory itself is left alone)
    for (auto const& entry : reads)
    {
        const std::filesystem::path target(config.output_prefix + entry.first);
        const std::filesystem::path dir = target.parent_path();
        if (!dir.empty() && !std::filesystem::is_directory(target))
            std::filesystem::create_directories(dir);
    }
    if (config.verbose)
    {
        tun().list_set(std::cerr);
        list_levels(std::cerr, levels);
        list_reads(std::cerr, reads);
        list_outputs(std::cerr, config, levels, reads);
    }

    // One worker (a thread with its own device streams) per --device entry; entries naming the same GPU share one copy of
    // the filters.  Without --device the reference's own knob decides: --threads classify workers (:1579-1597), here on
    // GPU 0 -- at least three, so that one batch's upload and another's fetch overlap a third one's kernels, at most four.
    std::vector<int> devices = config.devices;
    if (!config.devices_given)
        devices.assign(std::min<size_t>(4, std::max<size_t>(3, config.threads)), 0);
    std::string  err;
    const double t_rt     = StartupLog::now();
    auto         backends = make_backends(devices, err);
    StartupLog::get().span("HIP runtime up, devices opened (first device call)", t_rt);
    if (backends.empty())
    {
        std::cerr << "ERROR: " << err << std::endl;
        return false;
    }
    if (config.long_reads)
    {
        for (auto& b : backends)
            if (!b->set_long_reads(true))
            {
                std::cerr << "ERROR: --long-reads is not supported by this backend" << std::endl;
                return false;
            }
    }
    const size_t n_workers = backends.size();
    // every worker thread drives two sets of device streams in turn (see the worker loop); $GANON_HOST_LANES=1: one
    std::vector<std::vector<std::unique_ptr<Backend>>> twins(n_workers);
    for (size_t i = 0; i < n_workers; ++i)
        for (size_t l = 1; l < tun().size(Knob::lanes, 2); ++l)
        {
            auto t = backends[i]->twin();
            if (!t)
                break;
            if (config.long_reads)
                t->set_long_reads(true);
            twins[i].push_back(std::move(t));
        }
    auto each_twin = [&](auto&& f) {
        for (auto& v : twins)
            for (auto& t : v)
                f(*t);
    };
    if (config.verbose)
        for (auto& b : backends)
            std::cerr << "Backend: " << b->describe() << "\n";

    RunReport  report;
    std::mutex report_mutex;
    // one stream per read set and kind of output; .rep always, .unc on request (:1416-1419), .all/.one per level below
    std::map<std::string, std::ofstream> out_rep, out_all, out_lca, out_unc;
    auto open_for_every_prefix = [&](std::map<std::string, std::ofstream>& files, const std::string& suffix, std::ios_base::openmode mode) {
        for (auto const& entry : reads)
            files[entry.first].open(config.output_prefix + entry.first + "." + suffix, mode);
    };
    open_for_every_prefix(out_rep, "rep", std::ofstream::out);
    if (config.output_unclassified)
        open_for_every_prefix(out_unc, "unc", std::ofstream::out);

    BatchQueue  queue1(2 + 2 * n_workers);
    // uncompressed single-end FASTQ goes to the workers as it lies in the file when the backend finds the records itself
    const bool  raw_fastq = backends.front()->tokenises_fastq();
    // ... and gzip files are inflated on the device, their text handed to the workers there (with further hierarchy levels the letters
    // come back with the results: reads left unclassified are passed on with them) -- $GANON_HOST_DEVICE_INFLATE=0 keeps the host inflater
    Backend*    device_text = raw_fastq && !tun().off(Knob::device_inflate) ? backends.front().get() : nullptr;
    if (device_text)
    {
        // ... and when the device has the room beside what is still to come: the filters are loaded AFTER the reader starts, and an
        // inflater takes its buffers (9 GB for a file below 1.5 GB, 18 GB above, twice for a pair) the moment a file is opened.  Room = free memory now
        // - the filter files' sizes - a share for the workers' batch buffers (as placement.hpp keeps: an eighth, 16 GiB at most of half)
        uint64_t filter_bytes = 0;
        for (auto const& l : levels)
            for (auto const& f : l.filters)
            {
                std::error_code ec;
                const auto      sz = std::filesystem::file_size(f.ibf_file, ec);
                // (an HIBF's rows are padded to whole 128-byte lines on the device, gn_hibf_row_stride_words: 3 -> 4, 5..8 -> 8, 9..16 -> 16
                //  words -- at most twice the file's payload per IBF; the shapes are not known before the file is parsed, so the bound is used)
                filter_bytes += ec ? 0 : (uint64_t)sz * (config.hibf ? 2u : 1u);
            }
        const uint64_t fr      = device_text->free_device_bytes();
        const uint64_t reserve = std::max<uint64_t>(fr / 8, std::min<uint64_t>(fr / 2, 16ull << 30)) + (24ull << 30);
        const uint64_t need    = tun().size(Knob::device_inflate_room, 40ull << 30);
        if (fr < filter_bytes + reserve + need)
        {
            if (config.verbose)
                std::cerr << "[host input] gzip input is inflated by the host: " << (fr >> 20) << " MiB free on the device, " << (filter_bytes >> 20)
                          << " MiB of filters to come" << std::endl;
            device_text = nullptr;
        }
    }
    {
        std::vector<int> d = devices;
        std::sort(d.begin(), d.end());
        g_distinct_devices = (unsigned)std::max<size_t>(1, (size_t)(std::unique(d.begin(), d.end()) - d.begin()));
    }
    std::thread read_task(parse_reads, std::ref(queue1), std::ref(report), std::ref(report_mutex), std::cref(reads), raw_fastq, device_text, levels.size() > 1);
    struct Joiner
    {
        std::thread& t;
        BatchQueue&  q;
        ~Joiner()
        {
            g_devices_ready.run = true; // (an early error return: nobody must wait for a level that never comes)
            if (t.joinable())
            {
                ReadBatch b; // drain so that a blocked producer can finish after an early error return
                while (q.pop(b))
                    b.ticket.reset(); // (a piece nobody will look at: its file must not wait for it)
                t.join();
            }
        }
    } joiner{ read_task, queue1 };

    std::vector<ReadBatch> carried; // unclassified reads kept for the next hierarchy level (:811-830)
    std::mutex             timing_mutex;
    double sec_device = 0, sec_post = 0; // where the host's classify time goes ($GANON_HOST_TIMING=1 prints it)

    std::vector<std::string> level_labels;
    for (auto const& l : levels)
        level_labels.push_back(l.label);

    for (size_t level_no = 0; level_no < levels.size(); ++level_no)
    {
        Level&     level       = levels[level_no];
        const bool first_level = level_no == 0;
        const bool last_level  = level_no + 1 == levels.size();

        // ---- load_files (:1007-1039): every filter of the level goes to every GPU
        loading.start();
        std::vector<FilterMeta>                     filters(level.filters.size());
        std::vector<std::map<std::string, TaxNode>> filter_tax(filters.size());
        each_twin([](Backend& b) { b.clear_filters(); });
        for (auto& b : backends)
            b->clear_filters();
        ReplicatingSink sink(backends);
        for (size_t i = 0; i < filters.size(); ++i)
        {
            try
            {
                const double t_f = StartupLog::now();
                load_filter_file(level.filters[i].ibf_file, config.hibf, filters[i], sink);
                {
                    const LoadTiming& lt = last_load_timing();
                    char              note[400];
                    std::snprintf(note, sizeof(note),
                                  "%.3f GiB: header %.3f s, device allocation %.3f s, page-locking the staging buffers %.3f s, file -> staging %.3f s (%.1f GB/s), "
                                  "staging -> HBM (waits) %.3f s, finalise %.3f s",
                                  lt.payload_bytes / 1073741824.0, lt.parse_s, lt.begin_s, lt.staging_s, lt.pread_s,
                                  lt.pread_s > 0 ? lt.payload_bytes / lt.pread_s / 1e9 : 0.0, lt.sink_s, lt.end_s);
                    StartupLog::get().span("level " + level.label + ": filter " + std::to_string(i) + " into HBM", t_f, note);
                }
                if (!level.filters[i].tax_file.empty())
                    filter_tax[i] = load_tax(level.filters[i].tax_file);
            }
            catch (std::exception const& e)
            {
                std::cerr << "ERROR: loading ibf or tax files: " << e.what() << std::endl;
                return false;
            }
        }
        loading.stop();
        if (config.verbose || tun().is_set(Knob::timing))
            for (auto& b : backends)
            {
                const std::string where = b->placement();
                if (!where.empty())
                    std::cerr << "[placement] level " << level.label << ": " << where;
            }

        level.kmer_size   = filters[0].ibf_config.kmer_size;
        level.window_size = filters[0].ibf_config.window_size;
        for (auto const& f : filters) // :1481-1494
            if (f.ibf_config.kmer_size != level.kmer_size || f.ibf_config.window_size != level.window_size)
            {
                std::cerr << "ERROR: databases on the same hierarchy should share same k-mer and window sizes" << std::endl;
                return false;
            }

        // ---- level-wide node namespace: targets of every filter, then tax nodes
        std::vector<std::string>                  node_names;
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

        // The level's taxonomy: the union of the filters' tax files, an earlier file winning where two define a node
        // (:1324-1341); a target no file knows hangs below the root as "no rank" (:1343-1362).  Only levels with a tax file.
        std::map<std::string, TaxNode> tax;
        const bool                     has_tax = !level.filters[0].tax_file.empty();
        if (has_tax)
        {
            for (auto const& one : filter_tax)
                for (auto const& kv : one)
                    tax.emplace(kv.first, kv.second); // (emplace keeps what is there)
            for (auto const& f : filters)
                for (auto const& target : f.targets)
                {
                    if (tax.find(target) != tax.end())
                        continue;
                    tax.emplace(target, TaxNode{ config.tax_root_node, "no rank", target });
                    if (!config.quiet)
                        std::cerr << "WARNING: target [" << target << "] without tax entry, setting parent as root node ["
                                  << config.tax_root_node << "]" << std::endl;
                }
        }
        // the LCA structure over that tree (:1506-1515); without --skip-lca the root has to be one of its nodes
        LCA lca;
        if (!config.skip_lca)
        {
            if (tax.find(config.tax_root_node) == tax.end())
            {
                std::cerr << "Root node [" << config.tax_root_node << "] not found (--tax-root-node)" << std::endl;
                return false;
            }
            for (auto const& kv : tax)
                lca.addEdge(kv.second.parent, kv.first);
            lca.doEulerWalk(config.tax_root_node);
        }

        // every node an LCA or the root fallback can name gets its id now: the post stage runs on several threads and only
        // looks ids up (`.rep` rows come in id order: targets in filter order, then the remaining tax nodes by name)
        for (auto const& [target, node] : tax)
        {
            nid(target);
            nid(node.parent);
        }
        nid(config.tax_root_node);
        auto known_nid = [&](const std::string& s) -> uint32_t { return node_ids.at(s); };

        // --reference-order: what the reference's robin_hood maps need to be replayed (robin_order.hpp) -- the hash of every
        // node name (TMatches is keyed by the target string, :53) and, per filter, the rank of every target in the iteration
        // order of the filter's TMap (:55; filled from the bin map in file order, :1021-1025), which is the order
        // select_matches offers targets to the read's map (:516,556)
        std::vector<uint64_t>              name_hash;
        std::vector<std::vector<uint32_t>> map_rank(filters.size());
        if (config.reference_order)
        {
            name_hash.resize(node_names.size());
            for (size_t g = 0; g < node_names.size(); ++g)
                name_hash[g] = rh_hash(node_names[g]);
            RobinSlots            tmap;
            std::vector<uint32_t> order;
            for (size_t i = 0; i < filters.size(); ++i)
            {
                tmap.clear();
                for (size_t t = 0; t < filters[i].targets.size(); ++t)
                    tmap.insert((uint32_t)t, rh_hash(filters[i].targets[t]));
                tmap.order(order);
                map_rank[i].assign(filters[i].targets.size(), 0u);
                for (size_t k = 0; k < order.size(); ++k)
                    map_rank[i][order[k]] = (uint32_t)k;
            }
        }
        // report rows in the order they were first touched, over all read sets (TRep is ONE map keyed by (prefix, target), :180)
        std::vector<std::pair<std::string, uint32_t>>            touched_rows;
        std::map<std::string, std::vector<uint8_t>>              touched_seen;

        const auto file_mode = first_level || !config.output_single ? std::ofstream::out : std::ofstream::app; // :1542
        if (config.output_lca && !config.skip_lca)
            open_for_every_prefix(out_lca, level.suffix_one, file_mode);
        if (config.output_all)
            open_for_every_prefix(out_all, level.suffix_all, file_mode);

        // per-level tallies: prefix -> target tallies (dense by node id, grown on demand) / read tallies
        std::map<std::string, std::vector<TargetTally>> target_tallies;
        std::map<std::string, ReadSetTally>             read_tallies;
        std::vector<double> rel_cutoffs;
        for (auto const& fc : level.filters)
            rel_cutoffs.push_back(fc.rel_cutoff);

        // the backends drop what filter_matches would drop where the matches are produced; with several filters that share
        // target names they also replay the level's merge (the larger count wins, :531-537) and hand over the winners
        bool shared_targets = false;
        std::atomic<uint64_t> diag_unmerged{ 0 }, diag_fpr_evals{ 0 }; // (GANON_HOST_TIMING: what was left to the host)
        std::atomic<uint64_t> diag_raw_pieces{ 0 }, diag_raw_reads{ 0 }, diag_raw_void{ 0 }, diag_parsed_pieces{ 0 };
        {
            PostFilterSpec spec;
            spec.rel_filter = level.rel_filter;
            spec.fpr_query  = level.fpr_query;
            std::unordered_map<std::string, int> owner;
            for (size_t i = 0; i < filters.size(); ++i)
            {
                spec.target_fpr.push_back(filters[i].target_fpr);
                for (auto const& t : filters[i].targets)
                    if (!owner.emplace(t, (int)i).second)
                        spec.disjoint_targets = false;
            }
            spec.target_gid = target_gid;
            shared_targets  = !spec.disjoint_targets;
            // (--reference-order replays the read's whole map of matches: nothing may be dropped before the host sees it)
            const bool want = !tun().is_set(Knob::no_prefilter) && !config.reference_order;
            bool       on   = true;
            for (auto& be : backends)
                on = be->set_postfilter(want ? &spec : nullptr) && on;
            each_twin([&](Backend& be) { on = be.set_postfilter(want ? &spec : nullptr) && on; });
            if (tun().is_set(Knob::timing))
                std::cerr << "[prefilter] level " << level.label << ": filter_matches pre-pass on the device "
                          << (on ? "on" : "off") << " (" << filters.size() << " filter(s), targets "
                          << (spec.disjoint_targets ? "disjoint" : "shared between filters") << ")" << std::endl;
        }

        const double t_setup = StartupLog::now();
        g_devices_ready.run = false; // (no new inflate step beside the stream set-up, see DeviceGate)
        loading.start(); // (device-side setup belongs to loading: the reference's agents exist once its filters are read)
        // device streams for the largest batch, and one tiny batch through every worker context: buffers are allocated and the
        // kernels' code is loaded here, not with the first reads (the reader is parsing its first slabs meanwhile)
        {
            std::vector<std::thread> setup;
            auto warm = [&](Backend* be) {
                if (!be->active())
                    return;
                const double t_p = StartupLog::now();
                be->prepare(std::min<size_t>(batch_reads(), 1u << 20), std::min<size_t>(kBatchBases, tun().size(Knob::slab_bytes, 48u << 20)));
                StartupLog::get().span("  a worker context: device streams", t_p);
                const double t_w = StartupLog::now();
                be->warm_up(level.kmer_size, level.window_size, std::vector<double>(filters.size(), 1.0));
                StartupLog::get().span("  a worker context: warm-up batch + page-locking its batch and result buffers", t_w);
            };
            for (auto& b : backends)
                setup.emplace_back(warm, b.get());
            each_twin([&](Backend& b) { setup.emplace_back(warm, &b); });
            for (auto& t : setup)
                t.join();
        }
        loading.stop();
        g_devices_ready.run = true;
        StartupLog::get().span("level " + level.label + ": device streams and the warm-up batch (all worker contexts at once)", t_setup);
        if (config.verbose)
            StartupLog::get().print(std::cerr);
        std::vector<ReadBatch> next_carried;
        classifying.start();
        rusage ru_level0;
        getrusage(RUSAGE_SELF, &ru_level0);
        const auto level_t0 = std::chrono::steady_clock::now();
        auto       since    = [level_t0] { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - level_t0).count(); };
        EventSpan  ev_taken, ev_begun, ev_fetched, ev_posted, ev_merged; // ($GANON_HOST_TIMING: ramp-up and tail of the pipeline)

        // the post stage (post.cpp) runs on a small pool of threads, one batch each, over this level's read-only tables
        const PostContext post_cx{ config,    level,    filters,     target_gid,  node_names,     node_ids,       lca,          name_hash,
                                   map_rank,  first_level, last_level, shared_targets, diag_unmerged, diag_fpr_evals, timing_mutex, sec_post };
        auto post_stage = [&](ReadBatch& rb, const BatchResult& res, PostScratch& sc, PostOutput& po) { gnhost::post_stage(post_cx, rb, res, sc, po); };
        // a batch's share goes into the level's tallies and files; called in input order on this thread
        auto merge_stage = [&](ClassifiedBatch& cb) {
            PostOutput& po = cb.post;
            read_tallies[cb.rb.prefix].add(po.reads);
            auto& rows = target_tallies[cb.rb.prefix];
            if (rows.size() < po.targets.size())
                rows.resize(po.targets.size());
            for (size_t g = 0; g < po.targets.size(); ++g)
                rows[g].add(po.targets[g]);
            if (config.reference_order)
            {
                auto& seen = touched_seen[cb.rb.prefix];
                if (seen.size() < node_names.size())
                    seen.resize(node_names.size(), 0);
                for (uint32_t g : po.touched)
                    if (!seen[g])
                    {
                        seen[g] = 1;
                        touched_rows.emplace_back(cb.rb.prefix, g);
                    }
            }
            if (config.output_all)
                out_all[cb.rb.prefix].write(po.all.data(), (std::streamsize)po.all.size());
            if (config.output_lca && !config.skip_lca)
                out_lca[cb.rb.prefix].write(po.lca.data(), (std::streamsize)po.lca.size());
            if (config.output_unclassified)
                out_unc[cb.rb.prefix].write(po.unc.data(), (std::streamsize)po.unc.size());
            if (po.has_left)
            {
                po.left.seq = next_carried.size();
                next_carried.push_back(std::move(po.left));
                po.left     = ReadBatch();
                po.has_left = false;
            }
        };

        // ---- reader -> [one device worker per GPU] -> post stage (this thread), results consumed in input order
        {
            // post pool: three threads keep up with one GPU; a node's worth of GPUs needs a node's worth of post threads -- two per
            // distinct device, within half of the usable cores (scripts/host_ceiling.py: ~CPU seconds per million reads and group)
            size_t n_distinct_dev = 0;
            {
                std::vector<int> d = devices;
                std::sort(d.begin(), d.end());
                n_distinct_dev = (size_t)(std::unique(d.begin(), d.end()) - d.begin());
            }
            const size_t             n_post = (size_t)tun().size(Knob::post_threads,
                                                               std::max<size_t>(3, std::min<size_t>(2 * n_distinct_dev, std::max<size_t>(3, usable_cores() / 2))));
            // (what may pile up in front of the writer while one batch is late: every batch held there is page-locked memory that is missing elsewhere)
            InOrder                  ordered(tun().size(Knob::lanes, 2) * n_workers + n_post + 2); // (more than what workers, post pool and queues hold at once)
            BoundedQueue<ClassifiedBatch> classified(n_post + 1);
            std::atomic<size_t>      workers_left{ n_workers };
            std::atomic<bool>        failed{ false };
            std::mutex               err_mutex, carried_mutex;
            size_t                   carried_next = 0;
            // where a worker gets its next batch: the reader's queue on the first level, the batches kept from the
            // previous level afterwards
            auto next_batch = [&](ReadBatch& rb) -> bool {
                if (first_level)
                    return queue1.pop(rb);
                std::lock_guard<std::mutex> lk(carried_mutex);
                if (carried_next >= carried.size())
                    return false;
                rb     = std::move(carried[carried_next]);
                rb.seq = carried_next++;
                return true;
            };
            std::vector<std::thread> workers;
            for (size_t wi = 0; wi < n_workers; ++wi)
                workers.emplace_back([&, wi] {
                    // One thread, several lanes (the worker's backend and its twins: own device streams, same filters), each holding
                    // one batch: IDLE -> UPLOADED (raw text on its way, records not yet known) -> QUEUED (kernels queued) -> IDLE.
                    // Per round the thread queues the kernels of the batch it uploaded last round, starts the next upload, and
                    // then waits for its oldest batch -- so the link, the compute units and the host side of a worker overlap.
                    enum class St { idle, uploaded, queued };
                    struct Lane
                    {
                        Backend*        be = nullptr;
                        ClassifiedBatch cb;
                        St              st = St::idle;
                        uint64_t        age = 0;
                    };
                    std::vector<Lane> lanes;
                    if (backends[wi]->active()) // (a level with a partitioned filter runs on a few of the workers)
                    {
                        lanes.emplace_back();
                        lanes.back().be = backends[wi].get();
                        for (auto& t : twins[wi])
                            if (t->active())
                            {
                                lanes.emplace_back();
                                lanes.back().be = t.get();
                            }
                    }
                    auto fail = [&](const std::string& e) {
                        std::lock_guard<std::mutex> lk(err_mutex);
                        if (!failed.exchange(true))
                            err = e;
                        ordered.abort();
                        return false;
                    };
                    auto timed = [&](auto&& f) {
                        const auto t0 = std::chrono::steady_clock::now();
                        const bool ok = f();
                        std::lock_guard<std::mutex> lk(timing_mutex);
                        sec_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        return ok;
                    };
                    // UPLOADED -> QUEUED: the records (a file's pieces are accepted in file order), then the kernels
                    auto queue_kernels = [&](Lane& x) -> bool {
                        std::string e;
                        ReadBatch&  rb = x.cb.rb;
                        if (rb.raw && rb.ticket)
                        {
                            uint32_t n = 0;
                            uint64_t parsed = 0;
                            if (!timed([&] { return x.be->tokenise_end(rb, n, parsed, e); }))
                                return fail(e);
                            // the piece says what it is right away; whether the pieces before it were all records is asked when its
                            // results arrive (they always are, unless the file is damaged: then this batch's results are dropped there)
                            rb.ticket->publish(parsed == rb.raw_bytes(), rb.text_at + parsed, rb.text2_at + rb.raw_parsed2);
                            rb.raw_keep = n;
                        }
                        if (!timed([&] { return x.be->classify_begin(rb, level.kmer_size, level.window_size, rel_cutoffs, e); }))
                            return fail(e);
                        x.st = St::queued;
                        ev_begun.mark(since());
                        return true;
                    };
                    // QUEUED -> IDLE: wait, fetch, hand over to the post pool
                    auto finish = [&](Lane& x) -> bool {
                        std::string e;
                        if (!timed([&] { return x.be->classify_end(x.cb.rb, level.kmer_size, level.window_size, rel_cutoffs, x.cb.res, e); }))
                            return fail(e);
                        ReadBatch& rb = x.cb.rb;
                        if (rb.ticket) // a piece of a file whose pieces are accepted in file order (text for the device, or parsed by a slab reader)
                        {
                            const size_t n = rb.size();
                            if (!rb.ticket->tracker->wait_prefix(rb.ticket->idx)) // a piece before this one stopped the file: none of this is input
                            {
                                rb.raw_keep = 0;
                                rb.rec_at.clear();
                                rb.seq_at.clear();
                                rb.seq_len.clear();
                                rb.seq_at2.clear();
                                rb.seq_len2.clear();
                                rb.id_buf.clear();
                                rb.id_off.assign(1, 0);
                                rb.bases.clear();
                                rb.off1.assign(1, 0);
                                x.cb.res.n_hashes.clear();
                                x.cb.res.status.clear();
                                x.cb.res.max_count.clear();
                                x.cb.res.prefiltered = false;
                                x.cb.res.dropped_rel_filter = x.cb.res.dropped_fpr_query = 0;
                                for (auto& fr : x.cb.res.per_filter)
                                {
                                    fr.match_off.assign(1, 0);
                                    fr.matches.clear();
                                    fr.fpr_ok.clear();
                                }
                            }
                            (rb.raw ? diag_raw_pieces : diag_parsed_pieces)++;
                            diag_raw_reads += rb.raw ? rb.size() : 0;
                            diag_raw_void += rb.size() == n ? 0 : 1;
                            if (rb.size())
                            {
                                std::lock_guard<std::mutex> lk(report_mutex);
                                report.count_input(rb.prefix, rb.size()); // :1253,1272
                            }
                        }
                        x.st = St::idle;
                        ev_fetched.mark(since());
                        classified.push(std::move(x.cb));
                        x.cb = ClassifiedBatch();
                        return true;
                    };
                    auto pick = [&](St st) -> Lane* { // the oldest lane in that state
                        Lane* best = nullptr;
                        for (auto& ln : lanes)
                            if (ln.st == st && (!best || ln.age < best->age))
                                best = &ln;
                        return best;
                    };
                    uint64_t age = 0;
                    bool     good = !lanes.empty(), more = true;
                    while (good)
                    {
                        if (Lane* u = pick(St::uploaded)) // last round's upload: its kernels
                            if (!(good = queue_kernels(*u)))
                                break;
                        if (failed) // (another worker gave up: deliver what is in flight, take nothing new)
                            more = false;
                        bool  took = false;
                        Lane* x    = more ? pick(St::idle) : nullptr;
                        if (x)
                        {
                            ordered.take_free(x->cb);
                            const bool in_flight = pick(St::queued) != nullptr;
                            int        got;
                            if (!first_level || !in_flight) // (nothing to do meanwhile: wait for the reader)
                                got = next_batch(x->cb.rb) ? 1 : -1;
                            else
                                got = queue1.try_pop(x->cb.rb);
                            if (got < 0)
                                more = false;
                            took = got > 0;
                        }
                        if (took)
                        {
                            // (too far ahead of the writer?  Never wait for it while holding results it may be waiting for: deliver first)
                            while (good && !ordered.try_turn(x->cb.rb.seq))
                            {
                                if (Lane* mine = pick(St::queued))
                                    good = finish(*mine);
                                else
                                {
                                    ordered.wait_turn(x->cb.rb.seq);
                                    break;
                                }
                            }
                            if (!good)
                                break;
                            ev_taken.mark(since());
                            x->age = age++;
                            std::string e;
                            if (x->cb.rb.raw && x->cb.rb.ticket)
                            {
                                if (!timed([&] { return x->be->tokenise_begin(x->cb.rb, e); })) // the text is on its way
                                {
                                    good = fail(e);
                                    break;
                                }
                                x->st = St::uploaded;
                            }
                            else if (!(good = queue_kernels(*x))) // (a parsed batch: upload and kernels in one go)
                                break;
                        }
                        Lane* q = pick(St::queued);
                        if (q && (!took || !pick(St::idle))) // every lane holds a batch, or there is nothing new to start: the oldest one's results
                        {
                            if (!(good = finish(*q)))
                                break;
                        }
                        else if (!took && !pick(St::uploaded) && !q && !more)
                            break; // everything delivered
                    }
                    g_cpu.worker.add_this_thread();
                    if (workers_left.fetch_sub(1) == 1)
                        classified.done(); // the last device worker out closes the post pool's queue
                });
            // post pool: batches in any order, results into `ordered`
            std::vector<std::thread> posters;
            for (size_t pi = 0; pi < n_post; ++pi)
                posters.emplace_back([&] {
                    PostScratch     scratch;
                    ClassifiedBatch cb;
                    while (classified.pop(cb))
                    {
                        post_stage(cb.rb, cb.res, scratch, cb.post);
                        ev_posted.mark(since());
                        const uint64_t seq = cb.rb.seq;
                        ordered.put(seq, std::move(cb));
                        cb = ClassifiedBatch();
                    }
                    g_cpu.post.add_this_thread();
                    ordered.producer_done();
                });
            ClassifiedBatch cb;
            uint64_t        merge_u0, merge_s0;
            CpuTally::thread_now(merge_u0, merge_s0);
            while (ordered.take(cb, n_post))
            {
                merge_stage(cb);
                ev_merged.mark(since());
                { // (a raw batch's text goes back to the allocator's pool right away, where the slab readers find it; the free lists would sit on it)
                    ByteBuf none;
                    none.swap(cb.rb.text);
                }
                if (first_level)
                    queue1.recycle(std::move(cb.rb));
                cb.rb = ReadBatch();
                ordered.recycle(std::move(cb));
                cb = ClassifiedBatch();
            }
            g_cpu.merge.add_since(merge_u0, merge_s0);
            if (failed && first_level)
            {
                ReadBatch b; // let the reader finish so that the workers blocked on it come back
                while (queue1.pop(b))
                    b.ticket.reset(); // (a piece nobody will look at: its file must not wait for it)
            }
            for (auto& w : workers)
                w.join();
            for (auto& t : posters)
                t.join();
            if (failed)
            {
                std::cerr << "ERROR: " << err << std::endl;
                return false;
            }
            if (first_level)
                read_task.join();
            if (config.verbose || tun().is_set(Knob::timing))
                for (auto& b : backends)
                {
                    const std::string moved = b->exchange_report();
                    size_t            at    = 0;
                    while (at < moved.size())
                    {
                        const size_t e = moved.find('\n', at);
                        std::cerr << "[gather] level " << level.label << ": " << moved.substr(at, e - at) << std::endl;
                        if (e == std::string::npos)
                            break;
                        at = e + 1;
                    }
                }
            if (tun().is_set(Knob::timing))
                std::cerr << "[host stalls] level " << level.label << ": reader blocked on a full batch queue " << queue1.blocked_push()
                          << " s, workers waiting for a batch " << queue1.blocked_pop() << " s (summed), workers waiting for their turn "
                          << ordered.blocked_turn() << " s (summed), post stage waiting for a result " << ordered.blocked_take() << " s"
                          << "; reads the pre-pass handed back whole " << diag_unmerged.load() << ", --fpr-query evaluations on the host "
                          << diag_fpr_evals.load() << std::endl;
            if (tun().is_set(Knob::timing))
            {
                std::cerr << "[host pipeline] level " << level.label << ", seconds since the level's start, first .. last: ";
                ev_taken.print(std::cerr, "batch taken by a worker");
                ev_begun.print(std::cerr, ", kernels queued");
                ev_fetched.print(std::cerr, ", results fetched");
                ev_posted.print(std::cerr, ", post-processed");
                ev_merged.print(std::cerr, ", merged + written");
                std::cerr << "; workers done at " << since() * 1e-6 << std::endl;
            }
            if (tun().is_set(Knob::timing) && diag_raw_pieces.load())
                std::cerr << "[host input] level " << level.label << ": " << diag_raw_pieces.load() << " pieces of FASTQ text tokenised on the device ("
                          << diag_raw_reads.load() << " reads), " << diag_parsed_pieces.load() << " parsed by the slab readers while the device side had text waiting, "
                          << diag_raw_void.load() << " dropped behind a piece that stopped its file" << std::endl;
        }
        carried.swap(next_carried);

        // reports (:1609-1617): `.rep` rows in node order (write_report :834-853)
        report.add_level(level.label, read_tallies, target_tallies);
        auto write_row = [&](const std::string& prefix, uint32_t gid) -> bool {
            auto it = target_tallies.find(prefix);
            if (it == target_tallies.end() || gid >= it->second.size())
                return true;
            const TargetTally& row = it->second[gid];
            if (!row.reported())
                return true;
            std::ofstream& rep = out_rep[prefix];
            rep << level.label << '\t' << node_names[gid] << '\t' << row.matches << '\t' << row.unique_reads << '\t' << row.lca_reads;
            if (!tax.empty())
            {
                auto node = tax.find(node_names[gid]);
                if (node == tax.end())
                {
                    std::cerr << "ERROR: node [" << node_names[gid] << "] not found in tax" << std::endl;
                    return false;
                }
                rep << '\t' << node->second.rank << '\t' << node->second.name;
            }
            rep << '\n';
            return true;
        };
        if (config.reference_order)
        {
            // the one classify thread's TRep in first-touch order, then sum_reports' copy of it (:475-490: a fresh map filled
            // in the first one's iteration order), whose iteration order write_report follows (:836)
            RobinSlots            thread_rep, summed;
            std::vector<uint32_t> order;
            std::vector<uint64_t> key_hash(touched_rows.size());
            thread_rep.clear();
            for (size_t x = 0; x < touched_rows.size(); ++x)
            {
                key_hash[x] = pair_hash(touched_rows[x].first, node_names[touched_rows[x].second]);
                thread_rep.insert((uint32_t)x, key_hash[x]);
            }
            thread_rep.order(order);
            summed.clear();
            for (uint32_t x : order)
                summed.insert(x, key_hash[x]);
            summed.order(order);
            for (uint32_t x : order)
                if (!write_row(touched_rows[x].first, touched_rows[x].second))
                    return false;
        }
        else
            for (auto& [prefix, rows] : target_tallies)
                for (uint32_t gid = 0; gid < rows.size(); ++gid)
                    if (!write_row(prefix, gid))
                        return false;
        classifying.stop();
        {
            // CPU of ALL threads between the level's start and its end -- what the steady state costs (the slab parsers' head start during
            // the filter load and the runtime's start-up are outside)
            rusage ru_level1;
            getrusage(RUSAGE_SELF, &ru_level1);
            auto us = [](const timeval& a, const timeval& b) { return (uint64_t)((b.tv_sec - a.tv_sec) * 1000000ll + (b.tv_usec - a.tv_usec)); };
            g_cpu.phase.user_us += us(ru_level0.ru_utime, ru_level1.ru_utime);
            g_cpu.phase.sys_us += us(ru_level0.ru_stime, ru_level1.ru_stime);
        }
        if (config.output_lca)
            for (auto& [prefix, file] : out_lca)
                file.close();
        if (config.output_all)
            for (auto& [prefix, file] : out_all)
                file.close();
    }
    each_twin([](Backend& b) { b.clear_filters(); });
    for (auto& b : backends)
        b->clear_filters();

    if (config.output_unclassified)
        for (auto& [prefix, file] : out_unc)
            file.close();

    // write_report_totals :855-863
    for (auto const& [prefix, files] : reads)
        report.touch(prefix); // every prefix has a (possibly empty) total
    for (auto const& [prefix, total] : report.overall())
    {
        out_rep[prefix] << "#total_classified\t" << total.reads_classified << '\n';
        out_rep[prefix] << "#total_unclassified\t" << total.reads_in - total.reads_classified << '\n';
    }
    for (auto& [prefix, file] : out_rep)
        file.close();
    if (config.output_stats)
        report.write_sta(config.output_prefix, level_labels);
    whole_run.stop();
    if (tun().is_set(Knob::timing))
        std::cerr << "[host timing] backend (upload+kernels+fetch, summed over " << n_workers << " worker(s)) " << sec_device
                  << " s, post-processing+writing " << sec_post << " s, loading filters " << loading.seconds()
                  << " s, classify+print wall " << classifying.seconds() << " s" << std::endl;
    if (tun().is_set(Knob::timing))
    {
        rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        std::cerr << "[host cpu] seconds user + system: ";
        g_cpu.parse.print(std::cerr, "slab parsers");
        g_cpu.reader.print(std::cerr, ", reader");
        g_cpu.mate.print(std::cerr, ", mate copiers");
        g_cpu.worker.print(std::cerr, ", device workers");
        g_cpu.post.print(std::cerr, ", post pool");
        g_cpu.merge.print(std::cerr, ", merge and write");
        g_cpu.inflate.print(std::cerr, ", device inflate feeders");
        g_cpu.phase.print(std::cerr, ", classify phase all threads");
        std::cerr << "; whole process " << ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 << " + " << ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6
                  << " on " << usable_cores() << " usable cores" << std::endl;
    }
    if (!config.quiet)
    {
        if (config.verbose)
            print_timing_block(std::cerr, whole_run, loading, classifying);
        report.print(std::cerr, classifying.seconds(), level_labels);
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
